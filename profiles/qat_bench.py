#!/usr/bin/env python3
"""Time the fused QAT forward op (csrc/bnm_qat.hip) against the same expression in eager PyTorch on the same GPU.
Shapes: the reference's FC model layers (256->64, 64->64, 64->10) at training batch sizes.  Prints one JSON object."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitnetmcu_amd import qat


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # microseconds


def main():
    out = {}
    for n in (128, 4096, 65536):
        for d, k in ((256, 64), (64, 64), (64, 10)):
            x = torch.randn(n, d, device="cuda")
            w = torch.randn(k, d, device="cuda") * 0.08
            s = (w.abs().mean() / 0.25).reshape(1)
            with torch.no_grad():
                fused = timeit(lambda: qat.bitlinear_forward(x, w, s, "4bitsym", "RMS"))
                eager = timeit(lambda: qat.ste_formula(x, w, s[0], "4bitsym", "RMS"))
            out[f"n{n}_d{d}_k{k}"] = {"fused_us": round(fused, 2), "eager_torch_us": round(eager, 2),
                                      "speedup": round(eager / fused, 2)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
