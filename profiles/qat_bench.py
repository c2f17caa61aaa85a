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
    # one training step (forward + backward, no optimizer) of the reference's FC topology 256-64-64-64-10
    # (models.py FCMNIST: BitLinear + ReLU stacks), fused-forward layers vs the same expression in eager PyTorch
    import torch.nn as nn

    class Eager(nn.Linear):
        def __init__(self, i, o):
            super().__init__(i, o, bias=False)
            self.s = nn.Parameter(torch.tensor(1.0), requires_grad=False)

        def forward(self, x):
            return qat.ste_formula(x, self.weight, self.s, "4bitsym", "RMS")

    def net(layer):
        dims = (256, 64, 64, 64, 10)
        mods = []
        for i in range(4):
            mods.append(layer(dims[i], dims[i + 1]))
            if i < 3:
                mods.append(nn.ReLU())
        m = nn.Sequential(*mods).cuda()
        for l in m:
            if hasattr(l, "s"):
                l.s.data = (l.weight.abs().mean() / 0.25).detach()
        return m

    fused_net = net(lambda i, o: qat.BitLinear(i, o, QuantType="4bitsym", NormType="RMS"))
    eager_net = net(Eager)
    for n in (128, 4096):
        x = torch.randn(n, 256, device="cuda")

        def step(m):
            m.zero_grad(set_to_none=True)
            m(x).square().mean().backward()

        out[f"train_step_n{n}"] = {"fused_forward_us": round(timeit(lambda: step(fused_net), 50), 1),
                                   "eager_torch_us": round(timeit(lambda: step(eager_net), 50), 1)}
        with torch.no_grad():
            out[f"forward_only_n{n}"] = {"fused_us": round(timeit(lambda: fused_net(x), 100), 1),
                                         "eager_torch_us": round(timeit(lambda: eager_net(x), 100), 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
