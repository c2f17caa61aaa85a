#!/usr/bin/env python3
"""bnm_quantize_input_device (float32 [n,256] -> int8, the reference's input quantisation; 1280 B per image): ms and TB/s for 2x10^7
images, and the result against the numpy formula on a sample.  BNM_LIBRARY selects the build."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bitnetmcu_amd as b          # noqa: E402
from bitnetmcu_amd import harness  # noqa: E402
import util                        # noqa: E402


def main():
    n = int(os.environ.get("N", 20_000_000))
    ctx = b.Context(util.load_golden_model("fc_4bitsym_64"))
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.empty((n, 256), dtype=torch.float32, device="cuda")
    for k in range(0, n, 2_000_000):
        x[k:k + 2_000_000].normal_(generator=g)
    out = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    for _ in range(2):
        ctx.quantize_device(x, out)
    torch.cuda.synchronize()
    ms = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.quantize_device(x, out)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    idx = torch.cat([torch.arange(0, 3000), torch.arange(n - 3001, n)]).cuda()
    want = harness.quantize_input(x[idx].cpu().numpy())
    ok = bool(np.array_equal(out[idx].cpu().numpy(), want))
    med = float(np.median(ms))
    print(f"{os.environ.get('BNM_LIBRARY', 'default')}: n={n} median {med:.3f} ms  {n * 1280 / med / 1e9:.2f} TB/s  sample_ok={ok}", flush=True)


if __name__ == "__main__":
    main()
