#!/usr/bin/env python3
"""Round 5: the generic kernel's 4-tile class with ONE (variant 7, the default) against TWO image tiles per wave (variant 8: every
fragment read from LDS feeds two MFMAs, two waves per SIMD instead of three), same process, interleaved, 10^8 resident images.
usage (GPU box): python profiles/r05_m4_t2_ab.py"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bitnetmcu_amd as b
    n = 100_000_000
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    for name in ("tern_96", "doc12k_2bit", "doc12k_ternary"):
        ctx = {v: b.Context(b.Model.from_zoo(name)) for v in (7, 8)}
        for v, c in ctx.items():
            c.set_tuning(variant=v)
        dig, ms = {}, {7: [], 8: []}
        for rnd in range(4):
            for v in (7, 8):
                c = ctx[v]
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
                c.infer_device(x, cls)
                ev[0].record()
                for k in range(5):
                    c.infer_device(x, cls)
                    ev[k + 1].record()
                torch.cuda.synchronize()
                ms[v] += [ev[k].elapsed_time(ev[k + 1]) for k in range(5)]
                dig[v] = b.synth.digest_device(cls, first=0, n_bins=10).cpu().numpy().tolist()
        same = dig[7] == dig[8]
        print(f"{name}: one tile per wave {np.median(ms[7]):.3f} ms (min {min(ms[7]):.3f}), two tiles per wave {np.median(ms[8]):.3f} ms (min {min(ms[8]):.3f}) "
              f"per 1e8 images; ratio {np.median(ms[8]) / np.median(ms[7]):.3f}; digests equal: {same}; kernels {ctx[7].last_kernel} / {ctx[8].last_kernel}", flush=True)
        for c in ctx.values():
            c.close()


if __name__ == "__main__":
    main()
