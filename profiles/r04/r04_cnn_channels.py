#!/usr/bin/env python3
"""Round 4: the two CNN front ends over channel counts (random-weight models in the reference's CNN topology, FC tail 96-64-10):
channel kernel (cnn_variant 1: a lane = a channel) against the lane = image kernel (3).  usage (GPU box): python profiles/r04_cnn_channels.py"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch
    import bitnetmcu_amd as b
    import test_gpu_parity as T
    n = 4_000_000
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    out = {}
    for C in [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 140]:
        rng = np.random.default_rng(C)
        w = lambda k: rng.integers(-128, 128, size=9 * C)
        model = b.Model.from_header_text(T._random_cnn_text(rng, C, (16, 4, 4), (96, 64), 10, w))
        row = {}
        ref = None
        for variant in (1, 3):
            ctx = b.Context(model)
            try:
                ctx.set_cnn_variant(variant)
            except b.BnmError:
                ctx.close()
                continue
            ctx.infer_device(x, cls)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            for k in range(3):
                ctx.infer_device(x, cls)
                ev[k + 1].record()
            torch.cuda.synchronize()
            ms = float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(3)]))
            d = b.synth.digest_device(cls, 0, 10).cpu().numpy()
            ref = d if ref is None else ref
            assert np.array_equal(d, ref), (C, variant)
            row[variant] = n / (ms * 1e-3)
            ctx.close()
        out[C] = row
        print(C, json.dumps(row), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
