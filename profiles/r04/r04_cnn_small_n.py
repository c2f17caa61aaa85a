#!/usr/bin/env python3
"""Round 4: the two CNN front ends over the number of images per call (device-resident images, median of 20 calls): the lane =
image kernel walks ALL channels in one wave per 32 images - its call time has a floor of one wave's walk - while the channel
kernel spreads one image's channels over a wave.  Where they cross decides which one a small call gets (bnm_capi.cpp).
usage (GPU box): python profiles/r04_cnn_small_n.py [model ...]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bitnetmcu_amd as b
    out = {}
    for name in sys.argv[1:] or ["cnn_64", "mcu_cnn_16", "mcu_cnn_48"]:
        model = b.Model.from_zoo(name)
        nmax = 1 << 19
        x = torch.empty((nmax, 256), dtype=torch.int8, device="cuda")
        b.synth.fill_device(x, first=0, dist=0)
        cls = torch.empty(nmax, dtype=torch.int32, device="cuda")
        rows = {}
        for n in (1, 32, 64, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288):
            row = {}
            for variant in (1, 3):
                ctx = b.Context(model)
                ctx.set_cnn_variant(variant)
                for _ in range(3):
                    ctx.infer_device(x[:n], cls[:n])
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
                ev[0].record()
                for k in range(20):
                    ctx.infer_device(x[:n], cls[:n])
                    ev[k + 1].record()
                torch.cuda.synchronize()
                row[variant] = float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(20)])) * 1e3
                ctx.close()
            rows[n] = row
            print(name, n, {k: round(v, 1) for k, v in row.items()}, "us", flush=True)
        out[name] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()
