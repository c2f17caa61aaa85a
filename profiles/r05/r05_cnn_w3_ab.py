#!/usr/bin/env python3
"""A/B of the one-kernel CNN: four waves per SIMD (product) against the software-pipelined three-wave form (BNM_CNN_W3=1, read at
every launch) in ONE process, launches alternating; class ids compared value for value."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bitnetmcu_amd as b  # noqa: E402


def main():
    n = int(float(os.environ.get("N", "1e7")))
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    for name in (sys.argv[1:] or ["cnn_64", "mcu_cnn_16", "mcu_cnn_48"]):
        model = b.Model.from_zoo(name)
        ctx = b.Context(model)
        ctx.set_cnn_variant(3)
        cls = {m: torch.empty(n, dtype=torch.int32, device="cuda") for m in (0, 1)}
        ms = {0: [], 1: []}
        for rnd in range(9):
            for m in (0, 1):
                os.environ["BNM_CNN_W3"] = str(m)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.infer_device(x, cls[m])
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ms[m].append(e0.elapsed_time(e1))
        diff = int((cls[0] != cls[1]).sum().item())
        print(json.dumps({"model": name, "n": n, "kernel": ctx.last_kernel, "differing_class_ids": diff,
                          "four_waves": {"median_ms": round(float(np.median(ms[0])), 3), "inf_per_s": round(n / np.median(ms[0]) * 1e3)},
                          "three_waves_pipelined": {"median_ms": round(float(np.median(ms[1])), 3), "inf_per_s": round(n / np.median(ms[1]) * 1e3)}}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
