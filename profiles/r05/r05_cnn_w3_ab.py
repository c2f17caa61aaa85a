#!/usr/bin/env python3
"""A/B of the one-kernel CNN: the product (mode 0) against experimental forms selected by BNM_CNN_W3 = mode (read at every launch:
1 = three waves per SIMD, software-pipelined; 2 = four waves with the cvt_pk epilogue) in ONE process, launches alternating;
class ids compared value for value with mode MODES[0]."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bitnetmcu_amd as b  # noqa: E402


MODES = [int(m) for m in os.environ.get("MODES", "0,1").split(",")]


def main():
    n = int(float(os.environ.get("N", "1e7")))
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    for name in (sys.argv[1:] or ["cnn_64", "mcu_cnn_16", "mcu_cnn_48"]):
        model = b.Model.from_zoo(name)
        ctx = b.Context(model)
        ctx.set_cnn_variant(3)
        cls = {m: torch.empty(n, dtype=torch.int32, device="cuda") for m in MODES}
        ms = {m: [] for m in MODES}
        for rnd in range(9):
            for m in MODES:
                os.environ["BNM_CNN_W3"] = str(m)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.infer_device(x, cls[m])
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ms[m].append(e0.elapsed_time(e1))
        out = {"model": name, "n": n, "kernel": ctx.last_kernel}
        for m in MODES:
            out[f"mode{m}"] = {"median_ms": round(float(np.median(ms[m])), 3), "inf_per_s": round(n / np.median(ms[m]) * 1e3),
                               "differing_class_ids": int((cls[MODES[0]] != cls[m]).sum().item())}
        print(json.dumps(out), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
