#!/usr/bin/env python3
"""A/B of the generic kernel's wave priorities (BNM_GENERIC_PRIO = 0 / 1 / 2, read at every launch) in ONE process on one box:
per model, the modes alternate launch by launch (7 rounds), median / min per mode; the class-id digest must not change."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bitnetmcu_amd as b  # noqa: E402


def main():
    n = int(float(os.environ.get("BNM_AB_N", "5e7")))
    modes = [int(m) for m in os.environ.get("BNM_AB_MODES", "0,1,2").split(",")]
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    for name, variant in [("doc12k_binary", -1), ("doc12k_ternary", -1), ("doc12k_2bit", -1), ("tern_96", -1), ("doc12k_8bit", -1), ("fc_4bitsym_64", 4)]:
        model = b.Model.from_zoo(name)
        ctx = b.Context(model)
        if variant >= 0:
            ctx.set_tuning(variant=variant)
        ms = {m: [] for m in modes}
        dig = {}
        for rnd in range(8):
            for m in modes:
                os.environ[os.environ.get("BNM_AB_KNOB", "BNM_GENERIC_PRIO")] = str(m)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.infer_device(x, cls)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ms[m].append(e0.elapsed_time(e1))
                else:
                    dig[m] = hex(int(b.synth.digest_device(cls, 0, model.num_classes).cpu().numpy()[0].astype(np.uint64)))
        out = {"model": name, "kernel": ctx.last_kernel, "n": n}
        for m in modes:
            out[f"prio{m}"] = {"median_ms": round(float(np.median(ms[m])), 4), "min_ms": round(float(np.min(ms[m])), 4), "digest": dig[m]}
        print(json.dumps(out), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
