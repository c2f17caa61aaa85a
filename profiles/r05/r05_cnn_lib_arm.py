#!/usr/bin/env python3
"""One arm of a library-against-library A/B of the CNN kernels on ONE box (the caller alternates processes): BNM_LIBRARY names the
build; per model 2 warm-ups + 9 launches of the default one-kernel form on 10^7 images (median / min), class ids compared with the
four-wave form (variant 6) of the same library."""
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
    out = {"library": os.path.basename(os.environ.get("BNM_LIBRARY", "libbitnetmcu_hip.so"))}
    for name in (sys.argv[1:] or ["cnn_64", "mcu_cnn_48", "mcu_cnn_16"]):
        model = b.Model.from_zoo(name)
        ctx = b.Context(model)
        ctx.set_cnn_variant(3)
        cls, ref = torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
        ms = []
        for k in range(11):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.infer_device(x, cls)
            e1.record()
            torch.cuda.synchronize()
            if k >= 2:
                ms.append(e0.elapsed_time(e1))
        kern = ctx.last_kernel
        ctx.set_cnn_variant(6)
        ctx.infer_device(x, ref)
        torch.cuda.synchronize()
        out[name] = {"kernel": kern, "median_ms": round(float(np.median(ms)), 3), "min_ms": round(float(np.min(ms)), 3),
                     "inf_per_s": round(n / np.median(ms) * 1e3), "differing_class_ids_vs_four_wave_form": int((cls != ref).sum().item())}
        ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
