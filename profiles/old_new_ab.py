#!/usr/bin/env python3
"""One arm of an old-library / new-library A/B on ONE box (profiles/r03/r03_old_new_ab.sh alternates processes): loads the library named
by BNM_AB_LIBRARY (bound non-strictly: an older build lacks the newest symbols), fills 1e8 images, and times the default kernel
of the headline model (ids; ids + logits), the generic kernel (variant 4) and the default kernel of every zoo model named in
BNM_AB_MODELS - 3 warm-ups, 15 launches, median / min."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from bitnetmcu_amd import _lib as L  # noqa: E402

path = os.environ["BNM_AB_LIBRARY"]
L._lib = L.bind(C.CDLL(path), strict=False)
import bitnetmcu_amd as b  # noqa: E402


def main():
    n = 100_000_000
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    lg = torch.empty((n, 10), dtype=torch.int32, device="cuda")
    out = {"library": os.path.basename(path)}
    arms = [("dual_ids", "fc_4bitsym_64", -1, None), ("dual_ids_logits", "fc_4bitsym_64", -1, lg), ("generic_ids", "fc_4bitsym_64", 4, None)]
    for extra in [m for m in os.environ.get("BNM_AB_MODELS", "").split(",") if m]:
        arms.append((extra, extra, -1, None))
    for label, name, variant, logits in arms:
        model = b.Model.from_zoo(name)
        ctx = b.Context(model)
        if variant >= 0:
            ctx.set_tuning(variant=variant)
        for _ in range(3):
            ctx.infer_device(x, cls, logits)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(16)]
        ev[0].record()
        for k in range(15):
            ctx.infer_device(x, cls, logits)
            ev[k + 1].record()
        torch.cuda.synchronize()
        ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(15)]
        out[label] = {"median_ms": float(np.median(ms)), "min_ms": float(np.min(ms))}
        d = b.synth.digest_device(cls, 0, model.num_classes).cpu().numpy()
        out[label]["digest"] = hex(int(d[0].astype(np.uint64)))
        ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
