#!/usr/bin/env python3
"""Where does a wave of the generic kernel spend a tile?  Needs the diagnostic build (python bitnetmcu_amd/build.py --diag-timing):
    BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_timing.so python profiles/r05/r05_generic_phases.py [model ...]
In that build fused_fc_generic_kernel stamps the shader clock at the phase boundaries of its uniform path and writes per-wave sums
where `logits` point: [0] wait for the tile, [1] layer-1 MFMAs, [2] ReLUNorm 1, [3] layer-2 MFMAs, [4] ReLUNorm 2, [5] layer-3
MFMAs, [6] ReLUNorm 3, [7] classifier MFMAs, [8] argmax, [9] loop overhead; [12] tiles, [13] HW_ID, [14] wave index.
Printed per model and waves per workgroup (BNM_GENERIC_WAVES): median clocks per tile and phase over the waves."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bitnetmcu_amd as b  # noqa: E402

NAMES = ["wait_tile", "L1_mfma", "relunorm1", "L2_mfma", "relunorm2", "L3_mfma", "relunorm3", "L4_mfma", "argmax", "loop"]


def main():
    if "timing" not in os.environ.get("BNM_LIBRARY", ""):
        sys.exit("set BNM_LIBRARY to the --diag-timing build")
    n = int(float(os.environ.get("N", "2e7")))
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    for name in (sys.argv[1:] or ["doc12k_binary"]):
        model = b.Model.from_zoo(name)
        ctx = b.Context(model)
        if name == "fc_4bitsym_64":
            ctx.set_tuning(variant=4)
        rec = torch.zeros((n, model.num_classes), dtype=torch.int32, device="cuda")
        for waves in [int(w) for w in os.environ.get("WAVES", "4,8,0").split(",")]:
            os.environ["BNM_GENERIC_WAVES"] = str(waves)
            for _ in range(2):
                rec.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.infer_device(x, cls, rec)
                e1.record()
                torch.cuda.synchronize()
            r = rec.view(-1)[: 16 * 4096].cpu().numpy().reshape(-1, 16)
            r = r[r[:, 12] > 0]
            per = r[:, :10].astype(np.float64) / r[:, 12:13]
            hw = r[:, 13].astype(np.uint32)
            simd = (hw >> 4) & 3
            slot = hw & 15
            # does wave index w and w + 4 of a workgroup share a SIMD?
            w0 = r[:8, 14], [int(v) for v in simd[:8]], [int(v) for v in slot[:8]]
            out = {"model": name, "waves_per_workgroup": waves, "ms": round(e0.elapsed_time(e1), 3), "waves": int(len(r)),
                   "tiles_per_wave_median": float(np.median(r[:, 12])),
                   "clocks_per_tile": {k: round(float(np.median(per[:, i])), 1) for i, k in enumerate(NAMES)},
                   "sum": round(float(np.median(per.sum(1))), 1),
                   "first_workgroup": {"wave": [int(v) for v in w0[0]], "simd": w0[1], "slot": w0[2]}}
            print(json.dumps(out), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
