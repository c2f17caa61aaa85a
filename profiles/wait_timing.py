#!/usr/bin/env python3
"""How long does a wave of the dual-tile kernel wait for its image tiles?  (round-2 plan, experiment 1)
(BNM_DIAG_SRC_WRAP is honoured by the diagnostic libraries only — build.py --diag / --diag-timing.)

Needs the diagnostic build:   python bitnetmcu_amd/build.py --diag-timing
Run on the GPU box:           BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_timing.so python profiles/wait_timing.py
                              BNM_DIAG_SRC_WRAP=256 BNM_LIBRARY=... python profiles/wait_timing.py     (cache-resident source)

In that build fused_fc_dual_kernel brackets its two `s_waitcnt vmcnt` with s_memtime stamps and writes, per wave,
{cycles in the loop, cycles in wait A, cycles in wait B, iterations} into the buffer the caller passes as `logits`.
Prints one JSON object: medians over the waves and the shares of loop time."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bitnetmcu_amd as b
from bitnetmcu_amd import synth
import util


def main():
    if "timing" not in os.environ.get("BNM_LIBRARY", ""):
        sys.exit("set BNM_LIBRARY to the --diag-timing build (see the docstring)")
    n = int(os.environ.get("N", 100_000_000))
    model = util.load_golden_model("fc_4bitsym_64")
    ctx = b.Context(model)
    ctx.set_tuning(variant=int(os.environ.get("VARIANT", 3)))
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    synth.fill_device(imgs, first=0, dist=b.DIST_U)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    rec = torch.zeros((n, model.num_classes), dtype=torch.int32, device="cuda")      # record array lives at its start
    for _ in range(3):
        ctx.infer_device(imgs, cls, rec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.infer_device(imgs, cls, rec)
    e1.record()
    torch.cuda.synchronize()
    waves = 2 * 4 * torch.cuda.get_device_properties(0).multi_processor_count   # both layouts: 8 waves per CU
    r = rec.view(-1)[: waves * 8].cpu().numpy().view(np.uint64).reshape(waves, 4).astype(np.float64)
    r = r[r[:, 3] > 0]
    loop, wa, wb, it = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    out = {
        "ms": e0.elapsed_time(e1), "waves": int(len(r)), "iterations_per_wave_median": float(np.median(it)),
        "loop_cycles_median": float(np.median(loop)),
        "wait_A_cycles_per_iteration_median": float(np.median(wa / it)),
        "wait_B_cycles_per_iteration_median": float(np.median(wb / it)),
        "wait_A_share_of_loop": float(np.median(wa / loop)), "wait_B_share_of_loop": float(np.median(wb / loop)),
        "wait_A_p95_share": float(np.percentile(wa / loop, 95)), "wait_B_p95_share": float(np.percentile(wb / loop, 95)),
        "loop_cycles_percentiles_1_10_25_50_75_90_99_100": [float(x) for x in np.percentile(loop, [1, 10, 25, 50, 75, 90, 99, 100])],
        "iterations_percentiles_1_50_99": [float(x) for x in np.percentile(it, [1, 50, 99])],
        "shader_clock_GHz": float(np.median(loop) / (e0.elapsed_time(e1) * 1e6)),
        "src_wrap": os.environ.get("BNM_DIAG_SRC_WRAP", "0"),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
