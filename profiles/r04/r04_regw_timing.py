#!/usr/bin/env python3
"""Where does a wave of fused_fc_regw_kernel spend its time?  Runs the timing build of the library (build.py --diag-timing:
-DBNM_REGW_TIMING stamps the shader clock around the loop body's two tile waits and writes per-wave sums into the logits buffer).
Usage (GPU box):  BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_timing.so python profiles/r04_regw_timing.py [model] [images]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bitnetmcu_amd as b
    name = sys.argv[1] if len(sys.argv) > 1 else "tern_96"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    model = b.Model.from_zoo(name)
    ctx = b.Context(model)
    ctx.set_tuning(variant=9)
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    lg = torch.zeros((n, model.num_classes), dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.infer_device(x, cls, lg)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    ctx.infer_device(x, cls, lg)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1])
    rec = lg.view(-1)[: 1024 * 8].cpu().numpy().view(np.uint64).reshape(-1, 4)
    rec = rec[rec[:, 3] > 0]
    tot, wb, wa, it = (rec[:, k].astype(np.float64) for k in range(4))
    out = {"model": name, "images": n, "ms": ms, "waves": int(len(rec)),
           "counter_ticks_per_wave": float(tot.mean()), "ticks_per_second": float(tot.mean() / (ms * 1e-3)),
           "iterations_per_wave": float(it.mean()),
           "ticks_per_iteration": float((tot / it).mean()),
           "wait_tile_B_block0_frac": float((wb / tot).mean()), "wait_tile_A_block1_frac": float((wa / tot).mean()),
           "loop_ticks_min_max": [float(tot.min()), float(tot.max())]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
