#!/usr/bin/env python3
"""Where does a wave of cnn_front_mfma_kernel wait?  (round 2)

Needs the diagnostic build:   python bitnetmcu_amd/build.py --diag-timing
Run on the GPU box:           BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_timing.so python profiles/cnn_wait_timing.py

In that build the kernel brackets three explicit waits with s_memtime stamps — the item's head loads (20 weight dwords + the
first patch tile), the patch tile of each later MFMA, the partner exchange (4 ds_bpermute) — and writes per wave
{loop cycles, head, tiles, exchange, items}.  Prints medians over the waves and shares of the loop."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bitnetmcu_amd as b
from bitnetmcu_amd import synth
import util


def main():
    if "timing" not in os.environ.get("BNM_LIBRARY", ""):
        sys.exit("set BNM_LIBRARY to the --diag-timing build (see the docstring)")
    n = int(os.environ.get("N", 1 << 20))
    lib = b.load()
    model = util.load_golden_model(os.environ.get("MODEL", "cnn_64"))
    ctx = b.Context(model)
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    synth.fill_device(imgs, first=0, dist=b.DIST_U)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    waves = 4 * 4 * torch.cuda.get_device_properties(0).multi_processor_count
    rec = torch.zeros((waves + 8, 8), dtype=torch.int64, device="cuda")
    for _ in range(2):
        ctx.infer_device(imgs, cls, None)
    torch.cuda.synchronize()
    assert lib.bnm_diag_cnn_set_record(C.c_void_p(rec.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.infer_device(imgs, cls, None)
    e1.record()
    torch.cuda.synchronize()
    lib.bnm_diag_cnn_set_record(C.c_void_p(0))
    r = rec.cpu().numpy().astype(np.float64)
    r = r[r[:, 4] > 1]
    loop, head, tile, xchg, items = (r[:, i] for i in range(5))
    start, xcc, hwid = r[:, 5], r[:, 6].astype(np.int64) & 15, r[:, 7].astype(np.int64)
    cu, se, simd = (hwid >> 8) & 15, (hwid >> 13) & 7, (hwid >> 4) & 3
    ms = e0.elapsed_time(e1)
    out = {"ms_front_plus_tail": ms, "waves": int(len(r)), "items_per_wave_median": float(np.median(items)),
           "loop_cycles_median": float(np.median(loop)), "loop_cycles_min": float(loop.min()), "loop_cycles_max": float(loop.max()),
           "cycles_per_item_median": float(np.median(loop / items)),
           "head_wait_cycles_per_item": float(np.median(head / items)),
           "tile_wait_cycles_per_item": float(np.median(tile / items)),
           "exchange_cycles_per_item": float(np.median(xchg / items)),
           "head_share": float(np.median(head / loop)), "tile_share": float(np.median(tile / loop)),
           "exchange_share": float(np.median(xchg / loop)),
           "loop_cycles_percentiles_1_10_25_50_75_90_99": [float(x) for x in np.percentile(loop, [1, 10, 25, 50, 75, 90, 99])],
           "start_stamp_spread_cycles": float(start.max() - start.min()),
           "end_stamp_percentiles_50_90_99_100_rel_first_start": [float(x) for x in np.percentile(start + loop - start.min(), [50, 90, 99, 100])],
           "loop_cycles_median_by_xcc": {int(x): float(np.median(loop[xcc == x])) for x in np.unique(xcc)},
           "waves_by_xcc": {int(x): int((xcc == x).sum()) for x in np.unique(xcc)},
           "loop_cycles_median_by_simd": {int(x): float(np.median(loop[simd == x])) for x in np.unique(simd)}}
    # waves per (xcc, se, cu): how many of the 16 slots each CU got
    key = xcc * 1000 + se * 100 + cu
    cnt = np.bincount(np.unique(key, return_inverse=True)[1])
    out["waves_per_cu_histogram"] = {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))}
    med_by_cu = np.array([np.median(loop[key == k]) for k in np.unique(key)])
    out["loop_median_by_cu_vs_waves_on_cu"] = {int(c): float(np.median(med_by_cu[cnt == c])) for c in np.unique(cnt)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
