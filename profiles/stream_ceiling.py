#!/usr/bin/env python3
"""Practical read ceiling of the image stream on this GPU, next to the real kernel (run on the GPU box).
Prints GB/s (256 B/image read; the tile-loop modes also write 4 B/image like the real kernel).
Modes 3/4 add a synthetic compute load of the real kernel's size to the LDS-DMA loop / to plain VGPR loads."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bitnetmcu_amd as b  # noqa: E402
from bitnetmcu_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
lib = b.load()
if not hasattr(lib, "bnm_diag_stream_device"):
    sys.exit("needs the diagnostic library: python bitnetmcu_amd/build.py --diag; BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so")
imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
b.synth.fill_device(imgs)
out = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
res = {}
for mode, grids in ((0, (0, 8192)), (1, (0,)), (2, (0,)), (3, (0,)), (4, (0,))):
    for g in grids:
        for _ in range(3):
            L.check(lib, lib.bnm_diag_stream_device(imgs.data_ptr(), n, mode, g, out.data_ptr(), s))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.check(lib, lib.bnm_diag_stream_device(imgs.data_ptr(), n, mode, g, out.data_ptr(), s))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        bytes_ = n * (256 if mode == 0 else 260)
        res[f"mode{mode}_grid{g}"] = {"ms": ms, "GB/s": bytes_ / ms / 1e6}
print(json.dumps(res, indent=1))
