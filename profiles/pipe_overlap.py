#!/usr/bin/env python3
"""Do the matrix pipe and the VALU of one SIMD overlap for the fused kernel's instruction mix?  (diag modes 5/6/7)"""
import json, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bitnetmcu_amd as b
from bitnetmcu_amd import _lib as L
lib = b.load()
if not hasattr(lib, "bnm_diag_stream_device"):
    sys.exit("needs the diagnostic library: python bitnetmcu_amd/build.py --diag; BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so")
out = torch.zeros(1024, dtype=torch.int32, device="cuda")
dummy = torch.zeros(256, dtype=torch.int8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
tiles = 1526     # tiles per wave of the 1e8-image benchmark (3.125e6 tiles / 2048 waves)
res = {}
for mode, name in ((5, "mfma_only"), (6, "valu_only"), (7, "both")):
    for _ in range(2):
        L.check(lib, lib.bnm_diag_stream_device(dummy.data_ptr(), tiles, mode, 0, out.data_ptr(), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.check(lib, lib.bnm_diag_stream_device(dummy.data_ptr(), tiles, mode, 0, out.data_ptr(), s))
    e1.record()
    torch.cuda.synchronize()
    res[name + "_ms"] = e0.elapsed_time(e1) / 5
print(json.dumps(res))
