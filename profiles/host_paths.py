#!/usr/bin/env python3
"""Rates of the HOST-pointer entry points (never bench.py's `value`): PCIe-inclusive batched bnm_infer_host and the
per-image Inference() call of the drop-in DLL.  Run on the GPU box; prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bitnetmcu_amd as b  # noqa: E402
import util  # noqa: E402

model = util.load_golden_model("fc_4bitsym_64")
ctx = b.Context(model)
x = b.synth.images(0, 4_000_000)
ctx.infer(x[:1000])
t0 = time.perf_counter()
cls = ctx.infer(x)
t1 = time.perf_counter()
res = {"bnm_infer_host_inf_per_s": len(x) / (t1 - t0), "bnm_infer_host_GBps_h2d": len(x) * 256 / (t1 - t0) / 1e9}
dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", "fc_4bitsym_64", "Bitnet_inf.dll")
if os.path.isfile(dll):
    lib = b.harness.load_inference_dll(dll)
    b.harness.run_inference_loop(lib, x[:200])
    t0 = time.perf_counter()
    out = b.harness.run_inference_loop(lib, x[:5000])
    t1 = time.perf_counter()
    assert np.array_equal(out, cls[:5000])
    res["Inference_us_per_call_incl_ctypes"] = (t1 - t0) / 5000 * 1e6
print(json.dumps(res))
