#!/usr/bin/env python3
"""Rates of the HOST-pointer entry points (never bench.py's `value`): PCIe-inclusive batched bnm_infer_host (pipelined
page-locked staging with 4/8/16 copy threads vs the HIP runtime's pageable copies) and the per-image Inference() call of the
drop-in DLL (polling the page-locked result vs waiting for the stream).  Run on the GPU box; prints one JSON object."""
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
n = int(os.environ.get("N", 8_000_000))
x = b.synth.images(0, n)
want = None
res = {"images": n, "host_cores": len(os.sched_getaffinity(0))}
for label, mode, threads in (("pipelined_8_threads", 0, 8), ("pipelined_4_threads", 0, 4), ("pipelined_16_threads", 0, 16),
                             ("pipelined_12_threads", 0, 12), ("hip_runtime_pageable", 1, 0)):
    ctx.set_host_tuning(mode, threads, True)
    ctx.infer(x[:300000])
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        cls = ctx.infer(x)
        best = min(best, time.perf_counter() - t0)
    if want is None:
        want = cls
    assert np.array_equal(cls, want)
    res[label] = {"inf_per_s": n / best, "GBps_h2d": n * 256 / best / 1e9}
ctx.set_host_tuning(0, 0, True)
# single image through the additive API (no ctypes loop overhead of the DLL harness besides one call)
one = x[:1].copy()
for spin in (True, False):
    ctx.set_host_tuning(0, 0, spin)
    for _ in range(200):
        ctx.infer(one)
    t0 = time.perf_counter()
    for _ in range(3000):
        ctx.infer(one)
    res[f"bnm_infer_host_1_image_us_{'poll' if spin else 'stream_wait'}"] = (time.perf_counter() - t0) / 3000 * 1e6
# ... and through the resident kernel (opt-in): the same call as a mailbox message; what the wave itself spent on it
import ctypes as C
ctx.set_persistent(True)
for _ in range(500):
    ctx.infer(one)
t0 = time.perf_counter()
for _ in range(3000):
    ctx.infer(one)
res["bnm_infer_host_1_image_us_resident_kernel"] = (time.perf_counter() - t0) / 3000 * 1e6
w, sc = C.c_uint32(), C.c_uint32()
b.load().bnm_ctx_persistent_last_call(ctx._h, C.byref(w), C.byref(sc))
res["resident_kernel_inside_the_wave"] = {"us": w.value / 100.0, "shader_clocks": sc.value, "shader_MHz": sc.value / max(w.value, 1) * 100.0}
ctx.set_persistent(False)
dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", "fc_4bitsym_64", "Bitnet_inf.dll")
if os.path.isfile(dll):
    lib = b.harness.load_inference_dll(dll)
    b.harness.run_inference_loop(lib, x[:200])
    t0 = time.perf_counter()
    out = b.harness.run_inference_loop(lib, x[:5000])
    t1 = time.perf_counter()
    assert np.array_equal(out, want[:5000])
    res["Inference_us_per_call_incl_ctypes"] = (t1 - t0) / 5000 * 1e6
    # the reference's own DLL under the same loop, same host (the number the drop-in is compared with)
    ref = os.path.join(REPO, "oracle", "_ref", "fc_4bitsym_64", "Bitnet_inf_O3.dll")
    if os.path.isfile(ref):
        rl = b.harness.load_inference_dll(ref)
        b.harness.run_inference_loop(rl, x[:200])
        t0 = time.perf_counter()
        out2 = b.harness.run_inference_loop(rl, x[:5000])
        res["reference_cpu_Inference_us_per_call_incl_ctypes"] = (time.perf_counter() - t0) / 5000 * 1e6
        assert np.array_equal(out2, out)
print(json.dumps(res))
