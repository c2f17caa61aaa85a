#!/usr/bin/env python3
"""Per-image `Inference()` latency under the reference's own loop (test_inference.py:136-168, torchvision-free: the same ctypes call
per image), three ways, each in a process of its own: the reference's compiled DLL (gcc -O3, this host's CPU), the product's drop-in
DLL with a launch per call, the product's drop-in DLL with the resident kernel (BNM_PERSISTENT=1).  10,000 MNIST-like images.
Run on the GPU box; prints one JSON object."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, {repo!r})
import bitnetmcu_amd as b
lib = b.harness.load_inference_dll({dll!r})
x = b.synth.images(0, 10000, b.DIST_M)
b.harness.run_inference_loop(lib, x[:500])
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    out = b.harness.run_inference_loop(lib, x)
    best = min(best, time.perf_counter() - t0)
# the ctypes loop's own cost: the same loop around a C function that does nothing
import ctypes
libc = ctypes.CDLL(None)
libc.abs.argtypes = [ctypes.POINTER(ctypes.c_int8)]
t0 = time.perf_counter()
for row in x:
    ptr = (ctypes.c_int8 * 256)(*row.tolist())
    libc.abs(ptr)
loop = time.perf_counter() - t0
print(json.dumps({{"us_per_call": best / len(x) * 1e6, "loop_overhead_us": loop / len(x) * 1e6, "digest": int(np.bitwise_xor.reduce(out * np.arange(1, len(out) + 1, dtype=np.uint32)))}}))
"""


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "fc_4bitsym_64"
    runs = {"reference_dll_cpu": (os.path.join(REPO, "oracle", "_ref", model, "Bitnet_inf_O3.dll"), "0"),
            "product_dll_launch_per_call": (os.path.join(REPO, "bitnetmcu_amd", "dlls", model, "Bitnet_inf.dll"), "0"),
            "product_dll_resident_kernel": (os.path.join(REPO, "bitnetmcu_amd", "dlls", model, "Bitnet_inf.dll"), "1")}
    res = {"model": model}
    for name, (dll, flag) in runs.items():
        if not os.path.isfile(dll):
            res[name] = "missing " + dll
            continue
        r = subprocess.run([sys.executable, "-c", DRIVER.format(repo=REPO, dll=dll)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, BNM_PERSISTENT=flag))
        res[name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"rc": r.returncode, "stderr": r.stderr[-500:]}
    digests = {v["digest"] for v in res.values() if isinstance(v, dict) and "digest" in v}
    res["same_class_ids"] = len(digests) == 1
    print(json.dumps(res))


if __name__ == "__main__":
    main()
