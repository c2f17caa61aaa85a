#!/usr/bin/env python3
"""what does the layer-by-layer symbol flow cost?  examples/mnist_test.c (processfclayer / ReLUNorm /
processconv33ReLU / processmaxpool22 called one by one with host pointers, the flow of BitNetMCU_MNIST_test.c:43-121) linked
against the GPU library, mean time per image, next to Inference() (one call per image) through the model-bound DLL and the
compiled reference on one host core.   usage (GPU box): python profiles/r04_symbol_flow.py"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import util
    import bitnetmcu_amd as b
    from bitnetmcu_amd import harness
    r = np.load(os.path.join(util.GOLDEN, "real_images.npz"))
    images, labels = r["images"][3:13], r["labels"][3:13]
    out = {}
    for name, reps in (("fc_4bitsym_64", 200), ("cnn_64", 20)):
        model = util.load_golden_model(name)
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, "BitNetMCU_model.h"), "w").write(model.to_header_text())
            open(os.path.join(d, "BitNetMCU_MNIST_test_data.h"), "w").write(util.test_data_header(images, labels))
            import pathlib
            exe = util.compile_c_host("mnist_test.c", pathlib.Path(d))
            p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=dict(os.environ, BNM_TIME_REPEATS=str(reps)))
            line = [l for l in p.stderr.splitlines() if l.startswith("symbol flow")]
            row = {"symbol_flow_us_per_image": float(line[0].split()[2]) if line else None, "stderr": p.stderr[-300:] if not line else None}
        # one Inference() call per image through the model-bound DLL (the harness's flow)
        dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", name, "Bitnet_inf.dll")
        if os.path.isfile(dll):
            lib = harness.load_inference_dll(dll)
            ptrs = [(C.c_int8 * 256)(*row_.tolist()) for row_ in images]
            for pz in ptrs:
                lib.Inference(pz)
            t0 = time.perf_counter()
            n = 2000
            for k in range(n):
                lib.Inference(ptrs[k % 10])
            row["inference_call_us_per_image"] = (time.perf_counter() - t0) / n * 1e6
        ref = util.ref_dll_path(name)
        if os.path.isfile(ref):
            lib = harness.load_inference_dll(ref)
            ptrs = [(C.c_int8 * 256)(*row_.tolist()) for row_ in images]
            t0 = time.perf_counter()
            n = 5000
            for k in range(n):
                lib.Inference(ptrs[k % 10])
            row["reference_cpu_us_per_image_one_core"] = (time.perf_counter() - t0) / n * 1e6
        calls = sum(1 for _ in model.fc_layers()) * 2
        if model.kind == b.KIND_CNN:
            calls += model.layer(0).out_channels * 5 + 1
        row["symbol_calls_per_image"] = calls
        if row.get("symbol_flow_us_per_image"):
            row["us_per_symbol_call"] = row["symbol_flow_us_per_image"] / calls
        out[name] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
