#!/usr/bin/env python3
"""Build the gfx950 native library in-tree.

  python bitnetmcu_amd/build.py                     -> bitnetmcu_amd/libbitnetmcu_hip.so
  python bitnetmcu_amd/build.py --dll <model.h> [-o dir]
                                                    -> <dir>/Bitnet_inf.dll bound to that model
                                                       (the artifact name test_inference.py:134 loads)

Explicit hipcc command lines; no cmake, no JIT cache: the .so travels with the source tree.
"""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libbitnetmcu_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def newer(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in src_list)


def objects(force=False, extra_flags=(), obj_dir=None):
    obj_dir = obj_dir or OBJ
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in ("bnm_kernels.h", "bnm_model.hpp", "bnm_device.hpp")] + \
           [os.path.join(HERE, "..", "include", "bitnetmcu_hip.h")]
    out, jobs = [], []
    for src, is_hip in (("bnm_fused_fc.hip", True), ("bnm_cnn.hip", True), ("bnm_ternary.hip", True), ("bnm_layerwise.hip", True),
                        ("bnm_support.hip", True), ("bnm_qat.hip", True), ("bnm_capi.cpp", False), ("bnm_model.cpp", False)):
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        if force or newer([s] + hdrs, o):
            cmd = [HIPCC, f"--offload-arch={ARCH}"] + CXXFLAGS + list(extra_flags) + ["-c", s, "-o", o]
            if not is_hip:
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            jobs.append(cmd)
        out.append(o)
    if jobs:   # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    return out


def build_lib(force=False):
    objs = objects(force)
    if force or newer(objs, LIB):
        run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs)
    return LIB


def build_diag_timing():
    """Diagnostic library: the dual-tile kernel stamps the shader clock around its two vmcnt waits and writes the
    per-wave sums into the logits buffer (profiles/wait_timing.py).  Never used by the product or the tests."""
    lib = os.path.join(HERE, "libbitnetmcu_hip_timing.so")
    objs = objects(extra_flags=["-DBNM_DIAG_TIMING"], obj_dir=os.path.join(HERE, "_build_timing"))
    run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-o", lib] + objs)
    return lib


def build_dll(header, outdir):
    header = os.path.abspath(header)
    os.makedirs(outdir, exist_ok=True)
    objs = objects()
    stub = os.path.join(outdir, "dll_stub.o")
    run(["gcc", "-O2", "-fPIC", "-w", "-c", os.path.join(CSRC, "dll_stub.c"), "-o", stub,
         f'-DBNM_MODEL_HEADER_PATH="{header}"', "-D_DLL"])
    dll = os.path.join(outdir, "Bitnet_inf.dll")
    run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-o", dll] + objs + [stub])
    os.remove(stub)
    return dll


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dll", metavar="MODEL_H", help="build a model-bound Bitnet_inf.dll from this header")
    ap.add_argument("-o", "--outdir", default=".")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--diag-timing", action="store_true", help="build libbitnetmcu_hip_timing.so (diagnostics only)")
    a = ap.parse_args()
    if a.diag_timing:
        print(build_diag_timing())
    elif a.dll:
        print(build_dll(a.dll, a.outdir))
    else:
        print(build_lib(a.force))


if __name__ == "__main__":
    sys.exit(main())
