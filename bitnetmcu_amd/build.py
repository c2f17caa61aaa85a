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


SOURCES = (("bnm_fused_fc.hip", True), ("bnm_fused_regw.hip", True, ("-mllvm", "-amdgpu-mfma-vgpr-form")), ("bnm_fused_generic.hip", True), ("bnm_fused_generic_m2.hip", True),
           ("bnm_fused_generic_m4.hip", True), ("bnm_fused_generic_m6_k8.hip", True), ("bnm_fused_generic_m8_k2.hip", True), ("bnm_fused_generic_m8_k4.hip", True),
           ("bnm_fused_generic_m8_k8.hip", True), ("bnm_fused_generic_m8_k16.hip", True), ("bnm_fused_generic_m2_t2.hip", True), ("bnm_fused_generic_m4_t2.hip", True), ("bnm_cnn.hip", True, ("-mllvm", "-amdgpu-mfma-vgpr-form")), ("bnm_cnn_li.hip", True), ("bnm_cnn_li_fused.hip", True), ("bnm_ternary.hip", True), ("bnm_ternary_s32.hip", True), ("bnm_ternary_s64.hip", True), ("bnm_ternary_s96.hip", True), ("bnm_ternary_s128.hip", True),
           ("bnm_layerwise.hip", True), ("bnm_support.hip", True), ("bnm_qat.hip", True), ("bnm_qat_model.hip", True), ("bnm_qat_cnn.hip", True),
           ("bnm_fused_f32.hip", True), ("bnm_fused_f32_m2.hip", True), ("bnm_fused_f32_m4.hip", True), ("bnm_fused_f32_m6.hip", True),
           ("bnm_capi.cpp", False), ("bnm_capi_model.cpp", False), ("bnm_capi_ctx.cpp", False), ("bnm_capi_infer.cpp", False), ("bnm_capi_host.cpp", False),
           ("bnm_capi_float.cpp", False), ("bnm_capi_qat.cpp", False), ("bnm_capi_multigpu.cpp", False), ("bnm_capi_symbols.cpp", False),
           ("bnm_model.cpp", False))
DIAG_SOURCES = (("bnm_diag.hip", True),)      # diagnostic library only (--diag / --diag-timing)


_TOOL = None


def tool_version():
    """`hipcc --version`, once: a compiler upgrade must rebuild everything."""
    global _TOOL
    if _TOOL is None:
        try:
            _TOOL = subprocess.run([HIPCC, "--version"], capture_output=True, text=True, timeout=120).stdout
        except Exception as e:      # no compiler: the compile step will say so
            _TOOL = f"unknown ({e})"
    return _TOOL


def command_stamp(cmd):
    """What an object was built WITH: the full compile command (flags, per-file flags, architecture, paths) + the compiler's version."""
    import hashlib
    return hashlib.sha1(("\x00".join(cmd) + "\x00" + tool_version()).encode()).hexdigest()


def deps_of(obj):
    """Prerequisites recorded by the compiler (-MD) beside an object, or None when there is no record."""
    d = os.path.splitext(obj)[0] + ".d"
    if not os.path.exists(d):
        return None
    text = open(d).read().replace("\\\n", " ")
    if ":" not in text:
        return None
    # make syntax: prerequisites separated by unescaped blanks, a blank inside a path written as "\ "
    import re
    deps = [t.replace("\\ ", " ") for t in re.split(r"(?<!\\)\s+", text.split(":", 1)[1].strip()) if t]
    return [t for t in deps if not t.startswith("/opt/rocm") and not t.startswith("/usr/")]


def objects(force=False, extra_flags=(), obj_dir=None, sources=SOURCES):
    """Compile what is out of date - a translation unit is rebuilt when one of the files IT includes (the compiler's -MD record)
    is newer than its object, or when the object was built by ANOTHER COMMAND (flags, architecture, compiler version: the stamp
    beside it, <object>.cmd); prints one 'build: compiled|reused <object>' line per translation unit so that a build log
    shows whether the compiler actually ran (BNM_FORCE_BUILD=1 or --force recompiles everything)."""
    obj_dir = obj_dir or OBJ
    force = force or os.environ.get("BNM_FORCE_BUILD") == "1"
    os.makedirs(obj_dir, exist_ok=True)
    out, jobs = [], []
    for entry in sources:
        src, is_hip, file_flags = entry[0], entry[1], (entry[2] if len(entry) > 2 else ())
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        deps = deps_of(o)
        cmd = [HIPCC, f"--offload-arch={ARCH}"] + CXXFLAGS + list(extra_flags) + list(file_flags) + \
              ["-MD", "-MF", os.path.splitext(o)[0] + ".d", "-c", s, "-o", o]
        if not is_hip:
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        stamp_file, stamp = o + ".cmd", command_stamp(cmd)
        same_command = os.path.exists(stamp_file) and open(stamp_file).read().strip() == stamp
        if force or deps is None or not same_command or any(not os.path.exists(d) for d in deps) or newer([s] + deps, o):
            for stale in (o, stamp_file):
                if os.path.exists(stale):
                    os.remove(stale)          # a failed compile must not leave an older object behind for the link step
            jobs.append((cmd, stamp_file, stamp))
            print("build: compiled", os.path.relpath(o, HERE), flush=True)
        else:
            print("build: reused  ", os.path.relpath(o, HERE), flush=True)
        out.append(o)
    def compile_one(job):
        cmd, stamp_file, stamp = job
        run(cmd)
        with open(stamp_file, "w") as f:      # (written only after the compiler succeeded)
            f.write(stamp + "\n")

    if jobs:   # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    return out


def build_lib(force=False):
    objs = objects(force)
    if force or newer(objs, LIB):
        run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs)
    return LIB


def build_diag(timing=False):
    """Diagnostic library (never used by the product, the tests or bench.py's default run): the product sources compiled
    with -DBNM_DIAG — the fused kernels honour bnm_diag_set_src_wrap (cache-resident source) — plus bnm_diag.hip's
    stream / pipe-overlap probes (csrc/bnm_diag.h).  timing=True additionally stamps the shader clock around the dual
    kernel's two vmcnt waits and writes the per-wave sums into the logits buffer (profiles/wait_timing.py)."""
    lib = os.path.join(HERE, "libbitnetmcu_hip_timing.so" if timing else "libbitnetmcu_hip_diag.so")
    flags = ["-DBNM_DIAG"] + (["-DBNM_DIAG_TIMING", "-DBNM_REGW_TIMING"] if timing else [])
    objs = objects(extra_flags=flags, obj_dir=os.path.join(HERE, "_build_timing" if timing else "_build_diag"),
                   sources=SOURCES + DIAG_SOURCES)
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
    ap.add_argument("--diag", action="store_true", help="build libbitnetmcu_hip_diag.so (diagnostics only, csrc/bnm_diag.h)")
    ap.add_argument("--diag-timing", action="store_true", help="build libbitnetmcu_hip_timing.so (diagnostics only)")
    a = ap.parse_args()
    if a.diag or a.diag_timing:
        print(build_diag(timing=a.diag_timing))
    elif a.dll:
        print(build_dll(a.dll, a.outdir))
    else:
        print(build_lib(a.force))


if __name__ == "__main__":
    sys.exit(main())
