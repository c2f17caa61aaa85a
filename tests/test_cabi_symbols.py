"""The C-ABI shared library loads without a GPU and exports every symbol include/bitnetmcu_hip.h declares
(no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import bitnetmcu_amd
from bitnetmcu_amd import _lib
from util import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "bitnetmcu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"BNM_API\s+[\w\s\*]+?\b(\w+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.PROTOTYPES)


def test_library_exports_all_symbols(bnm):
    for name in declared_symbols():
        assert hasattr(bnm, name), name
    assert b"gfx950" in bnm.bnm_version()


def test_reference_symbol_set_of_the_dropin_dll():
    """A model-bound Bitnet_inf.dll exports what the reference DLL exports."""
    dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", "tern_96", "Bitnet_inf.dll")
    if not os.path.isfile(dll):
        pytest.skip("run __graft_entry__.build() first")
    out = subprocess.run(["nm", "-D", "--defined-only", dll], capture_output=True, text=True).stdout
    syms = {l.split()[-1] for l in out.splitlines() if l.strip()}
    for s in ("Inference", "BitMnistInference", "processfclayer", "ReLUNorm", "processconv33ReLU", "processmaxpool22",
              "L1_weights", "L2_weights", "L3_weights", "L4_weights"):
        assert s in syms, s
    lib = C.CDLL(dll)   # loads on a machine without a GPU; calling Inference there aborts by design
    assert lib.Inference is not None


def test_no_gpu_means_error_not_fallback(bnm):
    """On a box without a HIP device the additive API reports BNM_EHIP; it never computes on the CPU."""
    if bnm.bnm_device_count() > 0:
        pytest.skip("a GPU is present")
    import util
    m = util.load_golden_model("mcu_1k")
    with pytest.raises(bitnetmcu_amd.BnmError) as e:
        bitnetmcu_amd.Context(m)
    assert "(-4)" in str(e.value)


def test_qat_entry_points_validate_arguments_before_touching_the_device(bnm):
    """Argument errors of the QAT ops are reported as such on any machine (no GPU needed, no compute happens)."""
    import ctypes as C
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    ws = bnm.bnm_qat_workspace_bytes(256, 64)
    assert ws >= (256 * 64 + 64) * 4
    lin = bnm.bnm_qat_bitlinear_forward_device
    EINVAL, EUNSUP = -1, -3
    assert lin(p, 1, 0, p, 4, p, 1, 6, 0, p, p, ws, None, None, None, None) == EINVAL          # d == 0
    assert lin(p, 1, 2000, p, 4, p, 1, 6, 0, p, p, 1 << 30, None, None, None, None) == EINVAL  # d > 1024
    assert lin(p, 1, 8, p, 4, p, 3, 6, 0, p, p, ws, None, None, None, None) == EINVAL          # s_count not 1 or k
    assert lin(p, 1, 8, p, 4, p, 1, 99, 0, p, p, ws, None, None, None, None) == EINVAL         # unknown QuantType
    assert lin(p, 1, 8, p, 4, p, 1, 6, 9, p, p, ws, None, None, None, None) == EINVAL          # unknown NormType
    assert lin(p, 1, 8, p, 4, p, 1, 6, 0, p, p, 16, None, None, None, None) == EINVAL          # workspace too small
    assert lin(None, 1, 8, p, 4, p, 1, 6, 0, p, p, ws, None, None, None, None) == EINVAL       # null x with n > 0
    assert b"workspace" in bnm.bnm_last_error() or b"null" in bnm.bnm_last_error()
    conv = bnm.bnm_qat_bitconv2d_forward_device
    wsc = bnm.bnm_qat_workspace_bytes(9, 8)
    assert conv(p, 1, 4, 8, 8, p, 8, 3, 3, 0, 1, 3, p, 10, 4, p, p, wsc, None) == EINVAL        # channels not a multiple of groups
    assert conv(p, 1, 1, 2, 2, p, 8, 3, 3, 0, 1, 1, p, 10, 4, p, p, wsc, None) == EINVAL        # kernel larger than the plane
    assert conv(p, 1, 1, 8, 8, p, 8, 3, 3, 0, 1, 1, p, 10, 3, p, p, wsc, None) == EINVAL        # LayerNorm is not a conv NormType
    assert conv(p, 1, 1, 8, 8, p, 8, 3, 3, 0, 0, 1, p, 10, 4, p, p, wsc, None) == EINVAL        # stride 0
    assert conv(p, 1, 1, 300, 300, p, 8, 3, 3, 0, 1, 1, p, 10, 4, p, p, wsc, None) == EUNSUP    # plane exceeds 160 KiB of LDS
    assert conv(p, 1, 4, 8, 8, p, 8, 3, 3, 0, 1, 2, p, 10, 4, p, p, 16, None) == EINVAL         # workspace too small for d = 2 * 9


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/bitnetmcu_hip.h is the drop-in boundary: plain C99 (and C++) with no torch / HIP types, and a C host
    links against the library with nothing but gcc."""
    hdr = os.path.join(REPO, "include")
    src = tmp_path / "hdr.c"
    src.write_text('#include "bitnetmcu_hip.h"\nint main(void) { return sizeof(bnm_layer_info) ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", hdr, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", hdr, str(src)])
    text = open(os.path.join(hdr, "bitnetmcu_hip.h")).read()
    assert "torch" not in text and "hip/hip_runtime" not in text
    exe = tmp_path / "batch_infer"
    libdir = os.path.join(REPO, "bitnetmcu_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", hdr,
                           os.path.join(REPO, "examples", "batch_infer.c"), "-L", libdir, "-lbitnetmcu_hip",
                           f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    # the training forward from C: the whole FC model, the whole CNNMNIST (the GPU runs: tests/test_gpu_qat_model.py, test_gpu_qat_cnn.py)
    for name in ("qat_forward", "qat_cnn_forward"):
        exe = tmp_path / name
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", hdr,
                               os.path.join(REPO, "examples", name + ".c"), "-L", libdir, "-lbitnetmcu_hip",
                               f"-Wl,-rpath,{libdir}", "-o", str(exe)])
        r = subprocess.run([str(exe)], capture_output=True, text=True)
        assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.parametrize("name", ["fc_4bitsym_64", "mcu_1k", "tern_96", "cnn_64"])
def test_layer_by_layer_c_host_compiles_and_refuses_to_run_without_a_gpu(name, tmp_path):
    """examples/mnist_test.c (the call sequence of BitNetMCU_MNIST_test.c:43-139 on the reference's symbols) compiles warning-free
    against exporter-dialect headers of 3- / 4-layer FC, ternary and CNN models; without a HIP device the first kernel symbol
    aborts with a message - there is no CPU fallback (the GPU run: tests/test_gpu_dropin.py)."""
    import numpy as np
    import torch
    import util
    import bitnetmcu_amd as b
    r = np.load(os.path.join(util.GOLDEN, "real_images.npz"))
    (tmp_path / "BitNetMCU_model.h").write_text(b.Model.from_zoo(name).to_header_text())
    (tmp_path / "BitNetMCU_MNIST_test_data.h").write_text(util.test_data_header(r["images"][:10], r["labels"][:10]))
    exe = util.compile_c_host("mnist_test.c", tmp_path)
    if torch.cuda.is_available():
        return
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode != 0 and out.stdout == "" and "no CPU fallback" in out.stderr
