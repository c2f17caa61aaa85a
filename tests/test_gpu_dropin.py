"""Drop-in boundary on the GPU box: a model-bound Bitnet_inf.dll built by this repository, loaded and driven
exactly as test_inference.py:134-150 does (CDLL, argtypes=[POINTER(c_int8)], restype=c_uint32, one call per image),
must return the class ids of the reference DLL."""
import os

import numpy as np
import pytest

import util
from util import GOLDEN, REPO
from bitnetmcu_amd import harness, synth, DIST_U, DIST_M

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["fc_4bitsym_64", "cnn_64", "tern_96", "mcu_1k", "mcu_12k_fp130", "doc12k_binary", "doc12k_ternary"])
def test_our_dll_in_the_reference_harness(name, gpu_ok, orc):
    dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", name, "Bitnet_inf.dll")
    if not os.path.isfile(dll):
        pytest.skip(f"{dll} not built (headers under /root/reference are only available in the build container)")
    ours = harness.load_inference_dll(dll)
    model = util.load_golden_model(name)
    om = util.OracleModel(model, orc)
    r = np.load(os.path.join(GOLDEN, "real_images.npz"))
    k = np.load(os.path.join(GOLDEN, f"kat_{name}.npz"))
    # 13 real images: labels for the trained 10-class models, reference class ids for all
    st = harness.cross_check(ours, om.infer, r["images"], r["labels"])
    assert st["mismatch"] == 0
    if not name.startswith(("tern", "doc12k")):       # trained models: every one of the 13 real images is classified correctly
        assert st["correct_c"] == 13
    assert np.array_equal(harness.run_inference_loop(ours, k["images"]), k["cls"])
    # config 1's contract (test_inference.py:136-168 loops all 10,000 test images through lib.Inference; SURVEY.md 8d: 13 real +
    # 10,000 synthetic): 10,000 one-image calls with varying inputs through the latency path's polling protocol
    x = np.concatenate([synth.images(0, 5000, DIST_U), synth.images(0, 5000, DIST_M)])
    st = harness.cross_check(ours, om.infer, x)
    assert st["counter"] == 10000 and st["mismatch"] == 0
    if util.have_ref_dll(name) and not (model.kind == 1 and model.layer(0).out_channels * 4 < 256):
        ref = harness.load_inference_dll(util.ref_dll_path(name))       # the compiled reference, same loop
        assert np.array_equal(harness.run_inference_loop(ref, x), st["result_c"])


@pytest.mark.parametrize("name", ["fc_4bitsym_64", "mcu_1k", "mcu_12k_fp130", "tern_96", "cnn_64", "mcu_cnn_16"])
def test_c_host_calling_the_kernel_symbols_one_by_one(name, gpu_ok, orc, tmp_path):
    """A C program written against BitNetMCU_inference.h - processfclayer / ReLUNorm / processconv33ReLU / processmaxpool22 called
    layer by layer on the arrays of an exporter-dialect BitNetMCU_model.h, the flow of BitNetMCU_MNIST_test.c:43-139 - compiled
    with gcc and linked against the GPU library instead of BitNetMCU_inference.c: prints the reference's lines with the
    oracle's class ids (3- and 4-layer FC, FP130, ternary uint16 rows, 64- and 16-channel CNN with in-place planes)."""
    import subprocess
    model = util.load_golden_model(name)
    r = np.load(os.path.join(GOLDEN, "real_images.npz"))
    images, labels = r["images"][3:13], r["labels"][3:13]
    (tmp_path / "BitNetMCU_model.h").write_text(model.to_header_text())
    (tmp_path / "BitNetMCU_MNIST_test_data.h").write_text(util.test_data_header(images, labels))
    exe = util.compile_c_host("mnist_test.c", tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    want = util.OracleModel(model, orc).infer(images)
    assert out.stdout.splitlines() == [f"label: {int(l)} predicted: {int(c)}" for l, c in zip(labels, want)]
