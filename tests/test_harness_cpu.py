"""BASELINE config 1 (plumbing, no GPU): the reference's ctypes harness loop (test_inference.py:134-175) restated
without torchvision, driven against the COMPILED REFERENCE DLL: 13 real images (labels must match) + 10,000
synthetic images, cross-checked against the oracle port."""
import os

import numpy as np
import pytest

import util
from util import GOLDEN
from bitnetmcu_amd import harness, synth, DIST_U, DIST_M


def test_quantize_input_matches_reference_formula():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 256)).astype(np.float32)
    q = harness.quantize_input(x)
    scale = 127.0 / np.maximum(np.abs(x).max(axis=-1, keepdims=True), 1e-5)    # test_inference.py:140
    assert np.array_equal(q, np.round(x * scale).clip(-128, 127).astype(np.int8))
    assert q.dtype == np.int8 and np.abs(q).max() == 127
    assert not harness.quantize_input(np.zeros((1, 256))).any()


@pytest.mark.parametrize("name", ["fc_4bitsym_64", "cnn_64"])
def test_reference_dll_through_harness(name, orc):
    if not util.have_ref_dll(name):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = harness.load_inference_dll(util.ref_dll_path(name))
    r = np.load(os.path.join(GOLDEN, "real_images.npz"))
    om = util.OracleModel(util.load_golden_model(name), orc)
    st = harness.cross_check(lib, om.infer, r["images"], r["labels"])
    assert st["mismatch"] == 0 and st["correct_c"] == 13 and st["counter"] == 13
    n = 10000 if name == "fc_4bitsym_64" else 1500
    x = np.concatenate([synth.images(0, n // 2, DIST_U), synth.images(0, n // 2, DIST_M)])
    st = harness.cross_check(lib, om.infer, x)
    assert st["counter"] == n and st["mismatch"] == 0


def test_the_two_statements_of_the_input_quantisation_agree():
    """oracle/checker.quantize_input (the checker's verbatim restatement of test_inference.py:140-141) and
    bitnetmcu_amd.harness.quantize_input (the product-side host statement) on the edge rows the GPU tests use."""
    import sys
    sys.path.insert(0, os.path.join(util.REPO, "oracle"))
    import checker
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2000, 256)).astype(np.float32)
    x[0] = 0.0
    x[1] = np.linspace(-1, 1, 256, dtype=np.float32)
    x[2, :] = 0.5; x[2, 0] = 127.0
    x[3, :] = -1.5; x[3, 0] = 127.0
    x[4] = rng.integers(-300, 300, 256).astype(np.float32) / 2.0
    x[5] *= np.float32(1e-7)
    x[6] *= np.float32(1e20)
    q = checker.quantize_input(x)
    assert q.dtype == np.int8 and np.array_equal(q, harness.quantize_input(x))
    assert q[0].tolist() == [0] * 256 and q[2, 1] == 0 and q[3, 1] == -2 and int(np.abs(q[1:]).max()) == 127
