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
