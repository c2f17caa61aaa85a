"""Live pin of the oracle port against the compiled reference (oracle/_ref/<model>/Bitnet_inf.dll, built from
the unmodified sources under /root/reference by oracle/build_oracle.py).  Skipped where the DLLs are absent."""
import ctypes as C

import numpy as np
import pytest

import util
from util import MODEL_NAMES
from bitnetmcu_amd import synth, DIST_U, DIST_M


@pytest.mark.parametrize("name", MODEL_NAMES)
def test_port_equals_reference_dll(name, orc):
    if not util.have_ref_dll(name):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    model = util.load_golden_model(name)
    if model.kind == 1 and model.layer(0).out_channels * 4 < 256:
        pytest.skip("reference wrapper overflows its stack buffer for this header (see tests/golden/make_golden.py)")
    dll = C.CDLL(util.ref_dll_path(name))
    dll.Inference.restype = C.c_uint32
    dll.Inference.argtypes = [C.POINTER(C.c_int8)]
    n = 3000 if model.kind == 0 else 600
    x = np.concatenate([synth.images(1000, n, DIST_U), synth.images(5000, n, DIST_M)])
    ref = np.array([dll.Inference(r.ctypes.data_as(C.POINTER(C.c_int8))) for r in x], np.uint32)
    got = util.OracleModel(model, orc).infer(x)
    assert np.array_equal(ref, got)


def test_port_kernels_equal_reference_kernels_random(orc_funcs):
    if not util.have_ref_dll("cnn_64"):
        pytest.skip("oracle/_ref not built")
    ref = util.Funcs(C.CDLL(util.ref_dll_path("cnn_64")))
    rng = np.random.default_rng(7)
    for _ in range(60):
        bpw = int(rng.choice([1, 2, 4, 12, 16, 20, 64, 36]))
        n_out = int(rng.integers(1, 70))
        if bpw == 64:
            n_in = 10 * int(rng.integers(1, 27))
            w = rng.integers(0, 65536, size=n_out * (n_in // 10), dtype=np.uint16)
        else:
            n_in = 32 * int(rng.integers(1, 9))
            fb = {1: 1, 2: 2, 4: 4, 12: 4, 20: 4, 16: 8, 36: 4}[bpw]
            w = rng.integers(0, 2**32, size=n_out * (n_in * fb // 32), dtype=np.uint32)
        act = rng.integers(-128, 128, size=n_in, dtype=np.int8)
        assert np.array_equal(ref.processfclayer(act, w, bpw, n_in, n_out), orc_funcs.processfclayer(act, w, bpw, n_in, n_out))
        v = (rng.integers(-2**20, 2**20, size=int(rng.integers(1, 300))) >> int(rng.integers(0, 20))).astype(np.int32)
        assert ref.relunorm(v)[1] == orc_funcs.relunorm(v)[1] and np.array_equal(ref.relunorm(v)[0], orc_funcs.relunorm(v)[0])
