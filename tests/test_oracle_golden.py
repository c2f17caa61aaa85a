"""The oracle port (oracle/bitnet_oracle.c) against the committed golden vectors, which were produced by
the COMPILED REFERENCE (tests/golden/make_golden.py).  This is what pins the oracle on machines where
/root/reference does not exist."""
import os

import numpy as np
import pytest

import util
from util import GOLDEN, MODEL_NAMES


def test_codec_layers(orc_funcs):
    k = np.load(os.path.join(GOLDEN, "kat_codecs.npz"))
    for c in range(int(k["n_cases"])):
        bpw, n_in, n_out = (int(v) for v in k[f"c{c}_meta"])
        for tag in "sux":
            out = orc_funcs.processfclayer(k[f"c{c}{tag}_act"], k[f"c{c}_w"], bpw, n_in, n_out)
            assert np.array_equal(out, k[f"c{c}{tag}_out"]), (bpw, n_in, n_out, tag)


def test_unknown_codec_yields_zeros(orc_funcs):
    w = np.full(64, 0xFFFFFFFF, np.uint32)
    act = np.full(64, 100, np.int8)
    for bpw in (0, 3, 8, 36, 65):   # 36 = NF4 id: exported but never decoded (exportquant.py:116-120)
        assert not orc_funcs.processfclayer(act, w, bpw, 64, 8).any()


def test_relunorm_edges(orc_funcs):
    k = np.load(os.path.join(GOLDEN, "kat_relunorm.npz"))
    for c in range(int(k["n_cases"])):
        out, pos = orc_funcs.relunorm(k[f"in{c}"])
        assert np.array_equal(out, k[f"out{c}"]) and pos == int(k[f"pos{c}"]), c
        out, pos = orc_funcs.relunorm_inplace(k[f"in{c}"])
        assert np.array_equal(out, k[f"out{c}"]) and pos == int(k[f"pos{c}"]), ("inplace", c)


def test_conv_pool(orc_funcs):
    k = np.load(os.path.join(GOLDEN, "kat_convpool.npz"))
    for c in range(int(k["n_conv"])):
        xy, shift = (int(v) for v in k[f"conv{c}_meta"])
        for inplace in (True, False):
            assert np.array_equal(orc_funcs.conv33(k[f"conv{c}_in"], k[f"conv{c}_w"], xy, shift, inplace), k[f"conv{c}_out"])
    for c in range(int(k["n_pool"])):
        xy = int(k[f"pool{c}_meta"][0])
        for inplace in (True, False):
            assert np.array_equal(orc_funcs.maxpool22(k[f"pool{c}_in"], xy, inplace), k[f"pool{c}_out"])


@pytest.mark.parametrize("name", MODEL_NAMES)
def test_model_kats(name, orc, orc_funcs):
    model = util.load_golden_model(name)
    k = np.load(os.path.join(GOLDEN, f"kat_{name}.npz"))
    x, cls = k["images"], k["cls"]
    om = util.OracleModel(model, orc)
    got_cls, got_logits = om.infer(x, logits=True)
    assert np.array_equal(got_cls, cls)
    tn = int(k["trace_n"])
    assert np.array_equal(got_logits[:tn], k["logits"])
    # per-function schedule through the port's four kernels: every int8 activation
    for i in range(min(tn, 16)):
        c, lg, acts = util.run_schedule(orc_funcs, model, x[i])
        assert c == cls[i] and np.array_equal(lg, k["logits"][i]) and np.array_equal(acts, k["acts"][i])


def test_real_images_labels(orc):
    """The reference's own smoke test: label == prediction for its embedded images
    (BitNetMCU_MNIST_test.c:17-40), for every shipped 10-class model."""
    r = np.load(os.path.join(GOLDEN, "real_images.npz"))
    for name in MODEL_NAMES:
        if name.startswith(("tern", "doc12k")) or name == "mcu_cnn_letters":
            continue   # random-init / 37-class letters model
        om = util.OracleModel(util.load_golden_model(name), orc)
        assert np.array_equal(om.infer(r["images"]), r["labels"]), name


def test_seed_kats_from_survey(orc):
    """values probed from the compiled reference before any of this code existed."""
    r = np.load(os.path.join(GOLDEN, "real_images.npz"))
    om = util.OracleModel(util.load_golden_model("fc_4bitsym_64"), orc)
    cls, lg = om.infer(r["images"][:2], logits=True)
    assert lg[0].tolist() == [-3277, -1343, -1315, 2957, -2401, -685, -3871, -929, -1771, 177] and cls[0] == 3
    assert lg[1].tolist() == [-1239, -1467, 2861, -861, -277, -1797, -1653, -439, -1535, -1483] and cls[1] == 2
