"""the float path says when its input was out of contract.  An image that holds a NaN or an infinity is outside
the reference's contract (test_inference.py:140-141 / BitNetMCU.py:435-436: numpy's cast of NaN is platform-defined); on x86 the
reference's expression turns such an image into ALL ZEROS.  Every float path of the product does the same - the fused FC kernel,
the one-kernel CNN in its float form, bnm_quantize_input_device and the two-kernel forms built on it - and COUNTS the image
(bnm_ctx_float_nonfinite, bnm_quantize_input_counted_device).  Finite images - denormal-only and all-zero ones included - are
bit-identical to numpy's float32 quantisation + the oracle and are not counted."""
import ctypes as C

import numpy as np
import pytest
import torch

import bitnetmcu_amd as b
import util
from bitnetmcu_amd import harness

pytestmark = pytest.mark.gpu


def images(n, seed):
    """finite random rows with the out-of-contract and the edge rows spliced in -> (x, is_nonfinite)"""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n, 256)) * rng.uniform(0.01, 3.0, (n, 1))).astype(np.float32)
    bad = np.zeros(n, bool)
    nan_payload = np.frombuffer(np.uint32(0x7FC012AB).tobytes(), np.float32)[0]      # a NaN whose low byte is not zero
    neg_nan = np.frombuffer(np.uint32(0xFFC00001).tobytes(), np.float32)[0]
    special = {
        3: ("nan", lambda r: r.__setitem__(17, np.nan)),
        4: ("inf", lambda r: r.__setitem__(200, np.inf)),
        5: ("-inf", lambda r: r.__setitem__(0, -np.inf)),
        6: ("nan payload, last pixel", lambda r: r.__setitem__(255, nan_payload)),
        7: ("all nan", lambda r: r.fill(np.nan)),
        9: ("inf and -inf and nan", lambda r: (r.__setitem__(1, np.inf), r.__setitem__(2, -np.inf), r.__setitem__(3, neg_nan))),
        40: ("nan in the upper half of the row", lambda r: r.__setitem__(128 + 77, np.nan)),
    }
    finite_edges = {
        10: lambda r: r.fill(0.0),                                    # all zero: scale 127 / 1e-5, every byte 0 - finite, not counted
        11: lambda r: r.__imul__(np.float32(1e-42)),                  # denormals only
        12: lambda r: r.__setitem__(slice(None), np.float32(3.0e38) * np.sign(r)),      # near FLT_MAX
        13: lambda r: r.__setitem__(5, np.float32(1e-45)),            # one denormal among ordinary values
    }
    for i, (_, f) in special.items():
        for row in (i, i + 64, i + 32 * 37):      # in the first tile, in a later group, in a far tile
            if row < n:
                f(x[row])
                bad[row] = True
    for i, f in finite_edges.items():
        if i < n:
            f(x[i])
    if n > 500:
        x[n - 1, 31] = np.nan                      # the last image of a ragged tile
        bad[n - 1] = True
    return x, bad


def expected(model, orc, x, bad):
    q = harness.quantize_input(np.where(bad[:, None], np.float32(0.0), x))      # non-finite images: the all-zero image
    assert (q[bad] == 0).all()
    return q, util.OracleModel(model, orc).infer(q, logits=True)


@pytest.mark.parametrize("name", ["fc_4bitsym_64", "tern_96", "doc12k_binary"])
def test_fc_float_paths_zero_and_count_nonfinite_images(name, gpu_ok, orc):
    model = util.load_golden_model(name)
    n = 3001
    x, bad = images(n, 11)
    q, (want_cls, want_lg) = expected(model, orc, x, bad)
    xd = torch.from_numpy(x).cuda()
    ctx = b.Context(model)
    assert ctx.float_nonfinite == 0
    seen = 0
    for mode in (0, 2):                      # one kernel; quantise + infer
        ctx.set_float_mode(mode)
        for m in (n, 1000, 33, 8):
            cls = torch.full((m,), -1, dtype=torch.int32, device="cuda")
            lg = torch.full((m, model.num_classes), -1, dtype=torch.int32, device="cuda")
            ctx.infer_float_device(xd[:m], cls, lg)
            seen += int(bad[:m].sum())
            assert ctx.float_nonfinite == seen, (name, mode, m)
            assert np.array_equal(cls.cpu().numpy().astype(np.uint32), want_cls[:m]), (name, mode, m)
            assert np.array_equal(lg.cpu().numpy(), want_lg[:m]), (name, mode, m)
    # a batch of finite images leaves the count where it was
    ok_rows = torch.from_numpy(np.ascontiguousarray(x[~bad][:1024])).cuda()
    ctx.set_float_mode(0)
    ctx.infer_float_device(ok_rows, torch.empty(1024, dtype=torch.int32, device="cuda"))
    assert ctx.float_nonfinite == seen
    # the quantisation kernel on its own: the same bytes, zeros for the out-of-contract images, its own counter
    out = ctx.quantize_device(xd)
    assert np.array_equal(out.cpu().numpy(), q)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    out2 = torch.empty_like(out)
    lib = b.load()
    for _ in range(2):
        rc = lib.bnm_quantize_input_counted_device(C.c_void_p(xd.data_ptr()), n, C.c_void_p(out2.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    assert int(cnt.item()) == 2 * int(bad.sum()) and torch.equal(out, out2)
    ctx.close()


@pytest.mark.parametrize("name", ["cnn_64", "mcu_cnn_16"])
def test_cnn_float_paths_zero_and_count_nonfinite_images(name, gpu_ok, orc):
    model = util.load_golden_model(name)
    n = 2000
    x, bad = images(n, 12)
    _, (want_cls, want_lg) = expected(model, orc, x, bad)
    xd = torch.from_numpy(x).cuda()
    ctx = b.Context(model)
    ctx.set_cnn_variant(3)                   # the one-kernel form for every call size
    seen = 0
    for mode in (0, 2):
        ctx.set_float_mode(mode)
        for m in (n, 999, 33, 5):
            cls = torch.full((m,), -1, dtype=torch.int32, device="cuda")
            lg = torch.full((m, model.num_classes), -1, dtype=torch.int32, device="cuda")
            ctx.infer_float_device(xd[:m], cls, lg)
            seen += int(bad[:m].sum())
            assert ctx.float_nonfinite == seen, (name, mode, m, ctx.last_kernel)
            assert ("<float>" in ctx.last_kernel) == (mode == 0)
            assert np.array_equal(cls.cpu().numpy().astype(np.uint32), want_cls[:m]), (name, mode, m)
            assert np.array_equal(lg.cpu().numpy(), want_lg[:m]), (name, mode, m)
    ctx.close()


def test_numpy_on_this_host_agrees_for_nonfinite_images():
    """What the doc comment claims about the reference's expression on x86: an image with a NaN or an infinity becomes all zeros.
    (Not a contract of the reference - a platform fact the canonical form was chosen to match; skipped where numpy says otherwise.)"""
    x, bad = images(200, 13)
    with np.errstate(all="ignore"):
        q = harness.quantize_input(x)
    if not (q[bad] == 0).all():
        pytest.skip("this host's numpy casts NaN to something else than 0")
    assert (q[bad] == 0).all()
