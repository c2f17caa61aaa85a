"""QAT forward op on the GPU: csrc/bnm_qat.hip through the C ABI against fixtures generated
from the reference's own BitLinear (tests/golden/make_qat_golden.py).

This is a FLOATING-POINT kernel, so parity is within tolerance, and the tolerances are:
  * activation integers x_int and weight levels w_int: equal to the reference's, except where the value being rounded
    lies within 2e-3 of a rounding tie (the GPU's reductions sum in a different order, so scales can differ by an ulp);
    a mismatch there is exactly one quantisation step;
  * x_scale: relative 1e-6;
  * y against an exact float64 evaluation of the kernel's OWN integers and scales: 1e-5 relative to max|y| (for the
    integer / half-integer QuantTypes the matrix-core sums are exact; '4bit' and 'NF4' carry fp32 GEMM rounding: 5e-5);
  * y against the reference's y on rows / outputs untouched by a tie flip: 2e-4 relative to max|y| (the reference's
    own fp32 GEMM rounding);
  * straight-through gradients: 1e-3 relative to the largest gradient entry.
"""
import os

import numpy as np
import pytest
import torch

import bitnetmcu_amd as b
from bitnetmcu_amd import qat
from util import GOLDEN

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "qat_bitlinear.npz"))
QUANTS = ["Binary", "BinarySym", "Ternary", "2bitsym", "4bit", "4bitsym", "FP130", "NF4", "5bitsym", "8bit"]
NORMS = ["RMS", "Lin", "BatchNorm", "LayerNorm"]
TIE = 2e-3


def dev(name):
    return torch.from_numpy(G[name]).cuda()


def near_tie(v):
    """distance of v from the nearest x.5"""
    return np.abs(np.abs(v - np.floor(v)) - 0.5)


def gpu_levels(d, k):
    """the weight levels the kernel actually used: workspace starts with uT [dpad][kpad]"""
    ws = qat._workspaces[(torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, d, k)]
    dpad, kpad = (d + 1) & ~1, (k + 31) & ~31
    return ws[: dpad * kpad].reshape(dpad, kpad)[:d, :k].t().cpu().numpy().astype(np.float64)


def check_case(tag, qt, nt):
    x, w = dev(f"{tag}/x"), dev(f"{tag}/w")
    n, d = x.shape
    k = w.shape[0]
    s = dev(f"{tag}/{qt}/s")
    y, xi, xs = qat.bitlinear_forward(x, w, s, qt, nt, return_int=True)
    torch.cuda.synchronize()
    y, xi, xs = y.cpu().numpy(), xi.cpu().numpy().astype(np.float64), xs.cpu().numpy().astype(np.float64)
    u = gpu_levels(d, k)

    # activation side
    if f"{tag}/{nt}/x_int" in G:
        ref_xi, ref_xs, ref_xn = G[f"{tag}/{nt}/x_int"], G[f"{tag}/{nt}/x_scale"], G[f"{tag}/{nt}/x_norm"]
        assert np.allclose(xs, ref_xs, rtol=1e-6, atol=0), (qt, nt)
        bad = xi != ref_xi
        if bad.any():
            assert (np.abs(xi - ref_xi)[bad] == 1).all(), (qt, nt)
            assert (near_tie(ref_xn.astype(np.float64) * ref_xs[:, None])[bad] < TIE).all(), (qt, nt)
            assert bad.mean() < 2e-3, (qt, nt, bad.mean())
        clean_rows = ~bad.any(axis=1)
    else:
        clean_rows = np.ones(n, bool)
    # weight side
    ref_u, ref_ws = G[f"{tag}/{qt}/w_int"].astype(np.float64), G[f"{tag}/{qt}/w_scale"].astype(np.float64)
    wbad = np.abs(u - ref_u) > 1e-6
    if wbad.any():
        assert wbad.mean() < 2e-3, (qt, nt, wbad.mean())
        if qt not in ("FP130", "NF4", "Binary"):      # uniform grids: a flip is one step at a rounding tie
            assert (np.abs(u - ref_u)[wbad] <= 1.0 + 1e-6).all(), (qt, nt)
    clean_outs = ~wbad.any(axis=1)
    # y against the kernel's own integers, exactly
    wsc = ref_ws if ref_ws.size == k else np.full(k, ref_ws[0])
    exact = (xi @ u.T) / xs[:, None] / wsc[None, :]
    scale = np.abs(exact).max()
    tol = 5e-5 if qt in ("4bit", "NF4") else 1e-5       # non-integer levels: fp32 GEMM rounding
    assert np.abs(y - exact).max() <= tol * scale, (qt, nt, np.abs(y - exact).max() / scale)
    # y against the reference's y where no tie flipped
    ref_y = G[f"{tag}/{qt}/{nt}/y"]
    sel = np.ix_(clean_rows, clean_outs)
    assert clean_rows.sum() >= n - 3 and clean_outs.sum() >= k - 3, (qt, nt)
    assert np.abs(y[sel] - ref_y[sel]).max() <= 2e-4 * np.abs(ref_y).max(), (qt, nt)


@pytest.mark.parametrize("qt", QUANTS)
def test_bitlinear_forward_all_norms(qt, gpu_ok):
    for nt in NORMS:
        check_case("a", qt, nt)


def test_odd_width_and_per_output_scales(gpu_ok):
    for qt in ("4bitsym", "8bit"):
        check_case("odd", qt, "RMS")
    for qt in ("4bitsym", "2bitsym"):
        check_case("perout", qt, "RMS")


def test_quant_type_none_is_a_plain_linear_on_the_normalised_input(gpu_ok):
    x, w = dev("a/x"), dev("a/w")
    for nt in NORMS:
        y = qat.bitlinear_forward(x, w, torch.ones(1), "None", nt)
        want = torch.nn.functional.linear(qat.normalize(x.double(), nt), w.double())
        assert (y.double() - want).abs().max() <= 1e-5 * want.abs().max(), nt


def test_shapes_empty_batch_large_batch_and_leading_dims(gpu_ok):
    w = dev("a/w")
    s = dev("a/4bitsym/s")
    assert qat.bitlinear_forward(torch.empty(0, 202, device="cuda"), w, s, "4bitsym", "RMS").shape == (0, 24)
    x = torch.randn(3, 700, 202, device="cuda")
    y = qat.bitlinear_forward(x, w, s, "4bitsym", "RMS")
    assert y.shape == (3, 700, 24)
    want = qat.ste_formula(x.reshape(-1, 202), w, s[0], "4bitsym", "RMS")
    # torch's own GPU kernels as the fp32 reference here; tie flips are possible, so compare in the mean
    rel = (y.reshape(-1, 24) - want).abs().max(dim=1).values / want.abs().max()
    assert (rel < 2e-4).float().mean() > 0.98 and rel.max() < 2e-2


def test_wide_layer_uses_large_lds_tile(gpu_ok):
    torch.manual_seed(3)
    x, w = torch.randn(70, 1024, device="cuda"), torch.randn(130, 1024, device="cuda") * 0.05
    s = w.abs().mean() / 0.25
    y, xi, xs = qat.bitlinear_forward(x, w, s, "4bitsym", "RMS", return_int=True)
    u, wsc = qat.weight_quant(w, s, "4bitsym")
    exact = (xi.double() @ u.double().t()) / xs.double()[:, None] / float(wsc)
    assert (y.double() - exact).abs().max() <= 1e-5 * exact.abs().max()
    with pytest.raises(Exception):
        qat.bitlinear_forward(torch.randn(4, 1026, device="cuda"), torch.randn(8, 1026, device="cuda"), s, "4bitsym", "RMS")


@pytest.mark.parametrize("qt,nt", [("4bitsym", "RMS"), ("Ternary", "Lin"), ("FP130", "LayerNorm"), ("2bitsym", "BatchNorm")])
def test_module_forward_backward(qt, nt, gpu_ok):
    layer = qat.BitLinear(202, 24, QuantType=qt, NormType=nt).cuda()
    with torch.no_grad():
        layer.weight.copy_(dev("a/w"))
    layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)
    x = dev("a/x").requires_grad_(True)
    y = layer(x)
    (y * dev("a/gy")).sum().backward()
    ref_y, ref_gx, ref_gw = G[f"a/{qt}/{nt}/y"], G[f"a/{qt}/{nt}/gx"], G[f"a/{qt}/{nt}/gw"]
    rel = np.abs(y.detach().cpu().numpy() - ref_y).max(axis=1) / np.abs(ref_y).max()
    assert (rel < 2e-4).mean() > 0.9 and rel.max() < 2e-2
    assert np.abs(x.grad.cpu().numpy() - ref_gx).max() <= 1e-3 * np.abs(ref_gx).max()
    assert np.abs(layer.weight.grad.cpu().numpy() - ref_gw).max() <= 1e-3 * np.abs(ref_gw).max()


# ---- BitConv2d forward: y within 2e-5 of the reference's y relative to max|y| on >= 99.5 % of the outputs; the rest
# ---- may carry one activation tie flip (one quantisation step of one image row: < 2e-2).
CONV_CASES = [("conv1", (1, 16, 1, "8bit", "None", 0)), ("convdw", (16, 16, 16, "8bit", "None", 0)),
              ("convdw_rms", (8, 16, 8, "4bitsym", "RMS", 1)), ("conv1_tern", (1, 12, 1, "Ternary", "RMS", 0))]


@pytest.mark.parametrize("tag,cfg", CONV_CASES)
def test_bitconv2d_forward(tag, cfg, gpu_ok):
    cin, cout, groups, qt, nt, pad = cfg
    y = qat.bitconv2d_forward(dev(f"{tag}/x"), dev(f"{tag}/w"), dev(f"{tag}/s"), qt, nt, 1, (pad, pad), groups)
    ref = G[f"{tag}/y"]
    assert tuple(y.shape) == ref.shape
    rel = np.abs(y.cpu().numpy() - ref) / np.abs(ref).max()
    assert (rel < 2e-5).mean() > 0.995 and rel.max() < 2e-2, (tag, rel.max(), (rel < 2e-5).mean())


GENERAL_CONV = [("g2", 3, 1, 1, 2, "4bitsym", "RMS"), ("full", 3, 1, 0, 1, "8bit", "None"), ("dw_s2", 3, 2, 1, 6, "4bitsym", "RMS"),
                ("g2_s2", 3, 2, 0, 2, "Ternary", "RMS"), ("k5", 5, 1, 2, 1, "2bitsym", "None"), ("k1_s3", 1, 3, 0, 4, "8bit", "RMS")]
GC = np.load(os.path.join(GOLDEN, "qat_bitconv2d_general.npz"))


@pytest.mark.parametrize("tag,ks,stride,pad,groups,qt,nt", GENERAL_CONV)
def test_bitconv2d_general_forward_and_module(tag, ks, stride, pad, groups, qt, nt, gpu_ok):
    """BitConv2d beyond the reference CNN's two layer kinds: groups between 1 and depthwise, stride 2 / 3, 5 x 5 and 1 x 1 kernels,
    a non-square plane - forward against the reference module's y (same tolerance as above), the drop-in module's gradients
    against the reference's."""
    gd = lambda k: torch.from_numpy(GC[k]).cuda()
    y = qat.bitconv2d_forward(gd(f"{tag}/x"), gd(f"{tag}/w"), gd(f"{tag}/s"), qt, nt, stride, (pad, pad), groups)
    ref = GC[f"{tag}/y"]
    assert tuple(y.shape) == ref.shape
    rel = np.abs(y.cpu().numpy() - ref) / np.abs(ref).max()
    assert (rel < 2e-5).mean() > 0.995 and rel.max() < 2e-2, (tag, rel.max(), (rel < 2e-5).mean())
    cout, cpg = GC[f"{tag}/w"].shape[:2]
    layer = qat.BitConv2d(cpg * groups, cout, kernel_size=ks, stride=stride, padding=(pad, pad), groups=groups, QuantType=qt, NormType=nt).cuda()
    with torch.no_grad():
        layer.weight.copy_(gd(f"{tag}/w"))
    layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)
    assert np.allclose(layer.s.detach().cpu().numpy().reshape(-1), GC[f"{tag}/s"], rtol=1e-6)
    x = gd(f"{tag}/x").requires_grad_(True)
    ym = layer(x)
    (ym * gd(f"{tag}/gy")).sum().backward()
    relm = np.abs(ym.detach().cpu().numpy() - ref) / np.abs(ref).max()      # (the module's own s may differ from the fixture's in the last ulp)
    assert (relm < 2e-5).mean() > 0.995 and relm.max() < 2e-2
    assert np.abs(x.grad.cpu().numpy() - GC[f"{tag}/gx"]).max() <= 1e-3 * np.abs(GC[f"{tag}/gx"]).max()
    assert np.abs(layer.weight.grad.cpu().numpy() - GC[f"{tag}/gw"]).max() <= 1e-3 * np.abs(GC[f"{tag}/gw"]).max()


def test_bitconv2d_module_forward_backward_and_refusals(gpu_ok):
    for tag, (cin, cout, groups, qt, nt, pad) in (CONV_CASES[0], CONV_CASES[2]):
        layer = qat.BitConv2d(cin, cout, kernel_size=3, stride=1, padding=(pad, pad), groups=groups, QuantType=qt, NormType=nt).cuda()
        with torch.no_grad():
            layer.weight.copy_(dev(f"{tag}/w"))
        layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)
        assert np.allclose(layer.s.detach().cpu().numpy().reshape(-1), G[f"{tag}/s"], rtol=1e-6)
        x = dev(f"{tag}/x").requires_grad_(True)
        y = layer(x)
        (y * dev(f"{tag}/gy")).sum().backward()
        rel = np.abs(y.detach().cpu().numpy() - G[f"{tag}/y"]) / np.abs(G[f"{tag}/y"]).max()
        assert (rel < 2e-5).mean() > 0.995 and rel.max() < 2e-2
        assert np.abs(x.grad.cpu().numpy() - G[f"{tag}/gx"]).max() <= 1e-3 * np.abs(G[f"{tag}/gx"]).max()
        assert np.abs(layer.weight.grad.cpu().numpy() - G[f"{tag}/gw"]).max() <= 1e-3 * np.abs(G[f"{tag}/gw"]).max()
    x = torch.randn(2, 4, 8, 8, device="cuda")
    with pytest.raises(ValueError):          # weight shape does not match the group structure
        qat.bitconv2d_forward(x, torch.randn(8, 1, 3, 3, device="cuda"), torch.ones(1), "8bit", "None", 1, 0, 2)
    with pytest.raises(NotImplementedError):  # PerOutput clipping scalars
        qat.bitconv2d_forward(x, torch.randn(4, 1, 3, 3, device="cuda"), torch.ones(4), "8bit", "None", 1, 0, 4)
    with pytest.raises(b.BnmError):           # one group's planes do not fit 160 KiB of LDS
        qat.bitconv2d_forward(torch.randn(1, 64, 40, 40, device="cuda"), torch.randn(8, 64, 3, 3, device="cuda"), torch.ones(1), "8bit", "None", 1, 0, 1)
    assert qat.bitconv2d_forward(x[:0], torch.randn(4, 1, 3, 3, device="cuda"), torch.ones(1), "8bit", "None", 1, 0, 4).shape == (0, 4, 6, 6)
