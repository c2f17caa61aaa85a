"""QAT forward op, CPU side: the PyTorch restatement that provides the straight-through
backward (bitnetmcu_amd/qat.py) is pinned against fixtures generated from the reference's own BitLinear
(tests/golden/make_qat_golden.py), and the product op refuses CPU tensors instead of falling back."""
import os

import numpy as np
import pytest
import torch

from bitnetmcu_amd import qat
import util
from util import GOLDEN

G = np.load(os.path.join(GOLDEN, "qat_bitlinear.npz"))
QUANTS = ["Binary", "BinarySym", "Ternary", "2bitsym", "4bit", "4bitsym", "FP130", "NF4", "5bitsym", "8bit"]
NORMS = ["RMS", "Lin", "BatchNorm", "LayerNorm"]


def t(name):
    return torch.from_numpy(G[name])


@pytest.mark.parametrize("nt", NORMS)
def test_normalize_and_activation_quant_equal_reference(nt):
    x = t("a/x")
    xn = qat.normalize(x, nt)
    assert torch.equal(xn, t(f"a/{nt}/x_norm"))
    xi, xs = qat.activation_quant(xn)
    assert torch.equal(xi, t(f"a/{nt}/x_int")) and torch.equal(xs.reshape(-1), t(f"a/{nt}/x_scale"))


@pytest.mark.parametrize("qt", QUANTS)
def test_weight_quant_and_forward_equal_reference(qt):
    x, w = t("a/x"), t("a/w")
    for nt in NORMS:
        s = t(f"a/{qt}/s")
        s = s[0] if s.numel() == 1 else s
        u, sc = qat.weight_quant(w, s, qt)
        assert torch.equal(u, t(f"a/{qt}/w_int"))
        assert torch.equal(torch.as_tensor(sc).reshape(-1), t(f"a/{qt}/w_scale"))
        assert torch.equal(qat.ste_formula(x, w, s, qt, nt), t(f"a/{qt}/{nt}/y"))


def test_per_output_scale_and_odd_width():
    for tag, qts in (("perout", ["4bitsym", "2bitsym"]), ("odd", ["4bitsym", "8bit"])):
        x, w = t(f"{tag}/x"), t(f"{tag}/w")
        for qt in qts:
            s = t(f"{tag}/{qt}/s")
            s = s[0] if s.numel() == 1 else s
            assert torch.equal(qat.ste_formula(x, w, s, qt, "RMS"), t(f"{tag}/{qt}/RMS/y"))


@pytest.mark.parametrize("qt,nt", [("4bitsym", "RMS"), ("Ternary", "Lin"), ("FP130", "LayerNorm"), ("2bitsym", "BatchNorm")])
def test_straight_through_gradients_equal_reference(qt, nt):
    x = t("a/x").clone().requires_grad_(True)
    w = t("a/w").clone().requires_grad_(True)
    y = qat.ste_formula(x, w, t(f"a/{qt}/s")[0], qt, nt)
    gx, gw = torch.autograd.grad(y, (x, w), t("a/gy"))
    assert torch.equal(gx, t(f"a/{qt}/{nt}/gx")) and torch.equal(gw, t(f"a/{qt}/{nt}/gw"))


def test_module_mirrors_reference_constructor_and_clipping_scalar():
    layer = qat.BitLinear(202, 24, QuantType="4bitsym", NormType="RMS")
    assert layer.bias is None and layer.bpw == 4 and layer.weight.shape == (24, 202) and not layer.s.requires_grad
    with torch.no_grad():
        layer.weight.copy_(t("a/w"))
    s = layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)
    assert torch.equal(s.reshape(-1), t("a/4bitsym/s")) and torch.equal(layer.s.data.reshape(-1), t("a/4bitsym/s"))
    assert float(layer.update_clipping_scalar(layer.weight.data, "octav")) > 0
    for bad in (dict(QuantType="3bit"), dict(WScale="PerRow"), dict(NormType="Group")):
        with pytest.raises(AssertionError):
            qat.BitLinear(8, 8, **bad)


def test_cpu_tensors_are_refused_not_emulated():
    layer = qat.BitLinear(16, 4, QuantType="4bitsym")
    with pytest.raises(RuntimeError, match="GPU op"):
        layer(torch.randn(3, 16))


def test_module_exposes_the_quantiser_pieces_the_exporter_calls():
    layer = qat.BitLinear(202, 24, QuantType="FP130", NormType="Lin")
    with torch.no_grad():
        layer.weight.copy_(t("a/w"))
    layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)
    u, scale, bpw = layer.weight_quant(layer.weight.data)
    assert bpw == 4 and torch.equal(u, t("a/FP130/w_int")) and torch.equal(scale.reshape(-1), t("a/FP130/w_scale"))
    xi, xs = layer.activation_quant(layer.Normalize(t("a/x")))
    assert torch.equal(xi, t("a/Lin/x_int"))


@pytest.mark.parametrize("tag,cfg", [("conv1", (1, 16, 1, "8bit", "None", 0)), ("convdw", (16, 16, 16, "8bit", "None", 0)),
                                     ("convdw_rms", (8, 16, 8, "4bitsym", "RMS", 1)), ("conv1_tern", (1, 12, 1, "Ternary", "RMS", 0))])
def test_conv_formula_equals_reference(tag, cfg):
    cin, cout, groups, qt, nt, pad = cfg
    x = t(f"{tag}/x").clone().requires_grad_(True)
    w = t(f"{tag}/w").clone().requires_grad_(True)
    y = qat.ste_conv_formula(x, w, t(f"{tag}/s")[0], qt, nt, 1, (pad, pad), groups)
    assert torch.equal(y, t(f"{tag}/y"))
    if f"{tag}/gx" in G:
        gx, gw = torch.autograd.grad(y, (x, w), t(f"{tag}/gy"))
        assert torch.equal(gx, t(f"{tag}/gx")) and torch.equal(gw, t(f"{tag}/gw"))


GENERAL_CONV = [("g2", 3, 1, 1, 2, "4bitsym", "RMS"), ("full", 3, 1, 0, 1, "8bit", "None"), ("dw_s2", 3, 2, 1, 6, "4bitsym", "RMS"),
                ("g2_s2", 3, 2, 0, 2, "Ternary", "RMS"), ("k5", 5, 1, 2, 1, "2bitsym", "None"), ("k1_s3", 1, 3, 0, 4, "8bit", "RMS")]


@pytest.mark.parametrize("tag,ks,stride,pad,groups,qt,nt", GENERAL_CONV)
def test_general_conv_formula_equals_reference(tag, ks, stride, pad, groups, qt, nt):
    """any groups / stride / kernel size: the restated expression (the module's backward) reproduces the reference module's outputs
    and gradients bit for bit (fixtures: tests/golden/make_qat_conv_general_golden.py)"""
    g = np.load(os.path.join(GOLDEN, "qat_bitconv2d_general.npz"))
    tt = lambda k: torch.from_numpy(g[k])
    x = tt(f"{tag}/x").clone().requires_grad_(True)
    w = tt(f"{tag}/w").clone().requires_grad_(True)
    y = qat.ste_conv_formula(x, w, tt(f"{tag}/s")[0], qt, nt, stride, (pad, pad), groups)
    assert torch.equal(y, tt(f"{tag}/y"))
    gx, gw = torch.autograd.grad(y, (x, w), tt(f"{tag}/gy"))
    assert torch.equal(gx, tt(f"{tag}/gx")) and torch.equal(gw, tt(f"{tag}/gw"))


def test_conv_module_mirrors_reference_constructor():
    layer = qat.BitConv2d(16, 16, kernel_size=3, stride=1, padding=(0, 0), groups=16, QuantType="8bit", NormType="None")
    assert layer.weight.shape == (16, 1, 3, 3) and layer.bias is None and layer.bpw == 8 and not layer.s.requires_grad
    assert float(layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)) > 0
    with pytest.raises(AssertionError):
        qat.BitConv2d(1, 4, 3, 1, 0, NormType="LayerNorm")
    with pytest.raises(RuntimeError, match="GPU op"):
        layer(torch.randn(2, 16, 14, 14))


@pytest.mark.parametrize("qt,nt", [("4bitsym", "RMS"), ("Ternary", "Lin"), ("FP130", "LayerNorm"), ("2bitsym", "BatchNorm")])
def test_backward_from_quantised_operands_equals_reference_gradients(qt, nt):
    """The module's backward does not re-run the forward: it multiplies by the quantised operands the forward op
    returns.  Same algebra as the reference's autograd graph; only the fp32 rounding of the two GEMMs may differ."""
    x, w = t("a/x"), t("a/w")
    xi, xs = qat.activation_quant(qat.normalize(x, nt))
    u, sc = qat.weight_quant(w, t(f"a/{qt}/s")[0], qt)
    gx, gw = qat.ste_backward(x, t("a/gy"), nt, xi / xs, u / sc)
    ref_gx, ref_gw = t(f"a/{qt}/{nt}/gx"), t(f"a/{qt}/{nt}/gw")
    assert (gx - ref_gx).abs().max() <= 1e-5 * ref_gx.abs().max()
    assert (gw - ref_gw).abs().max() <= 1e-5 * ref_gw.abs().max()


# ---- whole model (models.py:56-90 FCMNIST): fixtures from the reference module itself (tests/golden/make_qat_model_golden.py) ----
GM = util.qat_model_golden()
MODEL_CONFIGS = {"a": ("4bitsym", "RMS"), "b": ("4bitsym", "RMS"), "c": ("Ternary", "Lin"), "d": ("8bit", "RMS"), "e": ("2bitsym", "RMS"),
                 "f": ("Binary", "RMS"), "g": ("4bitsym", "LayerNorm"), "h": ("8bit", "LayerNorm"),
                 "i": ("Binary", "RMS"), "j": ("4bitsym", "Lin"), "k": ("8bit", "LayerNorm")}      # i - k: hidden widths 129 .. 192


def model_case(tag):
    w1, w2, w3, ncls = (int(v) for v in GM[f"{tag}/cfg"])
    nl = 4 if w3 else 3
    ws = [torch.from_numpy(GM[f"{tag}/w{l}"]) for l in range(nl)]
    ss = [torch.from_numpy(GM[f"{tag}/s{l}"]) for l in range(nl)]
    ss = [s[0] if s.numel() == 1 else s for s in ss]
    return torch.from_numpy(GM[f"{tag}/x"]), ws, ss, (w1, w2, w3, ncls)


@pytest.mark.parametrize("tag", sorted(MODEL_CONFIGS))
def test_whole_model_formula_equals_the_reference_module(tag):
    """qat.fc_model_reference (what the fused op's backward differentiates) == the reference's FCMNIST.forward, bit for bit on CPU
    PyTorch: logits, every hidden layer's output, NaN logits for the all-zero row."""
    qt, nt = MODEL_CONFIGS[tag]
    x, ws, ss, _ = model_case(tag)
    logits, hidden = qat.fc_model_reference(x, ws, ss, [qt] * len(ws), nt)
    ref_logits, ref_hidden = GM[f"{tag}/logits"], GM[f"{tag}/hidden"]
    if nt == "LayerNorm":      # (its epsilon keeps the all-zero row finite)
        assert not np.isnan(ref_logits).any()
    else:
        assert np.isnan(ref_logits[5]).all() and not np.isnan(np.delete(ref_logits, 5, axis=0)).any()
    assert np.array_equal(logits.numpy(), ref_logits, equal_nan=True)
    assert np.array_equal(hidden.numpy(), ref_hidden, equal_nan=True)
    for l, w in enumerate(ws):
        u, sc = qat.weight_quant(w, ss[l].reshape(-1, 1) if ss[l].numel() > 1 else ss[l], qt)
        assert torch.equal(u, torch.from_numpy(GM[f"{tag}/w_int{l}"]))
        assert torch.equal(torch.as_tensor(sc).reshape(-1), torch.from_numpy(GM[f"{tag}/w_scale{l}"]))


def test_whole_model_backward_from_saved_tensors_equals_reference_gradients():
    """The fused op's backward never re-runs the forward: it works from x, the saved hidden activations and w_int / w_scale.  Fed
    with the reference's own hidden activations it must reproduce the reference's autograd gradients (fp32 GEMM rounding apart)."""
    x, ws, ss, (w1, w2, w3, ncls) = model_case("a")
    keep = np.ones(len(x), bool)
    keep[5] = False
    xk = x[keep]
    hidden = torch.from_numpy(GM["a/hidden"][keep])
    wdq = [torch.from_numpy(GM[f"a/w_int{l}"]) / float(GM[f"a/w_scale{l}"][0]) for l in range(4)]
    gx, gws = qat.fc_model_backward(xk, hidden, wdq, torch.from_numpy(GM["a/gy"]), "RMS", [w1, w2, w3, ncls])
    assert (gx - torch.from_numpy(GM["a/gx"])).abs().max() <= 1e-4 * np.abs(GM["a/gx"]).max()
    for l in range(4):
        ref = torch.from_numpy(GM[f"a/gw{l}"])
        assert (gws[l] - ref).abs().max() <= 1e-4 * ref.abs().max(), l


def test_fcmnist_module_mirrors_the_reference_module():
    """Same constructor, same parameter names as models.py's FCMNIST (a reference checkpoint's state_dict loads); CPU tensors refused."""
    m = qat.FCMNIST(96, 64, 0, QuantType="4bitsym", NormType="RMS", WScale="PerTensor", num_classes=47)
    assert sorted(m.state_dict()) == ["classifier.s", "classifier.weight", "model.1.s", "model.1.weight", "model.3.s", "model.3.weight"]
    m4 = qat.FCMNIST(64, 64, 64, QuantType="Ternary")
    assert sorted(k for k in m4.state_dict() if k.endswith("weight")) == ["classifier.weight", "model.1.weight", "model.3.weight", "model.fc3.weight"]
    assert [l.out_features for l in m4.bitlinear_layers()] == [64, 64, 64, 10] and m.classifier.out_features == 47
    with pytest.raises(RuntimeError, match="GPU op"):
        m(torch.randn(3, 1, 16, 16))
    with pytest.raises(RuntimeError, match="GPU op"):
        qat.fc_model_forward(torch.randn(3, 256), [torch.randn(8, 256), torch.randn(4, 8)], [torch.ones(1)] * 2, ["8bit"] * 2, "RMS")


def test_cnnmnist_module_mirrors_the_reference_module():
    """qat.CNNMNIST: the constructor and parameter names of models.py's CNNMNIST (the model trainingparameters.yaml names): a reference
    checkpoint's state_dict loads; the fixture's weights go in by those names."""
    m = qat.CNNMNIST(96, 64, 0, cnn_width=64, QuantType="4bitsym", WScale="PerTensor", NormType="RMS", num_classes=10)
    assert sorted(m.state_dict()) == list(GM["cnn/state_keys"])
    layers = [x for x in list(m.model) + [m.classifier] if hasattr(x, "weight_quant")]
    assert [tuple(x.weight.shape) for x in layers] == [tuple(GM[f"cnn/w{l}"].shape) for l in range(6)]
    assert [x.QuantType for x in layers] == ["8bit", "8bit", "8bit", "2bitsym", "4bitsym", "4bitsym"]
    assert [x.out_features for x in m.bitlinear_layers()] == [96, 64, 10] and m.bitlinear_layers()[0].in_features == 256
    m4 = qat.CNNMNIST(64, 64, 64, cnn_width=32)
    assert "model.fc3.weight" in m4.state_dict() and m4.bitlinear_layers()[0].in_features == 128
    with pytest.raises(RuntimeError, match="GPU op"):
        m(torch.randn(2, 1, 16, 16))


@pytest.mark.parametrize("pad,nt,qt", [(0, "None", "8bit"), (1, "None", "4bitsym"), (0, "RMS", "8bit"), (2, "RMS", "None")])
def test_depthwise_backward_from_forward_passes_equals_autograd(pad, nt, qt):
    """qat.depthwise_conv_backward + the straight-through rule (what _BitConv2dFn.backward does for depthwise layers) against autograd
    through the restated BitConv2d formula, CPU: the same gradients up to float32 summation order."""
    torch.manual_seed(pad + 7)
    c, n = 6, 5
    x = torch.randn(n, c, 9, 7)
    w = torch.randn(c, 1, 3, 3) * 0.4
    s = w.abs().mean() / 0.25
    gy = torch.randn(n, c, 9 + 2 * pad - 2, 7 + 2 * pad - 2)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    want = torch.autograd.grad(qat.ste_conv_formula(xr, wr, s, qt, nt, padding=pad, groups=c), (xr, wr), gy)
    ctx = type("Ctx", (), {"saved_tensors": (x, w, s), "cfg": (qt, nt, 1, pad, c)})()
    gx, gw = qat._BitConv2dFn.backward(ctx, gy)[:2]
    for a, b_ in ((gx, want[0]), (gw, want[1])):
        assert a.shape == b_.shape and (a - b_).abs().max() <= 2e-5 * b_.abs().max()


def test_front_backward_from_saved_planes_equals_autograd_through_the_whole_front():
    """qat.cnn_front_backward (the fused front's backward: from x and the three convolutions' outputs before their ReLU) against
    autograd through the composed formula - BitConv2d / ReLU / BitConv2d / ReLU / MaxPool / BitConv2d / ReLU / MaxPool / Flatten as
    differentiable torch ops - on CPU, where the planes it is given are exactly the formula's: equal up to float32 rounding."""
    torch.manual_seed(3)
    channels, n = 6, 5
    ws = [torch.randn(channels, 1, 3, 3) * 0.3 for _ in range(3)]
    ss = [w.abs().mean() / 0.25 for w in ws]
    qts = ["8bit", "4bitsym", "8bit"]
    x = torch.randn(n, 1, 16, 16).requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in ws]
    y1 = qat.ste_conv_formula(x, wr[0], ss[0], qts[0], "None", groups=1)
    y2 = qat.ste_conv_formula(torch.relu(y1), wr[1], ss[1], qts[1], "None", groups=channels)
    y3 = qat.ste_conv_formula(torch.nn.functional.max_pool2d(torch.relu(y2), 2), wr[2], ss[2], qts[2], "None", groups=channels)
    f = torch.nn.functional.max_pool2d(torch.relu(y3), 2).flatten(1)
    g = torch.randn_like(f)
    want = torch.autograd.grad(f, [x] + wr, g)
    planes = [t.detach().contiguous(memory_format=torch.channels_last) for t in (y1, y2, y3)]
    gx, gws = qat.cnn_front_backward(x.detach(), *planes, ws, ss, qts, g)
    for a, b_ in zip([gx] + gws, want):
        assert a.shape == b_.shape and (a - b_).abs().max() <= 2e-5 * b_.abs().max().clamp(min=1e-30)
    assert qat.cnn_front_backward(x.detach(), *planes, ws, ss, qts, g, need_gx=False)[0] is None
