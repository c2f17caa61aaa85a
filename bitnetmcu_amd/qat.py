"""QAT forward op — SURVEY.md §8(f) row 4.

The reference trains with `BitLinear` (BitNetMCU.py:198-262): Normalize -> activation_quant -> weight_quant ->
F.linear, with the straight-through estimator written as `x + (q(x) - x).detach()`.  Here the forward pass is the
fused gfx950 op `bnm_qat_bitlinear_forward_device` (csrc/bnm_qat.hip); `BitLinear` below mirrors the reference
layer's constructor, attributes and `update_clipping_scalar` so it can stand in for it in `BitNetMCU.py`'s models.

What is and is not native:
  * forward on CUDA tensors: native, one weight-quant launch + one fused normalise/quantise/GEMM launch;
  * backward (BitLinear): no forward recompute — the op hands back x_int, x_scale and w_int / w_scale, and the
    straight-through gradient is two library GEMMs against them plus Normalize's own backward through autograd
    (`ste_backward`); BitConv2d's backward re-runs the restated expression `ste_conv_formula` under autograd.  Neither
    is a hand-written HIP kernel (training is minutes-long on MNIST; row 4 is the lowest-ranked "next" item);
  * CPU tensors: refused.  There is no CPU implementation of the op in the product path.
  * `BitConv2d` (BitNetMCU.py:264-322): forward native for any group structure and stride (the reference's CNN uses
    stride 1 with a single-channel first layer and depthwise groups, models.py:111-116); PerTensor clipping scalar.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L

QUANT_TYPES = {"None": 0, "Binary": 1, "BinarySym": 2, "Ternary": 3, "2bitsym": 4, "4bit": 5, "4bitsym": 6, "FP130": 7,
               "NF4": 8, "5bitsym": 9, "8bit": 10}
NORM_TYPES = {"RMS": 0, "Lin": 1, "BatchNorm": 2, "LayerNorm": 3, "None": 4}
# bits per weight as BitQuant.__init__ assigns them (BitNetMCU.py:53-66)
BPW = {"Binary": 1, "BinarySym": 1, "2bitsym": 2, "Ternary": 1.6, "4bit": 4, "4bitsym": 4, "FP130": 4, "NF4": 4,
       "5bitsym": 5, "8bit": 8}
NF4_LEVELS = [-1.0, -0.6962, -0.5251, -0.3949, -0.2844, -0.1848, -0.0911, 0.0, 0.0796, 0.1609, 0.2461, 0.3379, 0.4407,
              0.5626, 0.723, 1.0]

_workspaces = {}


def _workspace(device, d, k):
    # one workspace per (device, STREAM, shape): it carries the quantised weights from the weight-quant launch to the GEMM
    # launch of the same call, so two streams (or two threads on different streams) must not share one
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream().cuda_stream
    key = (device.index, stream, d, k)
    ws = _workspaces.get(key)
    if ws is None:
        nbytes = int(L.load().bnm_qat_workspace_bytes(d, k))
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def bitlinear_forward(x, w, s, quant_type, norm_type, return_int=False, return_w_deq=False):
    """y = F.linear(act_quant(Normalize(x)), weight_quant(w)) on the GPU.
    x [n,d], w [k,d], s scalar or [k]/[k,1] (the layer's clipping scalar), all float32 CUDA tensors.
    return_int: also return activation_quant's integers [n,d] and scales [n]; return_w_deq: also the fake-quantised
    weights w_int / w_scale [k,d]."""
    if not (x.is_cuda and w.is_cuda):
        raise RuntimeError("bitlinear_forward is a GPU op: x and w must be CUDA tensors (there is no CPU path)")
    lib = L.load()
    qt, nt = QUANT_TYPES[quant_type], NORM_TYPES[norm_type]
    x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
    w2 = w.contiguous().float()
    s2 = torch.as_tensor(s, dtype=torch.float32, device=x.device).reshape(-1).contiguous()
    n, d = x2.shape
    k = w2.shape[0]
    if w2.shape[1] != d:
        raise ValueError("w must be [k, d]")
    y = torch.empty((n, k), dtype=torch.float32, device=x.device)
    xi = torch.empty((n, d), dtype=torch.float32, device=x.device) if return_int else None
    xsc = torch.empty((n,), dtype=torch.float32, device=x.device) if return_int else None
    wdq = torch.empty((k, d), dtype=torch.float32, device=x.device) if return_w_deq else None
    ws = _workspace(x.device, d, k)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        L.check(lib, lib.bnm_qat_bitlinear_forward_device(
            C.c_void_p(x2.data_ptr()), n, d, C.c_void_p(w2.data_ptr()), k, C.c_void_p(s2.data_ptr()), s2.numel(), qt, nt,
            C.c_void_p(y.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel() * 4,
            C.c_void_p(xi.data_ptr()) if return_int else None, C.c_void_p(xsc.data_ptr()) if return_int else None,
            C.c_void_p(wdq.data_ptr()) if return_w_deq else None, C.c_void_p(stream)), "bnm_qat_bitlinear_forward_device")
    y = y.reshape(*x.shape[:-1], k)
    extra = ((xi, xsc) if return_int else ()) + ((wdq,) if return_w_deq else ())
    return (y,) + extra if extra else y


def fc_model_supported(widths, quant_types, norm_type):
    """True when the ONE-kernel whole-model forward serves this stack of BitLinear layers (bnm_qat_model_supported): at most 256
    inputs, hidden widths <= 192, <= 64 classes, QuantTypes whose levels are int8, NormType RMS, Lin or (256 inputs) LayerNorm."""
    if any(q not in QUANT_TYPES for q in quant_types) or norm_type not in NORM_TYPES:
        return False
    nl = len(quant_types)
    if len(widths) != nl + 1 or not 2 <= nl <= 4:
        return False
    wa = (C.c_uint32 * (nl + 1))(*widths)
    qa = (C.c_int * nl)(*[QUANT_TYPES[q] for q in quant_types])
    return L.load().bnm_qat_model_supported(nl, wa, qa, NORM_TYPES[norm_type]) == 1


_model_workspaces = {}


def fc_model_forward(x, weights, scalars, quant_types, norm_type, return_hidden=False, return_w_deq=False):
    """The forward pass of a stack of BitLinear layers with ReLU between them (models.py:70-90 FCMNIST) in ONE kernel
    (bnm_qat_model_forward_device, csrc/bnm_qat_model.hip) behind a weight-preparation launch.
    x [n, d <= 256] (or [n, 1, 16, 16]: flattened; d < 256: padded with zeros for the kernel here), weights [w_l [k_l, d_l]], scalars
    [s_l] (each a scalar tensor or [k_l] / [k_l, 1]),
    quant_types [str] one per layer, all float32 CUDA tensors.  Returns logits [n, classes]; with return_hidden also the hidden
    layers' outputs after ReLU as ONE tensor [n, sum of hidden widths] (layer after layer within a row); with return_w_deq also
    the list of fake-quantised weights w_int / w_scale."""
    if not x.is_cuda:
        raise RuntimeError("fc_model_forward is a GPU op: x must be a CUDA tensor (there is no CPU path)")
    lib = L.load()
    nl = len(weights)
    x2 = x.flatten(1).contiguous().float()
    n, d = x2.shape
    if d < 256:      # (the kernel reads rows of 256 floats; the zero columns meet zero weight fragments)
        x2 = F.pad(x2, (0, 256 - d))
    ws = [w.contiguous().float() for w in weights]
    ss = [torch.as_tensor(s, dtype=torch.float32, device=x.device).reshape(-1).contiguous() for s in scalars]
    widths = [d] + [w.shape[0] for w in ws]
    for l, w in enumerate(ws):
        if w.shape[1] != widths[l]:
            raise ValueError(f"layer {l}: weight is {tuple(w.shape)}, its input has {widths[l]} values")
    if not fc_model_supported(widths, quant_types, norm_type):
        raise NotImplementedError(f"fc_model_forward: widths {widths} / QuantTypes {list(quant_types)} / NormType {norm_type} are not served by the "
                                  "fused kernel (fc_model_supported); run the layers with bitlinear_forward")
    wa = (C.c_uint32 * (nl + 1))(*widths)
    qa = (C.c_int * nl)(*[QUANT_TYPES[q] for q in quant_types])
    sc = (C.c_uint32 * nl)(*[s.numel() for s in ss])
    wp = (C.c_void_p * nl)(*[w.data_ptr() for w in ws])
    sp = (C.c_void_p * nl)(*[s.data_ptr() for s in ss])
    logits = torch.empty((n, widths[-1]), dtype=torch.float32, device=x.device)
    hidden = torch.empty((n, sum(widths[1:-1])), dtype=torch.float32, device=x.device) if return_hidden else None
    wdq = [torch.empty_like(w) for w in ws] if return_w_deq else None
    dq = (C.c_void_p * nl)(*[t.data_ptr() for t in wdq]) if return_w_deq else None
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        key = (x.device.index, stream, tuple(widths))
        wsp = _model_workspaces.get(key)
        if wsp is None:
            wsp = torch.empty((int(lib.bnm_qat_model_workspace_bytes(nl, wa)) + 3) // 4, dtype=torch.float32, device=x.device)
            _model_workspaces[key] = wsp
        L.check(lib, lib.bnm_qat_model_forward_device(
            C.c_void_p(x2.data_ptr()), n, nl, wa, wp, sp, sc, qa, NORM_TYPES[norm_type], C.c_void_p(logits.data_ptr()),
            C.c_void_p(hidden.data_ptr()) if return_hidden else None, dq, C.c_void_p(wsp.data_ptr()), wsp.numel() * 4,
            C.c_void_p(stream)), "bnm_qat_model_forward_device")
    out = (logits,) + ((hidden,) if return_hidden else ()) + ((wdq,) if return_w_deq else ())
    return out if len(out) > 1 else logits


_front_workspaces = {}


def cnn_front_supported(channels, scalars, quant_types=("8bit", "8bit", "8bit")):
    """True when the ONE-kernel convolution front serves CNNMNIST's three BitConv2d layers (bnm_qat_cnn_front_supported): an even
    number of 16 .. 128 channels, per-tensor clipping scalars, a QuantType other than 'None'."""
    if len(scalars) != 3 or len(quant_types) != 3 or any(q not in QUANT_TYPES for q in quant_types):
        return False
    counts = (C.c_uint32 * 3)(*[int(torch.as_tensor(s).numel()) for s in scalars])
    qa = (C.c_int * 3)(*[QUANT_TYPES[q] for q in quant_types])
    return bool(L.load().bnm_qat_cnn_front_supported(int(channels), counts, qa))


def cnn_front_forward(x, weights, scalars, quant_types=("8bit", "8bit", "8bit"), return_planes=False):
    """The convolution front of the reference's CNNMNIST (models.py:109-119) - BitConv2d(1 -> C) / ReLU / depthwise BitConv2d / ReLU /
    MaxPool / depthwise BitConv2d / ReLU / MaxPool / Flatten, NormType 'None' - in ONE kernel (bnm_qat_cnn_front_forward_device,
    csrc/bnm_qat_cnn.hip).  x [n, 1, 16, 16] (or [n, 256]), weights three [C, 1, 3, 3] tensors, scalars their clipping scalars;
    float32 CUDA tensors.  Returns the features [n, 4 C]; with return_planes (the training form) also the three convolutions' outputs
    before their ReLU - y1 [n, C, 14, 14], y2 [n, C, 12, 12], y3 [n, C, 4, 4], torch.channels_last tensors."""
    if not x.is_cuda:
        raise RuntimeError("cnn_front_forward is a GPU op: x must be a CUDA tensor (there is no CPU path)")
    lib = L.load()
    x2 = x.flatten(1).contiguous().float()
    n, d = x2.shape
    if d != 256 or len(weights) != 3 or len(scalars) != 3:
        raise ValueError("cnn_front_forward: 16x16 images and three convolution layers")
    ws = [w.contiguous().float() for w in weights]
    channels = ws[0].shape[0]
    if any(tuple(w.shape) != (channels, 1, 3, 3) for w in ws):
        raise ValueError(f"cnn_front_forward: weights must be [C, 1, 3, 3], got {[tuple(w.shape) for w in ws]}")
    ss = [torch.as_tensor(s, dtype=torch.float32, device=x.device).reshape(-1).contiguous() for s in scalars]
    if not cnn_front_supported(channels, ss, quant_types):
        raise NotImplementedError(f"cnn_front_forward: {channels} channels / {[s.numel() for s in ss]} clipping scalars / {list(quant_types)} are not served by the "
                                  "fused kernel (cnn_front_supported); run the layers with bitconv2d_forward")
    sc = (C.c_uint32 * 3)(*[s.numel() for s in ss])
    qa = (C.c_int * 3)(*[QUANT_TYPES[q] for q in quant_types])
    wp = (C.c_void_p * 3)(*[w.data_ptr() for w in ws])
    sp = (C.c_void_p * 3)(*[s.data_ptr() for s in ss])
    features = torch.empty((n, 4 * channels), dtype=torch.float32, device=x.device)
    planes = [torch.empty((n, channels, k, k), dtype=torch.float32, device=x.device, memory_format=torch.channels_last) for k in (14, 12, 4)] \
        if return_planes else [None] * 3
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        key = (x.device.index, stream, channels)
        wsp = _front_workspaces.get(key)
        if wsp is None:
            wsp = torch.empty((int(lib.bnm_qat_cnn_front_workspace_bytes(channels)) + 3) // 4, dtype=torch.float32, device=x.device)
            _front_workspaces[key] = wsp
        L.check(lib, lib.bnm_qat_cnn_front_forward_train_device(
            C.c_void_p(x2.data_ptr()), n, channels, wp, sp, sc, qa, C.c_void_p(features.data_ptr()),
            *[C.c_void_p(t.data_ptr()) if return_planes else None for t in planes], C.c_void_p(wsp.data_ptr()),
            wsp.numel() * 4, C.c_void_p(stream)), "bnm_qat_cnn_front_forward_train_device")
    return (features, *planes) if return_planes else features


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def bitconv2d_forward(x, w, s, quant_type, norm_type, stride=1, padding=0, groups=1):
    """y = F.conv2d(act_quant(Normalize(x)), weight_quant(w), stride, padding, groups) on the GPU: any group structure, any
    (square) stride, symmetric zero padding.  x [n,cin,h,w], w [cout,cin/groups,kh,kw], s the PerTensor clipping scalar; float32
    CUDA tensors.  (PerOutput clipping is not offered for the convolution: the reference's own PerOutput scalars do not broadcast
    against a 4-D weight, BitNetMCU.py:102-108 / :136-148.)"""
    if not (x.is_cuda and w.is_cuda):
        raise RuntimeError("bitconv2d_forward is a GPU op: x and w must be CUDA tensors (there is no CPU path)")
    sh, sw = _pair(stride)
    if sh != sw:
        raise NotImplementedError("bitconv2d_forward: square stride only")
    ph, pw = _pair(padding)
    if ph != pw:
        raise NotImplementedError("bitconv2d_forward: symmetric padding only")
    lib = L.load()
    x4 = x.contiguous().float()
    w4 = w.contiguous().float()
    n, cin, h, wd = x4.shape
    cout, cpg, kh, kw = w4.shape
    if cin % groups or cout % groups or cpg != cin // groups:
        raise ValueError("w must be [cout, cin / groups, kh, kw] with cin and cout multiples of groups")
    s2 = torch.as_tensor(s, dtype=torch.float32, device=x.device).reshape(-1).contiguous()
    if s2.numel() != 1:
        raise NotImplementedError("bitconv2d_forward: PerTensor clipping scalar only")
    y = torch.empty((n, cout, (h + 2 * ph - kh) // sh + 1, (wd + 2 * ph - kw) // sh + 1), dtype=torch.float32, device=x.device)
    ws = _workspace(x.device, cpg * kh * kw, cout)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        L.check(lib, lib.bnm_qat_bitconv2d_forward_device(
            C.c_void_p(x4.data_ptr()), n, cin, h, wd, C.c_void_p(w4.data_ptr()), cout, kh, kw, ph, sh, groups,
            C.c_void_p(s2.data_ptr()), QUANT_TYPES[quant_type], NORM_TYPES[norm_type], C.c_void_p(y.data_ptr()),
            C.c_void_p(ws.data_ptr()), ws.numel() * 4, C.c_void_p(stream)), "bnm_qat_bitconv2d_forward_device")
    return y


# ---- the reference's expression, restated (used for the straight-through backward and pinned by
# ---- tests/test_qat_cpu.py against fixtures generated from the reference module itself) -------------------------
def normalize(x, norm_type):
    """BitLinear.Normalize (BitNetMCU.py:237-262)."""
    if norm_type == "RMS":
        return x / torch.sqrt(torch.mean(x ** 2, dim=-1, keepdim=True))
    if norm_type == "Lin":
        return x / torch.mean(torch.abs(x), dim=-1, keepdim=True)
    if norm_type == "BatchNorm":
        return (x - torch.mean(x, dim=0, keepdim=True)) / torch.sqrt(torch.var(x, dim=0, keepdim=True, unbiased=False) + 1e-5)
    if norm_type == "LayerNorm":
        return (x - torch.mean(x, dim=-1, keepdim=True)) / torch.sqrt(torch.var(x, dim=-1, keepdim=True, unbiased=False) + 1e-5)
    if norm_type == "None":
        return x
    raise AssertionError(f"Invalid NormType: {norm_type}")


def activation_quant(x):
    """BitQuant.activation_quant (BitNetMCU.py:119-128): per-row 8-bit absmax."""
    scale = 127.0 / x.abs().max(dim=-1, keepdim=True).values.clamp(min=1e-5)
    return (x * scale).round().clamp(-128, 127), scale


def weight_quant(w, s, quant_type):
    """BitQuant.weight_quant (BitNetMCU.py:130-177): levels u and scale."""
    if quant_type == "FP130":
        scale = 128.0 / s
    elif quant_type == "NF4":
        scale = 1.0 / s
    elif quant_type == "Ternary":
        scale = 1.0 / w.abs().mean().clamp(min=1e-5)
    else:
        scale = (2.0 ** (BPW[quant_type] - 1)) / s
    if quant_type == "Ternary":
        u = (w * scale).round().clamp(-1, 1)
    elif quant_type == "Binary":
        u = (w - w.mean()).sign()
    elif quant_type == "BinarySym":
        u = w.sign()
    elif quant_type == "2bitsym":
        u = (w * scale - 0.5).round().clamp(-2, 1) + 0.5
    elif quant_type == "4bit":
        u = (w * scale - 0.01).round().clamp(-8, 7) + 0.01
    elif quant_type == "4bitsym":
        u = (w * scale - 0.5).round().clamp(-8, 7) + 0.5
    elif quant_type == "5bitsym":
        u = (w * scale - 0.5).round().clamp(-16, 15) + 0.5
    elif quant_type == "8bit":
        u = (w * scale).round().clamp(-128, 127)
    elif quant_type == "FP130":
        e = (w * scale).abs().log2().floor().clamp(0, 7)
        u = w.sign() * e.exp2()
    elif quant_type == "NF4":
        levels = torch.tensor(NF4_LEVELS, device=w.device)
        u = levels[torch.argmin(torch.abs((w * scale).unsqueeze(-1) - levels), dim=-1)]
    else:
        raise AssertionError(f"Invalid QuantType: {quant_type}")
    return u, scale


def ste_formula(x, w, s, quant_type, norm_type):
    """BitLinear.forward (BitNetMCU.py:214-235) as differentiable PyTorch ops (straight-through estimator)."""
    x_norm = normalize(x, norm_type)
    if quant_type == "None":
        return F.linear(x_norm, w)
    if torch.is_tensor(s) and s.numel() > 1:
        # PerOutput: one clipping scalar per weight ROW.  (Divergence from BitNetMCU.py:102-103: the reference's octav
        # branch stacks the per-row scalars into shape [k], which broadcasts against w [k, d] along its LAST axis - an error
        # unless k == d, and per COLUMN when k == d; its 'prop' branch (:107-108, keepdim) and the option's name say per
        # row, which is what this mirror and the fused op apply for every shape.)
        s = s.reshape(-1, 1)
    x_int, x_scale = activation_quant(x_norm)
    x_quant = x_norm + (x_int / x_scale - x_norm).detach()
    w_int, w_scale = weight_quant(w, s, quant_type)
    w_quant = w + (w_int / w_scale - w).detach()
    return F.linear(x_quant, w_quant)


def ste_conv_formula(x, w, s, quant_type, norm_type, stride=1, padding=0, groups=1):
    """BitConv2d.forward (BitNetMCU.py:284-305) as differentiable PyTorch ops."""
    if norm_type == "RMS":
        x_norm = x / torch.sqrt(torch.mean(x ** 2, dim=(-2, -1), keepdim=True))     # :306-308, per plane
    elif norm_type == "None":
        x_norm = x
    else:
        raise AssertionError(f"Invalid NormType: {norm_type}. Expected one of: 'RMS', 'None'")
    if quant_type == "None":
        return F.conv2d(x_norm, w, stride=stride, padding=padding, groups=groups)
    x_int, x_scale = activation_quant(x_norm)                                        # per image row (last dimension)
    x_quant = x_norm + (x_int / x_scale - x_norm).detach()
    w_int, w_scale = weight_quant(w, s, quant_type)
    w_quant = w + (w_int / w_scale - w).detach()
    return F.conv2d(x_quant, w_quant, groups=groups, stride=stride, padding=padding, bias=None)


def depthwise_conv_backward(gy, x_q, w_q, padding):
    """Gradients of y = F.conv2d(x_q, w_q, stride 1, padding, groups = channels) for a depthwise layer (one tap set per channel) from
    library FORWARD passes only: the input gradient is the full correlation of gy with the flipped taps (a depthwise convolution of the
    padded gy), the weight gradient kh x kw slice-multiply-reduce passes.  (The library's own depthwise convolution_backward runs at
    66 GFLOP/s on this GPU - 41 ms for 16,384 images of 64 channels x 14 x 14, 20 x its forward - and is all of a CNNMNIST training step:
    profiles/probes/cnnmnist_backward_parts.py.)"""
    kh, kw = w_q.shape[2], w_q.shape[3]
    ph, pw = _pair(padding)
    gx = F.conv2d(F.pad(gy, (kw - 1 - pw, kw - 1 - pw, kh - 1 - ph, kh - 1 - ph)), w_q.flip(2, 3), groups=w_q.shape[0])
    xp = F.pad(x_q, (pw, pw, ph, ph)) if ph or pw else x_q
    ho, wo = gy.shape[2], gy.shape[3]
    gw = torch.stack([(xp[:, :, dy:dy + ho, dx:dx + wo] * gy).sum(dim=(0, 2, 3)) for dy in range(kh) for dx in range(kw)], dim=1)
    return gx, gw.reshape(w_q.shape)


class _BitConv2dFn(torch.autograd.Function):
    """Forward: the fused HIP op.  Backward: the straight-through gradients of BitConv2d.forward (BitNetMCU.py:284-305; the detach()
    terms carry none): dy / dx_norm and dy / dw are the convolution's gradients at the quantised operands x_int / x_scale and
    w_int / w_scale, which the restated formula recomputes from the saved input; Normalize's own backward through autograd.  Depthwise
    layers of stride 1 take `depthwise_conv_backward`; everything else the formula under autograd (the library's gradients)."""

    @staticmethod
    def forward(ctx, x, w, s, quant_type, norm_type, stride, padding, groups):
        ctx.save_for_backward(x, w, s)
        ctx.cfg = (quant_type, norm_type, stride, padding, groups)
        return bitconv2d_forward(x, w, s, quant_type, norm_type, stride, padding, groups)

    @staticmethod
    def backward(ctx, gy):
        x, w, s = ctx.saved_tensors
        quant_type, norm_type, stride, padding, groups = ctx.cfg
        depthwise = groups == x.shape[1] == w.shape[0] and w.shape[1] == 1 and _pair(stride) == (1, 1)
        ph, pw = _pair(padding)
        if depthwise and ph < w.shape[2] and pw < w.shape[3]:
            with torch.enable_grad():
                xr = x.detach().requires_grad_(norm_type != "None")
                x_norm = xr if norm_type == "None" else xr / torch.sqrt(torch.mean(xr ** 2, dim=(-2, -1), keepdim=True))
            with torch.no_grad():
                if quant_type == "None":
                    x_q, w_q = x_norm, w
                else:
                    x_int, x_scale = activation_quant(x_norm)
                    x_q = x_int / x_scale
                    w_int, w_scale = weight_quant(w, s, quant_type)
                    w_q = w_int / w_scale
                gx, gw = depthwise_conv_backward(gy.contiguous(), x_q, w_q, padding)
            if norm_type != "None":
                (gx,) = torch.autograd.grad(x_norm, xr, gx)
            return gx, gw, None, None, None, None, None, None
        with torch.enable_grad():
            xr = x.detach().requires_grad_(True)
            wr = w.detach().requires_grad_(True)
            y = ste_conv_formula(xr, wr, s.detach(), *ctx.cfg)
            gx, gw = torch.autograd.grad(y, (xr, wr), gy)
        return gx, gw, None, None, None, None, None, None


def cnn_front_backward(x, y1, y2, y3, weights, scalars, quant_types, g_features, need_gx=True):
    """Gradients of the convolution front from what its training form saved - the input and the three convolutions' outputs before their
    ReLU.  Pooling and ReLU backward through autograd on those planes; every BitConv2d's straight-through gradients (BitNetMCU.py:284-305)
    at the quantised operands, which are functions of the plane in front of the layer: the depthwise layers through
    `depthwise_conv_backward`, conv1 (one input plane shared by all channels) as nine slice-multiply-reduce passes for the taps and -
    only where the images themselves want a gradient - a library convolution for the input.  Returns (gx or None, [gw1, gw2, gw3])."""
    n, channels = y1.shape[0], y1.shape[1]
    g = g_features.reshape(n, channels, 2, 2)
    gws = [None] * 3
    with torch.enable_grad():
        leaf = y3.detach().requires_grad_(True)
        (g,) = torch.autograd.grad(F.max_pool2d(torch.relu(leaf), 2), leaf, g)
    for l, plane in ((2, y2), (1, y1)):
        with torch.enable_grad():
            leaf = plane.detach().requires_grad_(True)
            a = torch.relu(leaf)
            a = F.max_pool2d(a, 2) if l == 2 else a      # the layer's input, as a function of the plane in front of it
        with torch.no_grad():
            if quant_types[l] == "None":
                x_q, w_q = a, weights[l]
            else:
                x_int, x_scale = activation_quant(a)
                x_q = x_int / x_scale
                w_int, w_scale = weight_quant(weights[l], scalars[l], quant_types[l])
                w_q = w_int / w_scale
            ga, gws[l] = depthwise_conv_backward(g.contiguous(memory_format=torch.channels_last), x_q, w_q, 0)
        (g,) = torch.autograd.grad(a, leaf, ga)
    with torch.no_grad():
        x4 = x.detach().reshape(n, 1, 16, 16)
        if quant_types[0] == "None":
            x_q, w_q = x4, weights[0]
        else:
            x_int, x_scale = activation_quant(x4)
            x_q = x_int / x_scale
            w_int, w_scale = weight_quant(weights[0], scalars[0], quant_types[0])
            w_q = w_int / w_scale
        gws[0] = torch.stack([(x_q[:, :, dy:dy + 14, dx:dx + 14] * g).sum(dim=(0, 2, 3)) for dy in range(3) for dx in range(3)], dim=1).reshape(weights[0].shape)
        gx = F.conv_transpose2d(g, w_q) if need_gx else None
    return gx, gws


class _CNNFrontFn(torch.autograd.Function):
    """Forward: the one-kernel front in its training form (the planes the backward needs are written once, by the kernel that computes
    them).  Backward: cnn_front_backward - library calls and elementwise passes, no hand-written kernel."""

    @staticmethod
    def forward(ctx, x, quant_types, w1, w2, w3, s1, s2, s3):
        features, y1, y2, y3 = cnn_front_forward(x, [w1, w2, w3], [s1, s2, s3], quant_types, return_planes=True)
        ctx.save_for_backward(x, y1, y2, y3, w1, w2, w3, s1, s2, s3)
        ctx.qts = tuple(quant_types)
        return features

    @staticmethod
    def backward(ctx, g_features):
        x, y1, y2, y3, w1, w2, w3, s1, s2, s3 = ctx.saved_tensors
        gx, gws = cnn_front_backward(x, y1, y2, y3, [w1, w2, w3], [s1, s2, s3], ctx.qts, g_features, need_gx=ctx.needs_input_grad[0])
        return (gx.reshape(x.shape) if gx is not None else None), None, gws[0], gws[1], gws[2], None, None, None


def ste_backward(x, gy, norm_type, x_q, w_q):
    """Gradients of BitLinear's straight-through forward given its quantised operands: x_q = x_int / x_scale (or the
    normalised input when QuantType is 'None'; pass None to use it), w_q = w_int / w_scale (or w)."""
    g2 = gy.reshape(-1, gy.shape[-1])
    with torch.enable_grad():
        xr = x.detach().requires_grad_(True)
        xn = normalize(xr, norm_type)
    if x_q is None:
        x_q = xn.detach().reshape(-1, x.shape[-1])
    gw = g2.t() @ x_q
    (gx,) = torch.autograd.grad(xn, xr, (g2 @ w_q).reshape(x.shape))
    return gx, gw


class _BitLinearFn(torch.autograd.Function):
    """Forward: the fused HIP op, which also hands back what the straight-through estimator differentiates against
    (x_int, x_scale, w_int / w_scale).  Backward: dy/dx_norm = w_quant and dy/dw = x_quant (BitNetMCU.py:228-234: the
    detach() terms carry no gradient), i.e. two library GEMMs, then the Normalize step's own backward via autograd."""

    @staticmethod
    def forward(ctx, x, w, s, quant_type, norm_type):
        if quant_type == "None":
            y = bitlinear_forward(x, w, s, quant_type, norm_type)
            ctx.save_for_backward(x, w)
        else:
            y, xi, xs, wq = bitlinear_forward(x, w, s, quant_type, norm_type, return_int=True, return_w_deq=True)
            ctx.save_for_backward(x, xi, xs, wq)
        ctx.qn = (quant_type, norm_type)
        return y

    @staticmethod
    def backward(ctx, gy):
        quant_type, norm_type = ctx.qn
        if quant_type == "None":
            x, w_q = ctx.saved_tensors
            x_q = None
        else:
            x, xi, xs, w_q = ctx.saved_tensors
            x_q = xi / xs[:, None]
        gx, gw = ste_backward(x, gy, norm_type, x_q, w_q)
        return gx, gw, None, None, None


class BitLinear(nn.Linear):
    """Stand-in for the reference's BitLinear (BitNetMCU.py:198-262): same constructor, `weight`, `s`, `bpw`,
    `QuantType`, `WScale`, `NormType`, `update_clipping_scalar`; forward runs the fused HIP op on CUDA inputs."""

    def __init__(self, in_features, out_features, bias=False, QuantType="Binary", WScale="PerTensor", NormType="RMS"):
        super().__init__(in_features, out_features, bias=False)
        if QuantType != "None" and QuantType not in BPW:
            raise AssertionError(f"Invalid QuantType: {QuantType}")
        if WScale not in ("PerOutput", "PerTensor"):
            raise AssertionError(f"Invalid WScale: {WScale}. Expected one of: 'PerTensor', 'PerOutput'")
        if NormType not in ("RMS", "Lin", "BatchNorm", "LayerNorm"):
            raise AssertionError(f"Invalid NormType: {NormType}. Expected one of: 'RMS', 'Lin', 'BatchNorm', 'LayerNorm'")
        self.QuantType, self.WScale, self.NormType = QuantType, WScale, NormType
        self.bpw = BPW.get(QuantType, 0)
        self.s = nn.Parameter(torch.tensor(1.0), requires_grad=False)

    def forward(self, x):
        return _BitLinearFn.apply(x, self.weight, self.s, self.QuantType, self.NormType)

    # the pieces the reference's exporter and QuantizedModel call on a layer (BitNetMCU.py:351-418 uses weight_quant)
    def Normalize(self, x):
        return normalize(x, self.NormType)

    def activation_quant(self, x):
        return activation_quant(x)

    def weight_quant(self, w):
        s = self.s.reshape(-1, 1) if self.s.numel() > 1 else self.s
        u, scale = weight_quant(w, s, self.QuantType)
        return u, scale, self.bpw

    def octav(self, tensor, num_iterations=10, s=-1):
        """Optimum clipping scalar by Newton iteration (BitNetMCU.py:71-83; C. Sakr et al. 2022)."""
        if s < 0:
            s = tensor.abs().mean().clamp(min=1e-5) * 0.25
        a = tensor.abs()
        for _ in range(num_iterations):
            inside = (a <= s).float()
            outside = (a > s).float()
            s = torch.sum(a * outside) / ((4 ** -self.bpw / 3) * torch.sum(inside) + torch.sum(outside))
        return s

    def update_clipping_scalar(self, w, algorithm="octav", quantscale=0.25):
        """BitQuant.update_clipping_scalar (BitNetMCU.py:85-117)."""
        s = self.s
        if algorithm == "octav":
            s = torch.stack([self.octav(row, 10) for row in w]) if self.WScale == "PerOutput" else self.octav(w, 10, s)
        elif algorithm == "prop":
            if self.WScale == "PerOutput":
                s = w.abs().max(dim=-1, keepdim=True)[0].clamp(min=1e-5) / quantscale
            else:
                s = w.abs().mean().clamp(min=1e-5) / quantscale
        else:
            raise AssertionError(f"Invalid algorithm: {algorithm}. Expected one of: 'octav', 'prop'")
        self.s = nn.Parameter(s.detach(), requires_grad=False)
        return s


class BitConv2d(nn.Conv2d):
    """Stand-in for the reference's BitConv2d (BitNetMCU.py:264-322): same constructor and attributes; forward runs
    the fused HIP op on CUDA inputs (any groups / stride)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, groups=1, QuantType="4bitsym",
                 WScale="PerTensor", NormType="RMS"):
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, groups=groups,
                         bias=False)
        if QuantType != "None" and QuantType not in BPW:
            raise AssertionError(f"Invalid QuantType: {QuantType}")
        if WScale not in ("PerOutput", "PerTensor"):
            raise AssertionError(f"Invalid WScale: {WScale}. Expected one of: 'PerTensor', 'PerOutput'")
        if NormType not in ("RMS", "None"):
            raise AssertionError(f"Invalid NormType: {NormType}. Expected one of: 'RMS', 'None'")
        self.QuantType, self.WScale, self.NormType = QuantType, WScale, NormType
        self.bpw = BPW.get(QuantType, 0)
        self.s = nn.Parameter(torch.tensor(1.0), requires_grad=False)

    def forward(self, x):
        return _BitConv2dFn.apply(x, self.weight, self.s, self.QuantType, self.NormType, self.stride, self.padding,
                                  self.groups)

    def weight_quant(self, w):
        u, scale = weight_quant(w, self.s, self.QuantType)
        return u, scale, self.bpw

    def activation_quant(self, x):
        return activation_quant(x)

    octav = BitLinear.octav
    update_clipping_scalar = BitLinear.update_clipping_scalar


def fc_model_reference(x, weights, scalars, quant_types, norm_type):
    """models.py:70-90 FCMNIST.forward as the restated differentiable PyTorch ops (`ste_formula` layer after layer, ReLU between):
    -> (logits, hidden [n, sum of hidden widths]).  What tests pin against the reference module and what the fused op's backward
    differentiates."""
    h = x.flatten(1)
    hidden = []
    for l, (w, s, qt) in enumerate(zip(weights, scalars, quant_types)):
        h = ste_formula(h, w, s, qt, norm_type)
        if l + 1 < len(weights):
            h = F.relu(h)
            hidden.append(h)
    return h, torch.cat(hidden, dim=1)


def fc_model_backward(x, hidden, w_deq, gy, norm_type, outs):
    """Gradients of the whole-model forward from what the fused op saved: x, the hidden layers' outputs after ReLU (one tensor,
    layer after layer within a row) and w_int / w_scale per layer.  Per layer, last to first: activation_quant of the saved layer
    input by the restated formula, the straight-through gradients (`ste_backward`), ReLU's mask.  -> (gx, [gw per layer])"""
    n_layers = len(w_deq)
    x2 = x.flatten(1)
    offs = [0]
    for k in outs[:-1]:
        offs.append(offs[-1] + k)
    gws = [None] * n_layers
    g, gx = gy, None
    for l in range(n_layers - 1, -1, -1):
        xin = x2 if l == 0 else hidden[:, offs[l - 1]:offs[l]]
        xi, xs = activation_quant(normalize(xin, norm_type))
        gx, gws[l] = ste_backward(xin, g, norm_type, xi / xs, w_deq[l])
        if l > 0:
            g = gx * (xin > 0).to(gx.dtype)
    return gx.reshape(x.shape), gws


class _FCModelFn(torch.autograd.Function):
    """Forward: the ONE-kernel whole-model op, which also writes what a backward pass needs (every hidden layer's output after ReLU
    = the next layer's input, and w_int / w_scale per layer).  Backward: per layer, last to first, the straight-through gradients
    (`ste_backward`: two library GEMMs + Normalize's own backward through autograd) with activation_quant of the saved layer input
    recomputed by the restated formula, and ReLU's mask from the saved outputs.  No hand-written backward kernels."""

    @staticmethod
    def forward(ctx, x, norm_type, quant_types, n_layers, *ws):
        weights, scalars = ws[:n_layers], ws[n_layers:]
        logits, hidden, wdq = fc_model_forward(x, weights, scalars, quant_types, norm_type, return_hidden=True, return_w_deq=True)
        ctx.save_for_backward(x, hidden, *wdq)
        ctx.cfg = (norm_type, n_layers, [w.shape[0] for w in weights])
        return logits

    @staticmethod
    def backward(ctx, gy):
        norm_type, n_layers, outs = ctx.cfg
        x, hidden = ctx.saved_tensors[:2]
        gx, gws = fc_model_backward(x, hidden, ctx.saved_tensors[2:], gy, norm_type, outs)
        return (gx, None, None, None) + tuple(gws) + (None,) * n_layers


class FCMNIST(nn.Module):
    """Stand-in for the reference's FCMNIST (models.py:56-90): same constructor, same module / parameter names (`model.1`, `model.3`,
    `model.fc3`, `classifier`), so the reference's checkpoints load.  On CUDA inputs the whole forward pass is ONE kernel
    (`fc_model_forward`) when the fused op serves the configuration, else layer by layer through `BitLinear`'s fused op."""

    def __init__(self, network_width1=64, network_width2=64, network_width3=64, QuantType="Binary", WScale="PerTensor", NormType="RMS",
                 num_classes: int = 10):
        super().__init__()
        self.network_width1, self.network_width2, self.network_width3 = network_width1, network_width2, network_width3
        self.model = nn.Sequential(
            nn.Flatten(),
            BitLinear(1 * 16 * 16, network_width1, QuantType=QuantType, NormType=NormType, WScale=WScale),
            nn.ReLU(),
            BitLinear(network_width1, network_width2, QuantType=QuantType, NormType=NormType, WScale=WScale),
            nn.ReLU())
        if network_width3 > 0:
            self.model.add_module("fc3", BitLinear(network_width2, network_width3, QuantType=QuantType, NormType=NormType, WScale=WScale))
            self.model.add_module("relu_fc2", nn.ReLU())
        last_width = network_width3 if network_width3 > 0 else network_width2
        self.classifier = BitLinear(last_width, num_classes, QuantType=QuantType, NormType=NormType, WScale=WScale)

    def bitlinear_layers(self):
        return [m for m in self.model if isinstance(m, BitLinear)] + [self.classifier]

    def fused(self, x):
        """True when forward(x) runs the one-kernel op."""
        ls = self.bitlinear_layers()
        return x.is_cuda and fc_model_supported([ls[0].in_features] + [m.out_features for m in ls], [m.QuantType for m in ls], ls[0].NormType)

    def forward(self, x):
        if not self.fused(x):
            return self.classifier(self.model(x))
        ls = self.bitlinear_layers()
        return _FCModelFn.apply(x, ls[0].NormType, [m.QuantType for m in ls], len(ls), *[m.weight for m in ls], *[m.s for m in ls])


class CNNMNIST(nn.Module):
    """Stand-in for the reference's CNNMNIST (models.py:93-139; the model trainingparameters.yaml names): same constructor, same
    module / parameter names (`model.0`, `model.2`, `model.5` the convolutions, `model.9`, `model.11`, `model.fc3`, `classifier`).
    The depthwise-separable front runs as ONE kernel (`cnn_front_forward`; with a gradient asked for: its training form, which writes
    the planes `cnn_front_backward` works from), layer by layer through `BitConv2d`'s op where the kernel does not serve the
    configuration; the FC stack behind Flatten - 64 channels x 4
    = 256 inputs - runs as ONE kernel (`fc_model_forward`: per-layer QuantTypes: 2bitsym first, then `QuantType`) where the fused op
    serves the configuration, else layer by layer."""

    def __init__(self, network_width1=64, network_width2=64, network_width3=64, cnn_width=64, QuantType="Binary", WScale="PerTensor",
                 NormType="RMS", num_classes: int = 10):
        super().__init__()
        self.network_width1, self.network_width2, self.network_width3, self.cnn_width = network_width1, network_width2, network_width3, cnn_width
        self.model = nn.Sequential(
            BitConv2d(1, cnn_width, kernel_size=3, stride=1, padding=(0, 0), groups=1, QuantType="8bit", NormType="None", WScale=WScale),
            nn.ReLU(),
            BitConv2d(cnn_width, cnn_width, kernel_size=3, stride=1, padding=(0, 0), groups=cnn_width, QuantType="8bit", NormType="None", WScale=WScale),
            nn.ReLU(),
            nn.MaxPool2d(kernel_size=2, stride=2),
            BitConv2d(cnn_width, cnn_width, kernel_size=3, stride=1, padding=(0, 0), groups=cnn_width, QuantType="8bit", NormType="None", WScale=WScale),
            nn.ReLU(),
            nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Flatten(),
            BitLinear(cnn_width * 4, network_width1, QuantType="2bitsym", NormType=NormType, WScale=WScale),
            nn.ReLU(),
            BitLinear(network_width1, network_width2, QuantType=QuantType, NormType=NormType, WScale=WScale),
            nn.ReLU())
        if network_width3 > 0:
            self.model.add_module("fc3", BitLinear(network_width2, network_width3, QuantType=QuantType, NormType=NormType, WScale=WScale))
            self.model.add_module("relu_fc2", nn.ReLU())
        last_width = network_width3 if network_width3 > 0 else network_width2
        self.classifier = BitLinear(last_width, num_classes, QuantType=QuantType, NormType=NormType, WScale=WScale)

    def front_fused(self, x):
        """True when front(x) runs as the one-kernel op: a CUDA batch of 16x16 images in a configuration the kernel serves (where a
        gradient is asked for: in its training form, which also writes the planes the backward pass works from)."""
        convs = [m for m in list(self.model)[:9] if isinstance(m, BitConv2d)]
        return (x.is_cuda and tuple(x.shape[1:]) == (1, 16, 16) and all(c.NormType == "None" for c in convs)
                and cnn_front_supported(self.cnn_width, [c.s for c in convs], [c.QuantType for c in convs]))

    def front(self, x):
        """the convolution front up to and including Flatten (models.py:109-119)"""
        if self.front_fused(x):
            convs = [m for m in list(self.model)[:9] if isinstance(m, BitConv2d)]
            if torch.is_grad_enabled() and (x.requires_grad or any(c.weight.requires_grad for c in convs)):
                return _CNNFrontFn.apply(x, [c.QuantType for c in convs], *[c.weight for c in convs], *[c.s for c in convs])
            return cnn_front_forward(x, [c.weight for c in convs], [c.s for c in convs], [c.QuantType for c in convs])
        for m in list(self.model)[:9]:
            x = m(x)
        return x

    def bitlinear_layers(self):
        return [m for m in list(self.model)[9:] if isinstance(m, BitLinear)] + [self.classifier]

    def fused(self, x):
        """True when forward(x) runs the FC stack as the one-kernel op."""
        ls = self.bitlinear_layers()
        return x.is_cuda and fc_model_supported([ls[0].in_features] + [m.out_features for m in ls], [m.QuantType for m in ls], ls[0].NormType)

    def forward(self, x):
        f = self.front(x)
        ls = self.bitlinear_layers()
        if not self.fused(x):
            for m in list(self.model)[9:]:
                f = m(f)
            return self.classifier(f)
        return _FCModelFn.apply(f, ls[0].NormType, [m.QuantType for m in ls], len(ls), *[m.weight for m in ls], *[m.s for m in ls])
