#!/usr/bin/env python3
"""Golden vectors for the QAT forward op, generated HERE by importing the reference's own
training layer `BitLinear` from /root/reference/BitNetMCU.py (CPU PyTorch) — the reference cannot travel to the GPU
box, the vectors can.

  python tests/golden/make_qat_golden.py      -> tests/golden/qat_bitlinear.npz

Shapes are deliberately awkward: n = 37 rows (ragged last 32-row block), d = 202 inputs (even, not a multiple of 64),
k = 24 outputs (pads to 32), plus one odd-d case (d = 77, k = 40); BitConv2d cases: the reference CNN's two layer kinds
(models.py:111-116) and padded / RMS / multi-output-per-group variants.  For every QuantType x NormType of the reference:
y; per NormType: activation_quant's integers and scales; for four combinations also the straight-through gradients.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
QUANTS = ["Binary", "BinarySym", "Ternary", "2bitsym", "4bit", "4bitsym", "FP130", "NF4", "5bitsym", "8bit"]
NORMS = ["RMS", "Lin", "BatchNorm", "LayerNorm"]
GRAD_CASES = [("4bitsym", "RMS"), ("Ternary", "Lin"), ("FP130", "LayerNorm"), ("2bitsym", "BatchNorm")]


def main():
    sys.path.insert(0, REF)
    import BitNetMCU as ref          # the reference module, unmodified
    torch.manual_seed(20240324)
    out = {}

    def case(tag, n, d, k, quants, norms, wscale="PerTensor"):
        x = torch.randn(n, d) * torch.rand(n, 1) * 3.0 + 0.1 * torch.randn(n, 1)
        x[1] *= 1e-3                    # a tiny row and a huge row: the per-row scale must absorb both
        x[2] *= 1e3
        w = torch.randn(k, d) * 0.08
        gy = torch.randn(n, k)
        out[f"{tag}/x"], out[f"{tag}/w"], out[f"{tag}/gy"] = x.numpy(), w.numpy(), gy.numpy()
        for qt in quants:
            for nt in norms:
                layer = ref.BitLinear(d, k, QuantType=qt, NormType=nt, WScale=wscale)
                with torch.no_grad():
                    layer.weight.copy_(w)
                layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)     # as training.py does before epoch 0
                xr = x.clone().requires_grad_(True)
                y = layer(xr)
                key = f"{tag}/{qt}/{nt}"
                out[key + "/y"] = y.detach().numpy()
                if qt == quants[0]:
                    xi, xs = layer.activation_quant(layer.Normalize(x))
                    out[f"{tag}/{nt}/x_int"] = xi.numpy()
                    out[f"{tag}/{nt}/x_scale"] = xs.numpy().reshape(-1)
                    out[f"{tag}/{nt}/x_norm"] = layer.Normalize(x).numpy()
                if nt == norms[0]:           # the weight side does not depend on the NormType
                    u, wsc, _ = layer.weight_quant(layer.weight.data)
                    out[f"{tag}/{qt}/s"] = layer.s.detach().numpy().reshape(-1)
                    out[f"{tag}/{qt}/w_int"] = u.numpy()
                    out[f"{tag}/{qt}/w_scale"] = np.asarray(wsc.detach().numpy(), dtype=np.float32).reshape(-1)
                if (qt, nt) in GRAD_CASES and wscale == "PerTensor":
                    gx, gw = torch.autograd.grad(y, (xr, layer.weight), gy)
                    out[key + "/gx"], out[key + "/gw"] = gx.numpy(), gw.numpy()

    case("a", 37, 202, 24, QUANTS, NORMS)
    case("odd", 5, 77, 40, ["4bitsym", "8bit"], ["RMS"])
    case("perout", 9, 64, 12, ["4bitsym", "2bitsym"], ["RMS"], wscale="PerOutput")
    # BitConv2d in the reference CNN's configuration (models.py:111-116) and two more: padding 1, RMS + 4bitsym
    def conv_case(tag, n, cin, cout, hw, groups, qt, nt, pad, grads=False):
        x = torch.randn(n, cin, hw, hw) * (torch.rand(n, cin, 1, 1) * 2 + 0.2)
        x[0, 0, 3] = 0.0                 # an all-zero image row: its scale is 127 / 1e-5
        layer = ref.BitConv2d(cin, cout, kernel_size=3, stride=1, padding=(pad, pad), groups=groups, QuantType=qt, NormType=nt)
        layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)
        xr = x.clone().requires_grad_(True)
        y = layer(xr)
        out[f"{tag}/x"], out[f"{tag}/w"], out[f"{tag}/s"] = x.numpy(), layer.weight.detach().numpy(), layer.s.detach().numpy().reshape(-1)
        out[f"{tag}/y"] = y.detach().numpy()
        if grads:
            gy = torch.randn_like(y)
            gx, gw = torch.autograd.grad(y, (xr, layer.weight), gy)
            out[f"{tag}/gy"], out[f"{tag}/gx"], out[f"{tag}/gw"] = gy.numpy(), gx.numpy(), gw.numpy()

    conv_case("conv1", 6, 1, 16, 16, 1, "8bit", "None", 0, grads=True)
    conv_case("convdw", 6, 16, 16, 14, 16, "8bit", "None", 0)
    conv_case("convdw_rms", 5, 8, 16, 7, 8, "4bitsym", "RMS", 1, grads=True)
    conv_case("conv1_tern", 3, 1, 12, 9, 1, "Ternary", "RMS", 0)
    # QuantType 'None' is not constructible through BitQuant.__init__ (BitNetMCU.py:53-66 raises), although
    # BitLinear.forward has a branch for it (:225-226): its expected output is F.linear(Normalize(x), w)
    path = os.path.join(HERE, "qat_bitlinear.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
