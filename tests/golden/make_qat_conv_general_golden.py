#!/usr/bin/env python3
"""Golden vectors for the GENERAL BitConv2d forward (any groups / stride / kernel size), generated HERE by importing the
reference's own `BitConv2d` from /root/reference/BitNetMCU.py (CPU PyTorch).

  python tests/golden/make_qat_conv_general_golden.py      -> tests/golden/qat_bitconv2d_general.npz

Cases: (tag, n, cin, cout, h, w, kernel, stride, padding, groups, QuantType, NormType).  The reference CNN's own two layer kinds
(stride 1, single-channel input / depthwise) are in qat_bitlinear.npz (make_qat_golden.py)."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [
    ("g2", 4, 4, 6, 10, 10, 3, 1, 1, 2, "4bitsym", "RMS"),        # two groups of two input channels, three outputs each
    ("full", 3, 3, 5, 9, 9, 3, 1, 0, 1, "8bit", "None"),          # groups = 1: every output sees all three input planes
    ("dw_s2", 5, 6, 6, 11, 11, 3, 2, 1, 6, "4bitsym", "RMS"),     # depthwise, stride 2
    ("g2_s2", 3, 4, 8, 12, 12, 3, 2, 0, 2, "Ternary", "RMS"),     # grouped + stride 2
    ("k5", 2, 2, 4, 12, 14, 5, 1, 2, 1, "2bitsym", "None"),       # 5 x 5 kernel, non-square plane
    ("k1_s3", 2, 8, 8, 7, 7, 1, 3, 0, 4, "8bit", "RMS"),          # 1 x 1 kernel, stride 3
]


def main():
    sys.path.insert(0, REF)
    import BitNetMCU as ref          # the reference module, unmodified
    torch.manual_seed(20240602)
    out = {}
    for tag, n, cin, cout, h, w, ks, stride, pad, groups, qt, nt in CASES:
        x = torch.randn(n, cin, h, w) * (torch.rand(n, cin, 1, 1) * 2 + 0.2)
        x[0, 0, 2] = 0.0                 # an all-zero image row: its scale is 127 / 1e-5
        layer = ref.BitConv2d(cin, cout, kernel_size=ks, stride=stride, padding=(pad, pad), groups=groups, QuantType=qt, NormType=nt)
        layer.update_clipping_scalar(layer.weight.data, "prop", 0.25)
        xr = x.clone().requires_grad_(True)
        y = layer(xr)
        gy = torch.randn_like(y)
        gx, gw = torch.autograd.grad(y, (xr, layer.weight), gy)
        out[f"{tag}/x"], out[f"{tag}/w"], out[f"{tag}/s"] = x.numpy(), layer.weight.detach().numpy(), layer.s.detach().numpy().reshape(-1)
        out[f"{tag}/y"], out[f"{tag}/gy"], out[f"{tag}/gx"], out[f"{tag}/gw"] = y.detach().numpy(), gy.numpy(), gx.numpy(), gw.numpy()
    path = os.path.join(HERE, "qat_bitconv2d_general.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
