#!/usr/bin/env python3
"""Where a CNNMNIST training step's time goes (GPU box): the whole step, the convolution front's backward as it is (the per-layer
ops' backward: ste_conv_formula re-run under autograd), and the library's convolution_backward alone on the three layers' shapes."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bitnetmcu_amd import qat  # noqa: E402


def timed(fn, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for k in range(steps):
        fn()
        ev[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    torch.manual_seed(0)
    m = qat.CNNMNIST(96, 64, 0, cnn_width=64, QuantType="4bitsym").cuda()
    x = torch.randn(n, 1, 16, 16, device="cuda")
    gy = torch.randn(n, 10, device="cuda")

    def step():
        m.zero_grad(set_to_none=True)
        (m(x) * gy).sum().backward()
    out = {"rows": n, "step_ms": timed(step)}
    with torch.no_grad():
        out["forward_only_ms"] = timed(lambda: m(x))
    # the library's convolution gradients alone, on the layers' shapes
    C = 64
    shapes = [((n, 1, 16, 16), (C, 1, 3, 3), 1), ((n, C, 14, 14), (C, 1, 3, 3), C), ((n, C, 6, 6), (C, 1, 3, 3), C)]
    for l, (xs, ws, groups) in enumerate(shapes):
        xx, ww = torch.randn(xs, device="cuda"), torch.randn(ws, device="cuda")
        g = torch.randn(n, C, xs[2] - 2, xs[3] - 2, device="cuda")
        out[f"conv{l + 1}_backward_ms"] = timed(lambda: torch.ops.aten.convolution_backward(g, xx, ww, None, [1, 1], [0, 0], [1, 1], False, [0, 0], groups, [True, True, False]))
        out[f"conv{l + 1}_forward_ms"] = timed(lambda: torch.nn.functional.conv2d(xx, ww, groups=groups))
        xr = xx.clone().requires_grad_(True)
        wr = ww.clone().requires_grad_(True)
        s = ww.abs().mean() / 0.25

        def formula():
            y = qat.ste_conv_formula(xr, wr, s, "8bit", "None", groups=groups)
            torch.autograd.grad(y, (xr, wr), g)
        out[f"conv{l + 1}_formula_fwd_bwd_ms"] = timed(formula)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
