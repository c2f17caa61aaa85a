#!/usr/bin/env python3
"""qat.CNNMNIST (the model the reference's trainingparameters.yaml names) forward under no_grad on the GPU box: images/s of the whole
module, of its convolution front alone and of its FC stack alone.
  python profiles/cnnmnist_forward_bench.py [--rows 65536] [--steps 10]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitnetmcu_amd import qat  # noqa: E402


def timed(fn, steps, warmup=2):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--channels", type=int, default=64, help="cnn_width (the FC stack then has 4 x channels inputs)")
    ap.add_argument("--front-only", action="store_true", help="only the one-kernel convolution front (counter passes)")
    a = ap.parse_args()
    torch.manual_seed(0)
    m = qat.CNNMNIST(96, 64, 0, cnn_width=a.channels, QuantType="4bitsym", WScale="PerTensor", NormType="RMS", num_classes=10).cuda()
    x = torch.randn(a.rows, 1, 16, 16, device="cuda") * (torch.rand(a.rows, 1, 1, 1, device="cuda") * 2 + 0.05)
    out = {"rows": a.rows, "channels": a.channels}
    if a.front_only:
        cs = [c for c in m.model if isinstance(c, qat.BitConv2d)]
        ms = timed(lambda: qat.cnn_front_forward(x, [c.weight for c in cs], [c.s for c in cs]), a.steps, 1)
        print(json.dumps({"rows": a.rows, "front_fused_ms": ms, "front_fused_images_per_s": a.rows / (ms * 1e-3)}))
        return
    with torch.no_grad():
        f = m.front(x)
        ms_all = timed(lambda: m(x), a.steps)
        ms_front = timed(lambda: m.front(x), a.steps)
        def layer_by_layer():
            y = x
            for k in list(m.model)[:9]:
                y = k(y)
            return y
        ms_layers = timed(layer_by_layer, 3, 1)
        out.update({"front_layer_by_layer_ms": ms_layers, "front_layer_by_layer_images_per_s": a.rows / (ms_layers * 1e-3)})
        ls = m.bitlinear_layers()
        ms_fc = timed(lambda: qat.fc_model_forward(f, [l.weight for l in ls], [l.s for l in ls], [l.QuantType for l in ls], "RMS"), a.steps)
        if hasattr(qat, "cnn_front_forward"):
            cs = [c for c in m.model if isinstance(c, qat.BitConv2d)]
            ms_fused = timed(lambda: qat.cnn_front_forward(x, [c.weight for c in cs], [c.s for c in cs]), a.steps)
            out.update({"front_fused_ms": ms_fused, "front_fused_images_per_s": a.rows / (ms_fused * 1e-3)})
    out.update({"module_ms": ms_all, "front_ms": ms_front, "fc_stack_ms": ms_fc, "module_images_per_s": a.rows / (ms_all * 1e-3),
                "front_images_per_s": a.rows / (ms_front * 1e-3), "fc_stack_rows_per_s": a.rows / (ms_fc * 1e-3)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
