#!/usr/bin/env python3
"""Whole-model QAT forward (bnm_qat_model_forward_device) timed on the GPU box: rows/s and the share of the HBM roofline on
1,024 B read + 4 x classes B written per row, next to the layer-by-layer path (bnm_qat_bitlinear_forward_device x layers).
  python profiles/qat_model_bench.py [--rows 1000000] [--widths 64 64 64] [--quant 4bitsym] [--norm RMS] [--classes 10]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitnetmcu_amd import qat  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for k in range(steps):
        fn()
        ev[k + 1].record()
    torch.cuda.synchronize()
    return [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--widths", type=int, nargs="+", default=[64, 64, 64])
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--quant", default="4bitsym")
    ap.add_argument("--norm", default="RMS")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layerwise", action="store_true", help="also time the layer-by-layer path")
    ap.add_argument("--hidden", action="store_true", help="also time the training form: hidden activations + w_int / w_scale written")
    a = ap.parse_args()
    torch.manual_seed(0)
    widths = [256] + [w for w in a.widths if w] + [a.classes]
    ws = [(torch.randn(widths[l + 1], widths[l], device="cuda") * 0.08) for l in range(len(widths) - 1)]
    ss = [w.abs().mean().reshape(1) / 0.25 for w in ws]
    qts = [a.quant] * len(ws)
    x = torch.randn(a.rows, 256, device="cuda") * (torch.rand(a.rows, 1, device="cuda") * 2 + 0.05)
    ms = timed(lambda: qat.fc_model_forward(x, ws, ss, qts, a.norm), a.steps, a.warmup)
    bpr = 1024 + 4 * a.classes
    out = {"rows": a.rows, "widths": widths, "quant": a.quant, "norm": a.norm, "ms_median": float(np.median(ms)), "ms_min": float(np.min(ms)),
           "rows_per_s": a.rows / (np.median(ms) * 1e-3), "bytes_per_row": bpr, "GB/s": a.rows * bpr / (np.median(ms) * 1e-3) / 1e9,
           "hbm_frac": a.rows * bpr / (np.median(ms) * 1e-3) / 8e12}
    if a.hidden:
        ms3 = timed(lambda: qat.fc_model_forward(x, ws, ss, qts, a.norm, return_hidden=True, return_w_deq=True), a.steps, a.warmup)
        hb = 4 * sum(widths[1:-1])
        out["hidden_ms_median"] = float(np.median(ms3))
        out["hidden_bytes_per_row"] = bpr + hb
        out["hidden_hbm_frac"] = a.rows * (bpr + hb) / (np.median(ms3) * 1e-3) / 8e12
    if a.layerwise:
        def lw():
            h = x
            for l, w in enumerate(ws):
                h = qat.bitlinear_forward(h, w, ss[l], a.quant, a.norm)
                if l + 1 < len(ws):
                    h = torch.relu(h)
            return h
        ms2 = timed(lw, max(3, a.steps // 4), 1)
        out["layerwise_ms_median"] = float(np.median(ms2))
        out["speedup_vs_layerwise"] = float(np.median(ms2) / np.median(ms))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
