#!/usr/bin/env python3
"""The default headline kernel next to plain streaming loads of the SAME 25.6 GB on the same box, same process, interleaved
(needs the diagnostic library for the streaming kernel: python bitnetmcu_amd/build.py --diag;
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so python profiles/ceiling_ab.py).
plain stream: bnm_diag_stream_device mode 0 (16 B/lane loads of every image byte, nothing else; 256 B per image);
kernel: the whole model, 260 B per image."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bitnetmcu_amd as b
from bitnetmcu_amd import _lib as L, synth
import util


def main():
    lib = b.load()
    if not hasattr(lib, "bnm_diag_stream_device"):
        sys.exit("needs the diagnostic library (see the docstring)")
    n = int(os.environ.get("N", 100_000_000))
    rounds, launches = int(os.environ.get("ROUNDS", 10)), 5
    model = util.load_golden_model("fc_4bitsym_64")
    ctx = b.Context(model)
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    synth.fill_device(imgs, first=0, dist=b.DIST_U)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    arms = {"plain_stream": lambda: L.check(lib, lib.bnm_diag_stream_device(imgs.data_ptr(), n, 0, 0, out.data_ptr(), s)),
            "kernel": lambda: ctx.infer_device(imgs, cls)}
    ms = {k: [] for k in arms}
    for f in arms.values():
        f()
    torch.cuda.synchronize()
    for r in range(rounds):
        for k in (list(arms) if r % 2 == 0 else list(arms)[::-1]):
            for _ in range(launches):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                arms[k]()
                e1.record()
                e1.synchronize()
                ms[k].append(e0.elapsed_time(e1))
    res = {k: {"median_ms": float(np.median(v)), "min_ms": float(np.min(v))} for k, v in ms.items()}
    res["plain_stream"]["TB/s"] = n * 256 / res["plain_stream"]["median_ms"] / 1e9
    res["kernel"]["TB/s"] = n * 260 / res["kernel"]["median_ms"] / 1e9
    res["kernel"]["variant"] = ctx.variant
    res["kernel_over_plain_stream_bytes_per_s"] = res["kernel"]["TB/s"] / res["plain_stream"]["TB/s"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
