#!/usr/bin/env python3
"""Same-process, interleaved A/B of the headline workload (fc_4bitsym_64, 1e8 Dist-U images resident): one context per
candidate, ROUNDS rounds of (every candidate: LAUNCHES timed launches), so that the box's thermal / power drift hits all
candidates alike.  Candidates: VARIANT[:WORK_BATCH] ...   e.g.  python profiles/headline_ab.py 3 5 6:8 6:16 6:32 4:16
Prints per candidate the median / min / mean ms over all its launches and the digest check."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bitnetmcu_amd as b
from bitnetmcu_amd import synth
import util

ORACLE_DIGEST_1E8 = 0x81b56c9fafee6636


def main():
    specs = sys.argv[1:] or ["3", "5", "6:8", "6:16", "4:16"]
    n = int(os.environ.get("N", 100_000_000))
    rounds, launches = int(os.environ.get("ROUNDS", 12)), int(os.environ.get("LAUNCHES", 5))
    model = util.load_golden_model(os.environ.get("MODEL", "fc_4bitsym_64"))
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    synth.fill_device(imgs, first=0, dist=b.DIST_U)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    ctxs = []
    for s in specs:
        v, _, wb = s.partition(":")
        c = b.Context(model)
        c.set_tuning(variant=int(v))
        if wb:
            c.set_work_batch(int(wb))
        ctxs.append(c)
    ms = {s: [] for s in specs}
    ok = {}
    for s, c in zip(specs, ctxs):       # warm-up + digest
        cls.fill_(-1)
        c.infer_device(imgs, cls)
        d = synth.digest_device(cls, first=0, n_bins=model.num_classes).cpu().numpy()
        ok[s] = hex(int(d[0].astype(np.uint64)))
    for r in range(rounds):
        order = list(range(len(specs)))
        if r & 1:
            order.reverse()
        for i in order:
            for _ in range(launches):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctxs[i].infer_device(imgs, cls)
                e1.record()
                e1.synchronize()
                ms[specs[i]].append(e0.elapsed_time(e1))
    out = {s: {"median_ms": float(np.median(v)), "min_ms": float(np.min(v)), "mean_ms": float(np.mean(v)), "launches": len(v),
               "digest": ok[s], "digest_ok": (int(ok[s], 16) == ORACLE_DIGEST_1E8) if n == 100_000_000 and "fc_4bitsym_64" in model_name() else None}
           for s, v in ms.items()}
    print(json.dumps(out, indent=1))


def model_name():
    return os.environ.get("MODEL", "fc_4bitsym_64")


if __name__ == "__main__":
    main()
