#!/usr/bin/env python3
"""The headline kernel WITH logits (300 B per image): how should the 40 bytes per image leave?  Same process, round-robin over the
diagnostic kernel's logits-store flavours (bits 48..55 of bnm_diag_set_src_wrap):
  0 whole tiles through LDS, nontemporal 16 B/lane stores (the product's) | 1 the same with plain stores |
  2 16-byte pieces at a 40-byte stride straight from the accumulators, plain (round 2's first form)
    python bitnetmcu_amd/build.py --diag; BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so python profiles/logits_store_ab.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bitnetmcu_amd as b                   # noqa: E402
from bitnetmcu_amd import _lib as L, synth  # noqa: E402
import util                                 # noqa: E402


def main():
    lib = b.load()
    if not hasattr(lib, "bnm_diag_set_src_wrap"):
        sys.exit("needs the diagnostic library (see the docstring)")
    n = int(os.environ.get("N", 100_000_000))
    rounds = int(os.environ.get("ROUNDS", 8))
    model = util.load_golden_model("fc_4bitsym_64")
    ctx = b.Context(model)
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    synth.fill_device(imgs, first=0, dist=b.DIST_U)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    lg = torch.empty((n, 10), dtype=torch.int32, device="cuda")

    def arm(mode):
        def f():
            L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, mode << 48))
            ctx.infer_device(imgs, cls, lg)
            L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, 0))
        return f

    arms = {"0_tiles_nontemporal": arm(0), "1_tiles_plain": arm(1), "2_pieces_plain": arm(2), "ids_only": lambda: ctx.infer_device(imgs, cls)}
    # the three flavours must write the same logits
    ref = None
    for k in list(arms)[:3]:
        lg.zero_()
        arms[k]()
        torch.cuda.synchronize()
        d = (int(lg.sum(dtype=torch.int64)), int((lg.to(torch.int64) * torch.arange(1, 11, device="cuda")).sum()))
        ref = ref or d
        assert d == ref, (k, d, ref)
    ms = {k: [] for k in arms}
    names = list(arms)
    for r in range(rounds):
        for k in (names if r % 2 == 0 else names[::-1]):
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                arms[k]()
                e1.record()
                e1.synchronize()
                ms[k].append(e0.elapsed_time(e1))
    res = {k: {"median_ms": round(float(np.median(v)), 3), "min_ms": round(float(np.min(v)), 3)} for k, v in ms.items()}
    for k, v in res.items():
        print(f"{k:24s} median {v['median_ms']:.3f}  min {v['min_ms']:.3f}", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
