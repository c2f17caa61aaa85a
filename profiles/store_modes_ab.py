#!/usr/bin/env python3
"""Which property of the class-id stores costs the headline kernel 0.5 ms on some boxes (r02/slow_state_probe_r02.log: 4.46 ms with
the stores, 3.97 ms without, plain stream 3.93 ms)?  Same process, round-robin over store flavours of the DIAGNOSTIC kernel
(bits 56..63 of bnm_diag_set_src_wrap; timing only - every flavour but 0 leaves wrong class ids):
  0 nontemporal dword store per pair (the product's) | 1 plain store | 2 sc0 sc1 | 3 same store, 256 KiB destination (no HBM writes) |
  4 byte stores (a quarter of the bytes) | 5 one wider store per batch of 2 / 4 pairs | 6 scalar-unit stores | 255 none
    python bitnetmcu_amd/build.py --diag; BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so python profiles/store_modes_ab.py"""
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
    if not hasattr(lib, "bnm_diag_stream_device"):
        sys.exit("needs the diagnostic library (see the docstring)")
    n = int(os.environ.get("N", 100_000_000))
    rounds = int(os.environ.get("ROUNDS", 8))
    model = util.load_golden_model("fc_4bitsym_64")
    ctx = b.Context(model)
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    synth.fill_device(imgs, first=0, dist=b.DIST_U)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def arm(mode, batch=2):
        def f():
            ctx.set_work_batch(batch)
            L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, mode << 56))
            ctx.infer_device(imgs, cls)
            L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, 0))
            ctx.set_work_batch(2)
        return f

    arms = {"0_nontemporal": arm(0), "1_plain": arm(1), "2_sc0_sc1": arm(2), "3_small_destination": arm(3), "4_byte_stores": arm(4),
            "5_one_store_per_2_pairs": arm(5, 2), "0_nontemporal_batch4": arm(0, 4), "5_one_store_per_4_pairs": arm(5, 4),
            "6_scalar_stores": arm(6), "255_no_store": arm(255),
            "plain_stream": lambda: L.check(lib, lib.bnm_diag_stream_device(imgs.data_ptr(), n, 0, 0, cls.data_ptr(), s))}
    ms = {k: [] for k in arms}
    for f in arms.values():
        f()
    torch.cuda.synchronize()
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
        print(f"{k:28s} median {v['median_ms']:.3f}  min {v['min_ms']:.3f}", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
