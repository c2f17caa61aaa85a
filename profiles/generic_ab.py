#!/usr/bin/env python3
"""Same-process, interleaved A/B of the generic fused kernel's forms on one resident image set (GPU box):
one / two image tiles per wave (variants 7 / 8), batches per take, against the specialised dual kernel and the box's plain
read rate.  Usage: python profiles/generic_ab.py [--images N] [--models a,b,...] [--rounds R]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=100_000_000)
    ap.add_argument("--models", default="fc_4bitsym_64,doc12k_8bit,tern_96,doc12k_2bit,doc12k_ternary,doc12k_binary")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--batches", default="0")
    a = ap.parse_args()
    import torch
    import bitnetmcu_amd as b
    n = a.images
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def timed(fn, reps=3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for k in range(reps):
            fn()
            ev[k + 1].record()
        torch.cuda.synchronize()
        return [ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]

    out = {"images": n}
    for name in a.models.split(","):
        model = b.Model.from_zoo(name)
        arms = {}
        ctxs = {}
        for label, variant in (("default", -1), ("generic_t1", 7), ("generic_t2", 8), ("regw", 9)):
            for wb in [int(v) for v in a.batches.split(",")]:
                ctx = b.Context(model)
                try:
                    if variant >= 0:
                        ctx.set_tuning(variant=variant)
                    if wb:
                        ctx.set_work_batch(wb)
                except b.BnmError:
                    ctx.close()
                    continue
                ctxs[f"{label}_b{wb}" if wb else label] = ctx
        ref = None
        for ctx in ctxs.values():          # warm-up + cross-check of the forms against each other
            ctx.infer_device(x, cls)
            d = b.synth.digest_device(cls, 0, model.num_classes).cpu().numpy()
            ref = d if ref is None else ref
            assert np.array_equal(d, ref), name
        arms = {k: [] for k in ctxs}
        arms["stream_read"] = []
        for _ in range(a.rounds):
            for k, ctx in ctxs.items():
                arms[k] += timed(lambda: ctx.infer_device(x, cls))
            arms["stream_read"] += timed(lambda: b.synth.stream_read_device(x, sink))
        out[name] = {k: {"median_ms": float(np.median(v)), "min_ms": float(np.min(v)), "variant": (ctxs[k].variant if k in ctxs else None)}
                     for k, v in arms.items()}
        print(name, json.dumps(out[name]), flush=True)
        for ctx in ctxs.values():
            ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
