#!/usr/bin/env python3
"""Round 4: the lane = image CNN front end (cnn_variant 3) against the oracle and against the channel kernel (cnn_variant 1):
class ids, logits and timing.  usage (GPU box): python profiles/r04_cnn_li_check.py [model ...]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))


def main():
    import torch
    import bitnetmcu_amd as b
    import checker
    names = sys.argv[1:] or ["cnn_64", "mcu_cnn_16", "mcu_cnn_32", "mcu_cnn_48", "mcu_cnn_16small"]
    out = {}
    for name in names:
        model = b.Model.from_zoo(name)
        orc = checker.OracleModel(model)
        for n in (1, 31, 33, 1000, 4097):
            x = b.synth.images(3, n, 0)
            want = orc.infer(x, logits=True)
            ctx = b.Context(model)
            ctx.set_cnn_variant(3)
            got = ctx.infer(x, logits=True)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (name, n)
            ctx.close()
        n = 10_000_000
        x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
        b.synth.fill_device(x, first=0, dist=0)
        res = {}
        ref = None
        for variant in (1, 3, 301, 304):
            ctx = b.Context(model)
            ctx.set_cnn_variant(variant)
            cls = torch.empty(n, dtype=torch.int32, device="cuda")
            ctx.infer_device(x, cls)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            for k in range(3):
                ctx.infer_device(x, cls)
                ev[k + 1].record()
            torch.cuda.synchronize()
            ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]
            d = b.synth.digest_device(cls, 0, model.num_classes).cpu().numpy()
            ref = d if ref is None else ref
            assert np.array_equal(d, ref), (name, variant, "digest differs from the channel kernel's")
            res[variant] = {"ms": float(np.median(ms)), "inferences_per_s": n / (float(np.median(ms)) * 1e-3)}
            ctx.close()
        out[name] = res
        print(name, json.dumps(res), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
