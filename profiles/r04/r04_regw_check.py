#!/usr/bin/env python3
"""Round 4: the register-resident-weight kernel (variant 9) against the oracle and against the generic kernel (variant 4) on the
same images.  Usage: python profiles/r04_regw_check.py [model ...]   (GPU box)"""
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
    names = sys.argv[1:] or ["tern_96"]
    for name in names:
        model = b.Model.from_zoo(name)
        orc = checker.OracleModel(model)
        for n in (64 * 1024 * 3, 64 * 1024 * 9 + 37, 1_000_003):
            x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
            b.synth.fill_device(x, first=0, dist=0)
            out = {}
            for variant in (9, 4):
                ctx = b.Context(model)
                ctx.set_tuning(variant=variant)
                cls = torch.full((n,), -1, dtype=torch.int32, device="cuda")
                lg = torch.full((n, model.num_classes), -7, dtype=torch.int32, device="cuda")
                ctx.infer_device(x, cls, lg)
                cls2 = torch.full((n,), -1, dtype=torch.int32, device="cuda")
                ctx.infer_device(x, cls2)
                torch.cuda.synchronize()
                assert torch.equal(cls, cls2), (name, n, variant, "ids differ between the logits and the ids-only call")
                out[variant] = (cls.cpu().numpy(), lg.cpu().numpy())
                assert ctx.variant == variant
                ctx.close()
            assert np.array_equal(out[9][0], out[4][0]), (name, n, "ids: regw != generic")
            assert np.array_equal(out[9][1], out[4][1]), (name, n, "logits: regw != generic")
            m = min(n, 200_000)
            want_cls, want_lg = orc.infer(x[:m].cpu().numpy(), logits=True)
            assert np.array_equal(out[9][0][:m].astype(np.uint32), want_cls), (name, n, "ids != oracle")
            assert np.array_equal(out[9][1][:m], want_lg), (name, n, "logits != oracle")
            print(f"{name}: n={n}: variant 9 == variant 4 (ids + logits, all), == oracle on the first {m}", flush=True)


if __name__ == "__main__":
    main()
