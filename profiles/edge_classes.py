"""One-off: class counts 1, 2, 3, 63, 64, 65, 100, 255, 256 (the maximum) on an FC, an all-ternary and a CNN model through every
path the model can run - ids, logits and ids-only against the oracle.  Run on the GPU box from the repo root."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import util, test_gpu_parity as t
import bitnetmcu_amd as b
from bitnetmcu_amd import synth
orc = util.load_oracle()
bad = []
for ncls in (1, 2, 3, 63, 64, 65, 100, 255, 256):
    for kind in ("fc", "cnn", "tern"):
        rng = np.random.default_rng(ncls * 7 + len(kind))
        try:
            if kind == "fc":
                text = t._random_model_text(rng, (4, 4, 4, 4), (64, 64, 64), ncls)
            elif kind == "tern":
                text = t._random_model_text(rng, (64, 64, 64, 64), (96, 96, 96), ncls)
            else:
                text = t._random_cnn_text(rng, 16, (4, 4, 4), (64, 32), ncls)
            model = b.Model.from_header_text(text)
        except Exception as e:
            print(kind, ncls, "load:", repr(e)[:120]); continue
        om = util.OracleModel(model, orc)
        x = np.concatenate([synth.images(1, 300, 0), synth.images(2, 301, 1), np.zeros((2, 256), np.int8)])
        want = om.infer(x, logits=True)
        ctx = b.Context(model)
        for label, setup in t.paths_for(ctx):
            try:
                setup(ctx)
                got = ctx.infer(x, logits=True)
                ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
                ids = np.array_equal(ctx.infer(x), want[0])
            except Exception as e:
                print(kind, ncls, label, "error:", repr(e)[:160]); ok = ids = True
            if not (ok and ids):
                bad.append((kind, ncls, label))
        ctx.close()
print("mismatches:", bad)
