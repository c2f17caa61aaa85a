#!/usr/bin/env python3
"""Round 4: the lane = image CNN front end against the channel kernel (an independent implementation) on a MILLION images per
channel count, full-range random conv kernels, act bytes compared one by one, the lane = image kernel run twice (run-to-run
identical).  The rare-event check behind DESIGN.md 4.3a's hazard note; the suite runs 300,000 images at two channel counts.
usage (GPU box): python profiles/r04_cnn_li_large_sample.py [channels ...]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import bitnetmcu_amd as b
    import test_gpu_parity as T
    n = 1_000_000
    x = b.synth.images(3, n, 0)
    bad = 0
    for C in [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 40, 48, 56, 64, 80, 100, 128, 160]:
        rng = np.random.default_rng(C)
        model = b.Model.from_header_text(T._random_cnn_text(rng, C, (16, 4, 4), (96, 64), 10, lambda k: rng.integers(-128, 128, size=9 * C)))
        taps = {}
        for key, v in (("channel", 1), ("li", 3), ("li again", 3)):
            ctx = b.Context(model)
            ctx.set_cnn_variant(v)
            taps[key] = ctx.activations(x)[:, :4 * C]
            ctx.close()
        d1 = int((taps["li"] != taps["channel"]).sum())
        d2 = int((taps["li"] != taps["li again"]).sum())
        bad += d1 + d2
        print(f"{C} channels: {taps['li'].size} act bytes, {d1} differ from the channel kernel's, {d2} between two runs", flush=True)
    print("TOTAL differing bytes:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
