#!/usr/bin/env python3
"""Small batches are launch-bound: bnm_infer_device (counter memset + kernel) launched eagerly vs replayed from a captured HIP graph
of 1 and of 16 calls.  Microseconds per call, back to back, events around 200 repetitions."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bitnetmcu_amd as b          # noqa: E402
from bitnetmcu_amd import synth    # noqa: E402
import util                        # noqa: E402


def us_per_call(fn, calls_per_fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / (reps * calls_per_fn), 2)


def main():
    model = util.load_golden_model("fc_4bitsym_64")
    ctx = b.Context(model)
    res = {}
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for n in (64, 1024, 16384, 262144):
            x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
            synth.fill_device(x, first=0, dist=b.DIST_U)
            cls = torch.empty(n, dtype=torch.int32, device="cuda")
            ctx.infer_device(x, cls)
            side.synchronize()
            g1, g16 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, stream=side):
                ctx.infer_device(x, cls)
            with torch.cuda.graph(g16, stream=side):
                for _ in range(16):
                    ctx.infer_device(x, cls)
            res[n] = {"eager_us": us_per_call(lambda: ctx.infer_device(x, cls), 1), "graph_of_1_us": us_per_call(g1.replay, 1),
                      "graph_of_16_us": us_per_call(g16.replay, 16, reps=50)}
            print(n, res[n], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
