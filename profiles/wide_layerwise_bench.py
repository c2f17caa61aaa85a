#!/usr/bin/env python3
"""Models outside the fused kernels (a layer wider than 256 outputs): the layer-wise MFMA path (one int8 GEMM kernel per layer +
ReLUNorm kernel, sums through HBM) against the bit-serial layer-wise kernels it replaces as the fallback.  Random weights;
class ids of the two paths are compared on the timed images.  GPU box."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def model_text(rng, widths, n_classes=10, bpw=4):
    lines = ["#include <stdint.h>", "#define MODEL_FCMNIST", "#define NUM_LAYERS %d" % (len(widths) + 1), "#define MAX_N_ACTIVATIONS 256"]
    n_in = 256
    for k, n_out in enumerate(list(widths) + [n_classes], start=1):
        w = rng.integers(0, 2**32, size=n_out * (n_in * bpw // 32), dtype=np.uint32)
        lines += [f"#define L{k}_active", f"#define L{k}_bitperweight {bpw}", f"#define L{k}_incoming_weights {n_in}",
                  f"#define L{k}_outgoing_weights {n_out}", f"const uint32_t L{k}_weights[] = {{" + ",".join(hex(int(x)) for x in w) + "};"]
        n_in = n_out
    return "\n".join(lines) + "\n"


def main():
    import torch
    import bitnetmcu_amd as b
    os.environ["BNM_QUIET"] = "1"
    rng = np.random.default_rng(5)
    n = 4_000_000
    x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(x, first=0, dist=0)
    out = {}
    for widths in ((320, 64, 64), (512, 512, 512)):
        model = b.Model.from_header_text(model_text(rng, widths))
        res = {}
        ids = {}
        for label, path, cnt in (("layerwise_mfma", b.PATH_LAYERWISE_MFMA, n), ("layerwise_bit_serial", b.PATH_LAYERWISE_ALU, 400_000)):
            ctx = b.Context(model)
            ctx.set_path(path)
            cls = torch.empty(cnt, dtype=torch.int32, device="cuda")
            ctx.infer_device(x[:cnt], cls)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            for k in range(3):
                ctx.infer_device(x[:cnt], cls)
                ev[k + 1].record()
            torch.cuda.synchronize()
            ms = min(ev[k].elapsed_time(ev[k + 1]) for k in range(3))
            res[label] = {"images": cnt, "ms": ms, "inferences_per_s": cnt / (ms * 1e-3)}
            ids[label] = cls[:400_000].clone()
            ctx.close()
        assert torch.equal(ids["layerwise_mfma"], ids["layerwise_bit_serial"])
        res["speedup"] = res["layerwise_mfma"]["inferences_per_s"] / res["layerwise_bit_serial"]["inferences_per_s"]
        out["-".join(map(str, widths))] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
