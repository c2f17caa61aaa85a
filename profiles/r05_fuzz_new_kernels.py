#!/usr/bin/env python3
"""Round 5, one-off runs beyond the suite for the round's two new kernels (GPU box):
  1. cnn_li_fused_kernel (front end + FC tail in one kernel) against the channel kernel + tail launch (an independent implementation)
     on a MILLION images per channel count, full-range random conv kernels, ids AND logits compared one by one, the one-kernel
     form run twice (run-to-run identical);
  2. fused_fc_f32_kernel on seeded random FC models (a codec per layer out of all seven, widths to 128, 2..64 classes) against
     numpy quantisation (oracle/checker.quantize_input) + the oracle, 20,000 float images each incl. the edge rows.
usage: python profiles/r05_fuzz_new_kernels.py [n_float_models]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))


def main():
    import torch
    import bitnetmcu_amd as b
    import checker
    import test_gpu_parity as T
    bad = 0
    n = 1_000_000
    x = b.synth.images(3, n, 0)
    for C in (4, 8, 16, 23, 24, 32, 40, 48, 56, 61, 64):
        rng = np.random.default_rng(C)
        widths = (int(rng.choice([32, 64, 96])), int(rng.choice([16, 48, 64, 96])))
        ncls = int(rng.integers(2, 41))
        codecs = (16, int(rng.choice([1, 2, 4, 12, 16])), int(rng.choice([2, 4, 12, 16])))
        need = {1: 32, 2: 16, 4: 8, 12: 8, 16: 4}
        widths = tuple(max(need[codecs[k]], w // need[codecs[k]] * need[codecs[k]]) for k, w in zip((1, 2), widths))
        model = b.Model.from_header_text(T._random_cnn_text(rng, C, codecs, widths, ncls, lambda k: rng.integers(-128, 128, size=9 * C)))
        outs = {}
        for key, v in (("channel", 1), ("one kernel", 3), ("one kernel again", 3), ("two launches", 4)):
            ctx = b.Context(model)
            ctx.set_cnn_variant(v)
            assert ctx.cnn_tail_fused == (v == 3), (C, v)
            outs[key] = ctx.infer(x, logits=True)
            ctx.close()
        d = [int((outs[k][0] != outs["channel"][0]).sum()) + int((outs[k][1] != outs["channel"][1]).sum()) for k in ("one kernel", "one kernel again", "two launches")]
        # float images into the one-kernel form (cnn_li_fused_kernel<., true>): 200,000 images x 1/127 against quantise_input + the channel kernel
        nf = 200_000
        xf = x[:nf].astype(np.float32) * b.synth.FLOAT_PIXEL
        ctx = b.Context(model)
        ctx.set_cnn_variant(3)
        cls = torch.empty(nf, dtype=torch.int32, device="cuda")
        lg = torch.empty((nf, ncls), dtype=torch.int32, device="cuda")
        ctx.infer_float_device(torch.from_numpy(xf).cuda(), cls, lg)
        torch.cuda.synchronize()
        assert ctx.last_kernel == "cnn_li_fused_kernel<float>", ctx.last_kernel
        ctx.set_cnn_variant(1)
        want = ctx.infer(checker.quantize_input(xf), logits=True)
        ctx.close()
        d.append(int((cls.cpu().numpy().astype(np.uint32) != want[0]).sum()) + int((lg.cpu().numpy() != want[1]).sum()))
        bad += sum(d)
        print(f"CNN {C} channels, tail {widths}-{ncls}, codecs {codecs}: {n} images, ids + logits differing from the channel kernel's: "
              f"one kernel {d[0]}, again {d[1]}, two launches {d[2]}; float one-kernel form on {nf} images {d[3]}", flush=True)
    n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    for seed in range(n_models):
        rng = np.random.default_rng(880000 + seed)
        n_layers = int(rng.choice([3, 4]))
        codecs = tuple(int(c) for c in rng.choice([1, 2, 4, 12, 16, 20, 64], size=n_layers))
        need = {1: 32, 2: 16, 4: 8, 12: 8, 20: 8, 16: 4, 64: 8}
        widths = []
        for k in range(1, n_layers):
            g = need[codecs[k]]
            widths.append(int(rng.integers(1, int(rng.choice([32, 64, 128])) // g + 1)) * g)
        ncls = int(rng.integers(2, 65))
        model = b.Model.from_header_text(T._random_model_text(rng, codecs, tuple(widths), ncls))
        ctx = b.Context(model)
        xf = T._float_edge_rows(rng, 20000)
        xf[11] /= np.float32(512.0)               # (finite inputs by contract: the near-FLT_MAX row under the scale below)
        xf = xf * np.float32(rng.choice([1e-4, 1.0, 300.0]))
        want = checker.OracleModel(model).infer(checker.quantize_input(xf), logits=True)
        xd = torch.from_numpy(xf).cuda()
        cls = torch.empty(len(xf), dtype=torch.int32, device="cuda")
        lg = torch.empty((len(xf), ncls), dtype=torch.int32, device="cuda")
        ctx.infer_float_device(xd, cls, lg)
        torch.cuda.synchronize()
        d = int((cls.cpu().numpy().astype(np.uint32) != want[0]).sum()) + int((lg.cpu().numpy() != want[1]).sum())
        bad += d
        print(f"float model {seed}: codecs {codecs} widths {widths} classes {ncls} fused {ctx.float_fused} ({ctx.last_kernel}): differing ids + logits {d}", flush=True)
        ctx.close()
    print("TOTAL differing values:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
