"""One-off fuzz over batch sizes and entry points: for every zoo model, random n around the library's internal boundaries
(64-image zero-copy threshold, tiles of 32 / pairs of 64, the host path's 2^18-image staging chunks, the 2^20 / 2^22-image launch
chunks), class ids (+ logits every other time) through bnm_infer_host AND bnm_infer_device on the same images; the two plumbing
paths must agree everywhere and equal the oracle on the first / last 3,000 images and a strided sample.
python profiles/fuzz_sizes.py [seed] [rounds per model]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))

import util                      # noqa: E402
import bitnetmcu_amd as b        # noqa: E402
from bitnetmcu_amd import synth  # noqa: E402
from bitnetmcu_amd import model as M   # noqa: E402

EDGES = [1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, (1 << 18) - 1, 1 << 18, (1 << 18) + 1, (1 << 18) + 77,
         (1 << 19) + 5, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, (1 << 20) + 12345]


def main():
    import torch
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rng = np.random.default_rng(seed)
    orc = util.load_oracle()
    names = sorted(os.path.splitext(f)[0] for f in os.listdir(M.ZOO_DIR) if f.endswith(".bnm"))
    t0, bad, runs = time.time(), [], 0
    nmax = (1 << 22) + 4099
    pool = torch.empty((nmax, 256), dtype=torch.int8, device="cuda")
    synth.fill_device(pool, first=seed * 1_000_003, dist=seed & 1)
    host_pool = pool.cpu().numpy()
    for name in names:
        model = b.Model.from_zoo(name)
        om = util.OracleModel(model, orc)
        ctx = b.Context(model)
        sizes = [int(rng.choice(EDGES)) for _ in range(rounds - 1)] + [int(rng.integers(1, nmax))]
        if model.kind == b.KIND_CNN and rng.integers(0, 2):
            sizes.append((1 << 22) + int(rng.integers(1, 4099)))      # across the CNN front end's launch chunk
        for k, n in enumerate(sizes):
            off = int(rng.integers(0, nmax - n + 1))
            want_lg = bool(k & 1)
            x = pool[off:off + n]
            cls = torch.full((n,), -1, dtype=torch.int32, device="cuda")
            lg = torch.full((n, model.num_classes), -7, dtype=torch.int32, device="cuda") if want_lg else None
            ctx.infer_device(x, cls, lg)
            torch.cuda.synchronize()
            got = ctx.infer(host_pool[off:off + n], logits=want_lg)
            hcls = got[0] if want_lg else got
            runs += 1
            dcls = cls.cpu().numpy().astype(np.uint32)
            if not np.array_equal(dcls, hcls):
                bad.append((name, n, off, "device ids != host ids", int((dcls != hcls).sum())))
                continue
            if want_lg and not np.array_equal(lg.cpu().numpy(), got[1]):
                bad.append((name, n, off, "device logits != host logits"))
                continue
            idx = np.unique(np.concatenate([np.arange(min(n, 3000)), np.arange(max(0, n - 3000), n), np.linspace(0, n - 1, 2000).astype(np.int64)]))
            w = om.infer(host_pool[off:off + n][idx], logits=True)
            if not np.array_equal(hcls[idx], w[0]) or (want_lg and not np.array_equal(got[1][idx], w[1])):
                bad.append((name, n, off, "!= oracle on the sample"))
        ctx.close()
    print(f"seed {seed}: {runs} (model, n, offset) cases over {len(names)} models, {len(bad)} failures, {time.time() - t0:.1f} s", flush=True)
    for x in bad[:30]:
        print("  FAILED", x, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
