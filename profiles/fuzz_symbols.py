"""One-off fuzz of the reference's four kernel symbols as the GPU library exports them (group A of include/bitnetmcu_hip.h)
against the oracle's: random sizes far beyond the model shapes, every codec id, random and extreme data, in-place and
out-of-place.  python profiles/fuzz_symbols.py [seed] [rounds] -> summary on stdout."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))

import util                      # noqa: E402
import bitnetmcu_amd             # noqa: E402

FIELD_BITS = {1: 1, 2: 2, 4: 4, 12: 4, 20: 4, 16: 8}


def fuzz(seed, rounds):
    """-> (calls per symbol, list of mismatches)"""
    rng = np.random.default_rng(seed)
    ours, ref = util.Funcs(bitnetmcu_amd.load()), util.Funcs(util.load_oracle(), "orc_")
    bad = []
    counts = {"fc": 0, "relunorm": 0, "conv": 0, "pool": 0}
    for r in range(rounds):
        # ---- processfclayer: any codec id (unknown ones yield zeros), any size the packing allows
        bpw = int(rng.choice([1, 2, 4, 12, 16, 20, 64, 36, 3]))
        n_out = int(rng.integers(1, 400))
        if bpw == 64:
            n_in = 10 * int(rng.integers(1, 300))
            w = rng.integers(0, 65536, size=n_out * (n_in // 10), dtype=np.uint16)
        else:
            fb = FIELD_BITS.get(bpw, 4)
            n_in = (32 // fb) * int(rng.integers(1, 1 + 3000 * fb // 32))
            w = rng.integers(0, 2**32, size=n_out * (n_in * fb // 32), dtype=np.uint32)
        kind = r % 4
        act = (rng.integers(-128, 128, size=n_in) if kind == 0 else rng.integers(0, 128, size=n_in) if kind == 1
               else np.full(n_in, -128) if kind == 2 else np.full(n_in, 127)).astype(np.int8)
        a, b = ours.processfclayer(act, w, bpw, n_in, n_out), ref.processfclayer(act, w, bpw, n_in, n_out)
        counts["fc"] += 1
        if not np.array_equal(a, b):
            bad.append(("fc", bpw, n_in, n_out, kind))
        # ---- ReLUNorm: any length, magnitudes from a few units to 2^31, ties, all negative, in place
        n = int(rng.integers(1, 6000))
        mag = int(rng.choice([3, 127, 128, 255, 256, 1 << 12, 1 << 20, (1 << 31) - 1]))
        x = rng.integers(-mag, mag + 1, size=n, dtype=np.int64).astype(np.int32)
        if r % 7 == 0:
            x = -np.abs(x)
        if r % 11 == 0:
            x[rng.integers(0, n, size=3)] = x.max()
        for fn in ("relunorm", "relunorm_inplace"):
            (oa, pa), (ob, pb) = getattr(ours, fn)(x), getattr(ref, fn)(x)
            counts["relunorm"] += 1
            if pa != pb or not np.array_equal(oa, ob):
                d = np.nonzero(oa != ob)[0]
                bad.append((fn, n, mag, r, "argmax", pa, pb, "first differing inputs", x[d[:4]].tolist(), "ours", oa[d[:4]].tolist(),
                            "oracle", ob[d[:4]].tolist(), "max", int(x.max())))
        # ---- conv / pool: planes up to 96 x 96, shifts 0..12, in place and not
        xy = int(rng.integers(3, 97))
        plane = rng.integers(-(1 << int(rng.integers(1, 20))), 1 << int(rng.integers(1, 20)), size=xy * xy).astype(np.int32)
        wc = rng.integers(-128, 128, size=9).astype(np.int8)
        sh = int(rng.integers(0, 13))
        for inplace in (True, False):
            counts["conv"] += 1
            if not np.array_equal(ours.conv33(plane, wc, xy, sh, inplace), ref.conv33(plane, wc, xy, sh, inplace)):
                bad.append(("conv", xy, sh, inplace))
        xp = 2 * int(rng.integers(1, 49))
        pl = rng.integers(-(1 << 30), 1 << 30, size=xp * xp).astype(np.int32)
        for inplace in (True, False):
            counts["pool"] += 1
            if not np.array_equal(ours.maxpool22(pl, xp, inplace), ref.maxpool22(pl, xp, inplace)):
                bad.append(("pool", xp, inplace))
    return counts, bad


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    t0 = time.time()
    counts, bad = fuzz(seed, rounds)
    print(f"seed {seed}: {counts} calls compared with the oracle, {len(bad)} mismatches, {time.time() - t0:.1f} s", flush=True)
    for b_ in bad[:40]:
        print("  MISMATCH", b_, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
