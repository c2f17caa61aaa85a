#!/usr/bin/env python3
"""Round 4: the reference's entry points from 1 / 2 / 4 / 8 host threads at once (they are re-entrant in the reference; here
every call leases a context or a staging slot from a pool).  Inference() on the model-bound DLLs and the processfclayer +
ReLUNorm pair on the library; calls per second, and every answer checked against the single-threaded run.
usage (GPU box): python profiles/r04_reentrancy.py"""
import json
import os
import sys
import threading
import time
from ctypes import POINTER, c_int8, c_int32, c_uint32, c_void_p, cast

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run_threads(n_threads, work):
    ts = [threading.Thread(target=work, args=(t, n_threads)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return time.perf_counter() - t0


def main():
    import bitnetmcu_amd as b
    from bitnetmcu_amd import harness, synth
    out = {}
    for name in ("fc_4bitsym_64", "cnn_64"):
        dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", name, "Bitnet_inf.dll")
        if not os.path.isfile(dll):
            continue
        lib = harness.load_inference_dll(dll)
        x = synth.images(5, 8000, 0)
        bufs = [(c_int8 * 256)(*row.tolist()) for row in x]
        ref = np.array([lib.Inference(bf) for bf in bufs], np.uint32)
        row = {}
        for T in (1, 2, 4, 8):
            got = np.zeros(len(bufs), np.uint32)

            def work(t, n):
                for i in range(t, len(bufs), n):
                    got[i] = lib.Inference(bufs[i])
            dt = run_threads(T, work)
            assert np.array_equal(got, ref), (name, T)
            row[T] = len(bufs) / dt
        out["Inference " + name] = row
        print(name, json.dumps(row), flush=True)
    # the kernel symbols: processfclayer (256 -> 64, 4-bit) + ReLUNorm per "call"
    lib = b.load()
    lib.processfclayer.argtypes = [POINTER(c_int8), POINTER(c_uint32), c_int32, c_uint32, c_uint32, POINTER(c_int32)]
    lib.processfclayer.restype = None
    lib.ReLUNorm.argtypes = [POINTER(c_int32), POINTER(c_int8), c_uint32]
    lib.ReLUNorm.restype = c_uint32
    rng = np.random.default_rng(1)
    w = rng.integers(0, 2**32, size=64 * 32, dtype=np.uint32)
    acts = rng.integers(-128, 128, size=(2000, 256)).astype(np.int8)
    wp = w.ctypes.data_as(POINTER(c_uint32))

    pa = [acts[i].ctypes.data_as(POINTER(c_int8)) for i in range(len(acts))]      # (pointer objects made once: the loop is the two calls)

    def pointers(sums, outs):
        return ([sums[i].ctypes.data_as(POINTER(c_int32)) for i in range(len(acts))],
                [outs[i].ctypes.data_as(POINTER(c_int8)) for i in range(len(acts))])

    def one(i, ps, po):
        lib.processfclayer(pa[i], wp, 4, 256, 64, ps[i])
        lib.ReLUNorm(ps[i], po[i], 64)
    ref_s, ref_o = np.zeros((len(acts), 64), np.int32), np.zeros((len(acts), 64), np.int8)
    ps, po = pointers(ref_s, ref_o)
    for i in range(len(acts)):
        one(i, ps, po)
    row = {}
    for T in (1, 2, 4, 8):
        s, o = np.zeros_like(ref_s), np.zeros_like(ref_o)
        ps, po = pointers(s, o)

        def work(t, n):
            for i in range(t, len(acts), n):
                one(i, ps, po)
        dt = run_threads(T, work)
        assert np.array_equal(o, ref_o) and np.array_equal(s, ref_s), T
        row[T] = 2 * len(acts) / dt
    out["processfclayer + ReLUNorm (symbol calls per second)"] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
