"""The three statements of the synthetic workload generator must agree byte for byte: numpy
(bitnetmcu_amd/synth.py) vs host C (oracle/synth.h).  The HIP statement is checked in test_gpu_*.py."""
import numpy as np

from bitnetmcu_amd import synth, DIST_U, DIST_M, SEED_DIST_U, SEED_DIST_M


def test_numpy_vs_host_c(orc):
    for dist, seed in ((DIST_U, SEED_DIST_U), (DIST_M, SEED_DIST_M)):
        for first, count in ((0, 100), (12345, 17), (10**8 - 5, 5), (2**40, 3)):
            a = synth.images(first, count, dist)
            b = np.zeros((count, 256), np.int8)
            orc.orc_synth(seed, dist, first, count, b.ctypes.data)
            assert np.array_equal(a, b)


def test_counter_based():
    a = synth.images(0, 50, DIST_U)
    assert np.array_equal(a[10:20], synth.images(10, 10, DIST_U))


def test_dist_statistics():
    u = synth.images(0, 4000, DIST_U).astype(np.int32)
    assert u.min() == -128 and u.max() == 127 and abs(u.mean() + 0.5) < 0.5
    m = synth.images(0, 4000, DIST_M).astype(np.int32)
    assert m.min() == -20 and m.max() == 127
    bg = (m == -20).mean()
    assert 0.64 < bg < 0.69          # 66 % background (+ the 1/148 of the foreground that lands on -20)


def test_digest_matches_host_c(orc):
    cls = np.random.default_rng(3).integers(0, 10, size=5000).astype(np.uint32)
    hist = np.zeros(10, np.uint64)
    d = orc.orc_class_digest(cls.ctypes.data, 777, len(cls), hist.ctypes.data, 10)
    assert d == synth.class_digest(cls, 777)
    assert np.array_equal(hist, np.bincount(cls, minlength=10).astype(np.uint64))
