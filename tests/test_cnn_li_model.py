"""The lane = image formulation of the CNN front end (tests/cnn_li_model.py: three Toeplitz products per channel on
v_mfma_i32_32x32x32_i8, int8 planes, in-lane pooling, compressed records for the fused ReLUNorm) against the oracle's
conv / pool / ReLUNorm, without a GPU: the data flow the HIP kernel bnm_cnn_li.hip implements."""
import numpy as np
import pytest

import util
import cnn_li_model as li
from bitnetmcu_amd import synth, DIST_U, DIST_M


def oracle_front_end(f, images, w1, w2, w3):
    C = w1.shape[0]
    feats = np.zeros((len(images), 4 * C), np.int64)
    acts = np.zeros((len(images), 4 * C), np.int8)
    for n, image in enumerate(images):
        row = []
        for c in range(C):
            p = image.astype(np.int32)
            p = f.conv33(p, w1[c], 16, 4)
            p = f.conv33(p, w2[c], 14, 4)
            p = f.maxpool22(p, 12)
            p = f.conv33(p, w3[c], 6, 4)
            row.append(f.maxpool22(p, 4))
        feats[n] = np.concatenate(row)
        acts[n], _ = f.relunorm_inplace(np.concatenate(row))
    return feats, acts


@pytest.mark.parametrize("C,wmax,seed", [(3, 6, 0), (5, 127, 1), (2, 128, 2), (4, 1, 3)])
def test_lane_image_formulation_equals_the_oracle(C, wmax, seed, orc_funcs):
    rng = np.random.default_rng(seed)
    f = orc_funcs
    lo = -128 if wmax == 128 else -wmax
    w1, w2, w3 = (rng.integers(lo, min(wmax, 127) + 1, size=(C, 9)).astype(np.int8) for _ in range(3))
    if wmax == 128:
        w1[0], w2[0], w3[0] = -128, -128, -128          # the extreme sums, every tap
        w1[1], w2[1], w3[1] = 127, 127, 127
    x = np.concatenate([synth.images(seed, 20, DIST_U), synth.images(seed, 20, DIST_M), np.full((2, 256), -128, np.int8),
                        np.full((2, 256), 127, np.int8), np.zeros((1, 256), np.int8)])
    want_f, want_a = oracle_front_end(f, x, w1, w2, w3)
    got_f, (rec, mx) = li.front_end_features(x, w1, w2, w3)
    assert np.array_equal(got_f, want_f)
    assert np.array_equal(li.relunorm_from_records(rec, mx), want_a)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_pipelined_epilogue_equals_the_oracle_up_to_its_bound(seed, orc_funcs):
    """cnn_li_fused_pipe_kernel's stage 1 (accumulators from -32768, saturating int16 pack = ReLU, hi plane as (w >> 8) - 8) on
    models whose conv1 sums stay below 2^16 - two channels sit AT the bound (65,532 and 65,408, attained by the all-127 / all -128 image)."""
    rng = np.random.default_rng(100 + seed)
    f = orc_funcs
    C = 5
    w1 = rng.integers(-50, 51, size=(C, 9)).astype(np.int8)
    w1[0] = [58] * 8 + [52]
    w1[1] = [-57] * 8 + [-55]
    assert all(127 * int(w[w > 0].sum()) - 128 * int(w[w < 0].sum()) <= 65535 for w in w1.astype(np.int64))
    w2, w3 = (rng.integers(-128, 128, size=(C, 9)).astype(np.int8) for _ in range(2))
    x = np.concatenate([synth.images(seed, 20, DIST_U), synth.images(seed, 20, DIST_M), np.full((2, 256), -128, np.int8),
                        np.full((2, 256), 127, np.int8), np.zeros((1, 256), np.int8)])
    want_f, want_a = oracle_front_end(f, x, w1, w2, w3)
    got_f, (rec, mx) = li.front_end_features(x, w1, w2, w3, pipelined=True)
    assert np.array_equal(got_f, want_f)
    assert np.array_equal(li.relunorm_from_records(rec, mx), want_a)


def test_toeplitz_fragments_are_int8_and_translation_invariant():
    w = np.array([1, -2, 3, -4, 5, -6, 7, -8, 9], np.int8)
    for T in (li.toeplitz_conv1(w), li.toeplitz_conv2(w), li.toeplitz_conv3(w)):
        assert T.shape == (32, 64) and np.abs(T).max() <= 9 and (T != 0).sum() in (2 * 14 * 9, 24 * 9, 16 * 9)
