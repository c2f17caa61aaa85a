#!/usr/bin/env python3
"""Generate the committed golden fixtures from the COMPILED REFERENCE (oracle/_ref/*/Bitnet_inf.dll,
built by oracle/build_oracle.py from the unmodified sources under /root/reference).

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
Outputs (small, committed; consumed on the GPU box where /root/reference does not exist):
  real_images.npz          the reference's 13 embedded MNIST images + labels
                           (BitNetMCU_MNIST_test_data.h:1-190, mcu/BitNetMCUdemo.c:23-28)
  ../../bitnetmcu_amd/zoo/<name>.bnm   BNMBLOB1 form of every model header (parsed by the product loader and
                           cross-checked here word-for-word against the DLL's own Lk_weights symbols)
  kat_<name>.npz           per model: inputs, class ids from Inference(), logits and all int8
                           activations from the DLL's own processfclayer/ReLUNorm/conv/pool
  kat_codecs.npz           processfclayer on random layers for every codec id incl. unknown ones
  kat_relunorm.npz         ReLUNorm edge cases (ties, all-negative, rounding overflow, in-place)
  kat_convpool.npz         processconv33ReLU / processmaxpool22 on random planes
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402
from util import REPO, REF_DIR, Funcs, run_schedule, ref_dll_path, parse_c_int8_arrays  # noqa: E402
sys.path.insert(0, os.path.join(REPO, "oracle"))
from build_oracle import REF_MODELS  # noqa: E402
from bitnetmcu_amd import Model, synth, DIST_U, DIST_M  # noqa: E402


def real_images():
    a, la = parse_c_int8_arrays(open(os.path.join(REF_DIR, "BitNetMCU_MNIST_test_data.h")).read())
    b, lb = parse_c_int8_arrays(open(os.path.join(REF_DIR, "mcu", "BitNetMCUdemo.c")).read())
    assert a.shape == (10, 256) and b.shape == (3, 256)
    assert la.tolist() == [3, 2, 0, 9, 0, 6, 9, 2, 7, 7] and lb.tolist() == [7, 1, 9]
    return np.concatenate([a, b]), np.concatenate([la, lb])


def kat_inputs(real):
    edge = np.zeros((6, 256), np.int8)
    edge[1] = -128
    edge[2] = 127
    edge[3, ::2] = 127
    edge[3, 1::2] = -128
    edge[4, 100] = 1
    edge[5] = -20
    return np.concatenate([real, synth.images(0, 64, DIST_U), synth.images(0, 64, DIST_M),
                           synth.images(10**8 - 32, 32, DIST_U), edge])


def main(only=None):
    """only: model names - (re)generate just those models' blobs and KATs and leave every other fixture untouched"""
    rng = np.random.default_rng(20240421)
    real, labels = real_images()
    if not only:
        np.savez_compressed(os.path.join(HERE, "real_images.npz"), images=real, labels=labels)
    x = kat_inputs(real)

    for name, hdr in REF_MODELS.items():
        if only and name not in only:
            continue
        model = Model.from_header(hdr)
        dll = C.CDLL(ref_dll_path(name))
        dll.Inference.restype = C.c_uint32
        dll.Inference.argtypes = [C.POINTER(C.c_int8)]
        # loader vs the C compiler: every array word for word
        for i, li in enumerate(model.layers()):
            if li.weight_count == 0:
                continue
            w = model.layer_weights(i)
            ct = {4: C.c_uint32, 2: C.c_uint16, 1: C.c_int8}[li.weight_elem_bytes]
            sym = (ct * li.weight_count).in_dll(dll, f"L{li.order}_weights")
            assert np.array_equal(np.ctypeslib.as_array(sym), w), (name, li.order)
        with open(os.path.join(REPO, "bitnetmcu_amd", "zoo", name + ".bnm"), "wb") as f:
            f.write(model.to_blob())
        f_ref = Funcs(dll)
        # Reference defect: the x86 CNN wrapper sizes its scratch `int32_t layer_out[MAX_N_ACTIVATIONS]`
        # (BitNetMCU_MNIST_dll.c:49) but copies the 256-pixel image into it (:68-70); headers with
        # MAX_N_ACTIVATIONS < 256 (cnn_16/16small/32/48) overflow the stack ("stack smashing detected").
        # mcu/BitNetMCUdemo.c:35 fixes the size to 256.  For those models the class ids come from the DLL's own
        # four kernels driven through the wrapper's schedule with correctly sized buffers.
        wrapper_ok = not (model.kind == 1 and model.layer(0).out_channels * 4 < 256)
        # the expensive per-function trace only on a subset for CNN models (64 channels x 5 calls per image)
        trace_n = len(x) if (model.kind == 0 or not wrapper_ok) else 24
        if wrapper_ok:
            cls_inf = np.array([dll.Inference(row.ctypes.data_as(C.POINTER(C.c_int8))) for row in x], np.uint32)
        else:
            cls_inf = None
        cls, logits, acts = [], [], []
        for row in x[:trace_n]:
            c, lg, a = run_schedule(f_ref, model, row)
            cls.append(c), logits.append(lg), acts.append(a)
        if cls_inf is None:
            cls_inf = np.array(cls, np.uint32)
        assert np.array_equal(np.array(cls, np.uint32), cls_inf[:trace_n]), name
        if name in ("fc_4bitsym_64", "mcu_12k"):
            # SURVEY.md §4 seed KATs (probed from the compiled reference during the survey)
            assert logits[0].tolist() == [-3277, -1343, -1315, 2957, -2401, -685, -3871, -929, -1771, 177]
            assert logits[1].tolist() == [-1239, -1467, 2861, -861, -277, -1797, -1653, -439, -1535, -1483]
        if model.num_classes == 10:
            print(f"{name:18s} real-image accuracy {int((cls_inf[:13] == labels).sum())}/13")
        np.savez_compressed(os.path.join(HERE, f"kat_{name}.npz"), images=x, cls=cls_inf, trace_n=trace_n,
                            logits=np.array(logits, np.int32), acts=np.array(acts, np.int8))

    if only:
        return
    # ---- per-codec layers ------------------------------------------------------------------------------
    f_ref = Funcs(C.CDLL(ref_dll_path("cnn_64")))   # a CNN build exports all four kernels
    codec = {}
    cases = [(1, 256, 40), (1, 64, 7), (2, 256, 96), (2, 16, 16), (4, 256, 64), (4, 64, 10), (4, 96, 64), (12, 128, 33),
             (16, 64, 20), (16, 256, 64), (20, 256, 64), (20, 64, 10), (64, 260, 96), (64, 100, 10), (64, 20, 5),
             (36, 64, 12), (8, 64, 12), (0, 32, 4)]
    for k, (bpw, n_in, n_out) in enumerate(cases):
        if bpw == 64:
            w = rng.integers(0, 65536, size=n_out * (n_in // 10), dtype=np.uint16)
        else:
            fb = {1: 1, 2: 2, 4: 4, 12: 4, 20: 4, 16: 8}.get(bpw, 4)
            w = rng.integers(0, 2**32, size=n_out * (n_in * fb // 32), dtype=np.uint32)
        for tag, act in (("s", rng.integers(-128, 128, size=n_in, dtype=np.int8)),
                         ("u", rng.integers(0, 128, size=n_in, dtype=np.int8)),
                         ("x", np.full(n_in, -128, np.int8))):
            out = f_ref.processfclayer(act, w, bpw, n_in, n_out)
            codec[f"c{k}{tag}_act"], codec[f"c{k}{tag}_out"] = act, out
        codec[f"c{k}_w"], codec[f"c{k}_meta"] = w, np.array([bpw, n_in, n_out])
    codec["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "kat_codecs.npz"), **codec)

    # ---- ReLUNorm edge cases -------------------------------------------------------------------------
    vecs = []
    for n in (10, 16, 37, 64, 96, 256):
        for scale in (1, 100, 127, 128, 129, 255, 256, 1000, 32768, 500000, 4_000_000):
            v = rng.integers(-scale, scale + 1, size=n).astype(np.int32)
            vecs.append(v)
        vecs.append(np.full(n, -5, np.int32))                    # all negative
        vecs.append(np.zeros(n, np.int32))                       # all zero
        t = rng.integers(-50, 50, size=n).astype(np.int32); t[[1, n - 1]] = 77; vecs.append(t)   # tie: first wins
        for top in (127, 128, 254, 255, 256, 509, 510, 511, 65407, 65408, 32640):                  # rounding -> clip
            t = rng.integers(-top, top + 1, size=n).astype(np.int32); t[n // 2] = top; vecs.append(t)
        t = np.full(n, np.iinfo(np.int32).min + 1, np.int32); vecs.append(t)                      # == -INT32_MAX
    rn = {"n_cases": np.array(len(vecs))}
    for k, v in enumerate(vecs):
        out, pos = f_ref.relunorm(v)
        out2, pos2 = f_ref.relunorm_inplace(v)
        assert np.array_equal(out, out2) and pos == pos2
        rn[f"in{k}"], rn[f"out{k}"], rn[f"pos{k}"] = v, out, np.array(pos)
    np.savez_compressed(os.path.join(HERE, "kat_relunorm.npz"), **rn)

    # ---- conv / pool ---------------------------------------------------------------------------------
    cp = {}
    k = 0
    for xy, scale in ((16, 128), (14, 9216), (6, 663552), (5, 1000), (3, 50)):
        for shift in (4, 0, 11):
            if shift != 4 and xy not in (16, 5):
                continue
            plane = rng.integers(-scale if xy == 16 else 0, scale + 1, size=xy * xy).astype(np.int32)
            w = rng.integers(-128, 128, size=9, dtype=np.int8)
            cp[f"conv{k}_in"], cp[f"conv{k}_w"], cp[f"conv{k}_meta"] = plane, w, np.array([xy, shift])
            cp[f"conv{k}_out"] = f_ref.conv33(plane, w, xy, shift)
            assert np.array_equal(cp[f"conv{k}_out"], f_ref.conv33(plane, w, xy, shift, inplace=False))
            k += 1
    cp["n_conv"] = np.array(k)
    k = 0
    for xy in (12, 4, 2, 8):
        plane = rng.integers(-10**6, 10**6, size=xy * xy).astype(np.int32)
        cp[f"pool{k}_in"], cp[f"pool{k}_meta"] = plane, np.array([xy])
        cp[f"pool{k}_out"] = f_ref.maxpool22(plane, xy)
        k += 1
    cp["n_pool"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "kat_convpool.npz"), **cp)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    if not util.have_reference():
        sys.exit("needs /root/reference and oracle/_ref (python oracle/build_oracle.py)")
    main(sys.argv[1:] or None)
