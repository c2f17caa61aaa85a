"""Torchvision-free restatement of the reference's DLL cross-check loop (test_inference.py:134-175).

The reference harness loads ``./Bitnet_inf.dll``, quantises one MNIST image at a time in Python and
calls ``lib.Inference(ptr)``.  MNIST/torchvision are not available offline, so the
image source is a parameter here; everything at the FFI boundary is the reference's: the artifact
name, ``argtypes=[POINTER(c_int8)]``, ``restype=c_uint32``, one synchronous call per image.
"""
from ctypes import CDLL, POINTER, c_int8, c_uint32

import numpy as np


def quantize_input(x):
    """Float image(s) -> int8, as test_inference.py:140-141 / BitNetMCU.py:435-436:
    scale = 127 / max|x| (per image), np.round (half to even), clip to [-128, 127]."""
    x = np.asarray(x, dtype=np.float32).reshape(-1, 256) if np.asarray(x).ndim > 1 else \
        np.asarray(x, dtype=np.float32).reshape(1, 256)
    scale = 127.0 / np.maximum(np.abs(x).max(axis=-1, keepdims=True), 1e-5)
    return np.round(x * scale).clip(-128, 127).astype(np.int8)


def load_inference_dll(path="./Bitnet_inf.dll"):
    """CDLL + the exact prototype the reference binds (test_inference.py:134,146-147)."""
    lib = CDLL(path)
    lib.Inference.argtypes = [POINTER(c_int8)]
    lib.Inference.restype = c_uint32
    return lib


def run_inference_loop(lib, images_int8):
    """Per-image loop of test_inference.py:136-150.  images_int8: [n,256] int8.  Returns uint32[n]."""
    images_int8 = np.ascontiguousarray(images_int8, dtype=np.int8).reshape(-1, 256)
    out = np.empty(images_int8.shape[0], np.uint32)
    for i, row in enumerate(images_int8):
        ptr = (c_int8 * 256)(*row.tolist())
        out[i] = lib.Inference(ptr)
    return out


def cross_check(lib_c, predict_other, images_int8, labels=None, verbose=False):
    """Counters of test_inference.py:123-175: C engine vs another engine (`predict_other`: callable
    [n,256] int8 -> class ids), optionally vs labels."""
    res_c = run_inference_loop(lib_c, images_int8)
    res_o = np.asarray(predict_other(images_int8), dtype=np.uint32)
    mismatch = np.nonzero(res_c != res_o)[0]
    if verbose:
        for i in mismatch:
            print(f"{i:5} Mismatch between inference engines found. Prediction C: {res_c[i]} "
                  f"Prediction other: {res_o[i]}")
    stats = {"counter": int(len(res_c)), "mismatch": int(len(mismatch)), "mismatch_idx": mismatch,
             "result_c": res_c, "result_other": res_o}
    if labels is not None:
        labels = np.asarray(labels)
        stats["correct_c"] = int((res_c == labels).sum())
        stats["correct_other"] = int((res_o == labels).sum())
    return stats
