"""The CHECKER: the oracle port (oracle/bitnet_oracle.c -> libbnm_oracle.so) bound to a parsed model.

Test infrastructure.  Only tests/, __graft_entry__.smoke() and bench.py (its verification step outside the timed region and
its cpu_baseline leg) may import this module; nothing under bitnetmcu_amd/ does - the product path has no CPU fallback.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
ORACLE_SO = os.path.join(HERE, "libbnm_oracle.so")


def load_oracle():
    if not os.path.isfile(ORACLE_SO):
        import subprocess
        subprocess.check_call([sys.executable, os.path.join(HERE, "build_oracle.py"), "--port"])
    lib = C.CDLL(ORACLE_SO)
    lib.orc_weight_at.restype = C.c_int32
    lib.orc_weight_at.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.orc_synth.restype = None
    lib.orc_synth.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.orc_class_digest.restype = C.c_uint64
    lib.orc_class_digest.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32]
    lib.orc_model_batch.restype = None
    lib.orc_model_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    return lib


class OrcFcLayer(C.Structure):
    _fields_ = [("bits_per_weight", C.c_int32), ("n_input", C.c_uint32), ("n_output", C.c_uint32), ("weights", C.c_void_p)]


class OrcCnnFront(C.Structure):
    _fields_ = [("channels", C.c_uint32), ("w_conv1", C.c_void_p), ("w_conv2", C.c_void_p), ("w_conv3", C.c_void_p),
                ("n_shift", C.c_uint32)]


class OracleModel:
    """A parsed Model bound to the oracle port's batch driver (orc_model_batch)."""

    def __init__(self, model, orc=None):
        from bitnetmcu_amd import _lib as L
        self.orc = orc or load_oracle()
        self.model = model
        layers = model.layers()
        self._keep = []
        fcs = [(i, l) for i, l in enumerate(layers) if l.type == L.LAYER_FC]
        self.n_layers = len(fcs)
        self.arr = (OrcFcLayer * len(fcs))()
        for k, (i, li) in enumerate(fcs):
            w = model.layer_weights(i)
            self._keep.append(w)
            self.arr[k] = OrcFcLayer(li.bits_per_weight, li.n_input, li.n_output, w.ctypes.data)
        self.front = None
        if model.kind == L.KIND_CNN:
            ws = [model.layer_weights(i) for i in (0, 1, 3)]
            self._keep += ws
            self.front = OrcCnnFront(layers[0].out_channels, ws[0].ctypes.data, ws[1].ctypes.data, ws[2].ctypes.data, 4)
        self.n_classes = model.num_classes

    def infer(self, images, logits=False):
        x = np.ascontiguousarray(images, dtype=np.int8).reshape(-1, 256)
        n = len(x)
        cls = np.zeros(n, np.uint32)
        lg = np.zeros((n, self.n_classes), np.int32)
        self.orc.orc_model_batch(x.ctypes.data, n, C.byref(self.front) if self.front else None, self.arr, self.n_layers,
                                 cls.ctypes.data, lg.ctypes.data)
        return (cls, lg) if logits else cls



def sample_indices(n):
    """head + tail + strided sample of [0, n)"""
    idx = np.concatenate([np.arange(0, 4096), np.arange(n - 2048, n), np.linspace(0, n - 1, 2048).astype(np.int64)])
    return np.unique(idx[(idx >= 0) & (idx < n)])


def quantize_input(x):
    """The reference's Python-side input quantisation, its arithmetic verbatim (test_inference.py:140-141; the same two lines at
    BitNetMCU.py:435-436) on float32 images [n, 256]: per-image scale 127 / max(max|x|, 1e-5), np.round (half to even), clip."""
    input_data = np.asarray(x, dtype=np.float32).reshape(-1, 256)
    scale = 127.0 / np.maximum(np.abs(input_data).max(axis=-1, keepdims=True), 1e-5)
    scaled_data = np.round(input_data * scale).clip(-128, 127)
    return scaled_data.astype(np.int8)


def verify_float_sample(torch, model, xf, cls, logits, n):
    """Float images resident on the device: quantise_input + the oracle on sample_indices(n) against the device's class ids (and logits)."""
    om = OracleModel(model)
    idx = sample_indices(n)
    ti = torch.from_numpy(idx).to(xf.device)
    want, want_lg = om.infer(quantize_input(xf[ti].cpu().numpy()), logits=True)
    ok = bool(np.array_equal(want, cls[ti].cpu().numpy().astype(np.uint32)))
    if logits is not None:
        ok = ok and bool(np.array_equal(want_lg, logits[ti].cpu().numpy()))
    return ok


def verify_sample(torch, model, images, cls, logits, n):
    """The device results of a resident image set against the oracle on sample_indices(n): class ids, and logits when written."""
    om = OracleModel(model)
    idx = sample_indices(n)
    ti = torch.from_numpy(idx).to(images.device)
    want, want_lg = om.infer(images[ti].cpu().numpy(), logits=True)
    ok = bool(np.array_equal(want, cls[ti].cpu().numpy().astype(np.uint32)))
    if logits is not None:
        ok = ok and bool(np.array_equal(want_lg, logits[ti].cpu().numpy()))
    return ok
