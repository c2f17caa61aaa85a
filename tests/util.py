"""Shared helpers for the test-suite: library loaders and a Python statement of the reference's
model schedule that can drive ANY implementation exposing the reference's four C signatures
(the compiled reference in oracle/_ref, the oracle port, or the product's own reference-ABI symbols).
"""
import ctypes as C
import os
import re
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
REF_DIR = "/root/reference"
ORACLE_SO = os.path.join(REPO, "oracle", "libbnm_oracle.so")
REF_OUT = os.path.join(REPO, "oracle", "_ref")

MODEL_NAMES = ["fc_4bitsym_64", "cnn_64", "mcu_12k", "mcu_12k_fp130", "mcu_1k", "mcu_cnn_16", "mcu_cnn_16small",
               "mcu_cnn_32", "mcu_cnn_48", "mcu_cnn_64", "mcu_cnn_letters", "tern_96", "tern_96_sparse",
               "doc12k_binary", "doc12k_ternary", "doc12k_2bit", "doc12k_8bit"]

_i8p, _u32p, _i32p = C.POINTER(C.c_int8), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)


def have_reference():
    return os.path.isfile(os.path.join(REF_DIR, "BitNetMCU_inference.c"))


def ref_dll_path(name, o3=False):
    return os.path.join(REF_OUT, name, "Bitnet_inf_O3.dll" if o3 else "Bitnet_inf.dll")


def have_ref_dll(name):
    return os.path.isfile(ref_dll_path(name))


class Funcs:
    """The reference's four kernels behind one interface (numpy in / numpy out)."""

    def __init__(self, lib, prefix=""):
        self.lib = lib
        g = lambda n: getattr(lib, prefix + n)
        self.fc, self.rn = g("processfclayer"), g("ReLUNorm")
        self.fc.restype = None
        self.fc.argtypes = [_i8p, C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, _i32p]
        self.rn.restype = C.c_uint32
        self.rn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        try:    # FC builds of the reference compile conv/pool out (BitNetMCU_inference.c:210 #ifndef MODEL_FCMNIST)
            self.conv, self.pool = g("processconv33ReLU"), g("processmaxpool22")
        except AttributeError:
            self.conv = self.pool = None
            return
        self.fc.argtypes = [_i8p, C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, _i32p]
        self.rn.restype = C.c_uint32
        self.rn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        self.conv.restype = C.c_void_p
        self.conv.argtypes = [C.c_void_p, _i8p, C.c_uint32, C.c_uint32, C.c_void_p]
        self.pool.restype = C.c_void_p
        self.pool.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]

    def processfclayer(self, act, weights, bpw, n_in, n_out):
        act = np.ascontiguousarray(act, dtype=np.int8)
        weights = np.ascontiguousarray(weights)
        out = np.zeros(n_out, np.int32)
        self.fc(act.ctypes.data_as(_i8p), weights.ctypes.data, bpw, n_in, n_out, out.ctypes.data_as(_i32p))
        return out

    def relunorm(self, x):
        x = np.ascontiguousarray(x, dtype=np.int32).copy()
        out = np.zeros(len(x), np.int8)
        pos = self.rn(x.ctypes.data, out.ctypes.data, len(x))
        return out, int(pos)

    def relunorm_inplace(self, x):
        """int32 -> int8 over the SAME buffer (BitNetMCU_MNIST_dll.c:80)."""
        buf = np.ascontiguousarray(x, dtype=np.int32).copy()
        pos = self.rn(buf.ctypes.data, buf.ctypes.data, len(buf))
        return buf.view(np.int8)[:len(buf)].copy(), int(pos)

    def conv33(self, plane, w, xy, shift, inplace=True):
        buf = np.ascontiguousarray(plane, dtype=np.int32).copy()
        w = np.ascontiguousarray(w, dtype=np.int8)
        o = xy - 2
        dst = buf if inplace else np.zeros(o * o, np.int32)
        end = self.conv(buf.ctypes.data, w.ctypes.data_as(_i8p), xy, shift, dst.ctypes.data)
        assert end == dst.ctypes.data + 4 * o * o
        return dst[:o * o].copy()

    def maxpool22(self, plane, xy, inplace=True):
        buf = np.ascontiguousarray(plane, dtype=np.int32).copy()
        o = xy // 2
        dst = buf if inplace else np.zeros(o * o, np.int32)
        end = self.pool(buf.ctypes.data, xy, dst.ctypes.data)
        assert end == dst.ctypes.data + 4 * o * o
        return dst[:o * o].copy()


def run_schedule(f, model, image):
    """The reference's BitMnistInference schedule (BitNetMCU_MNIST_dll.c:48-121) in terms of `f`.
    Returns (class id, logits int32[n_classes], acts int8 concatenated after every ReLUNorm)."""
    from bitnetmcu_amd import _lib as L
    layers = model.layers()
    acts = []
    if model.kind == L.KIND_CNN:
        C_ = layers[0].out_channels
        w1, w2, w3 = model.layer_weights(0), model.layer_weights(1), model.layer_weights(3)
        feat = []
        for c in range(C_):
            p = image.astype(np.int32)
            p = f.conv33(p, w1[9 * c:9 * c + 9], 16, 4)
            p = f.conv33(p, w2[9 * c:9 * c + 9], 14, 4)
            p = f.maxpool22(p, 12)
            p = f.conv33(p, w3[9 * c:9 * c + 9], 6, 4)
            feat.append(f.maxpool22(p, 4))
        act, _ = f.relunorm_inplace(np.concatenate(feat))
        acts.append(act)
        fcs = [(i, l) for i, l in enumerate(layers) if l.type == L.LAYER_FC]
    else:
        act = image.astype(np.int8)
        fcs = list(enumerate(layers))
    cls, logits = 255, None
    for i, li in fcs:
        # the reference passes its fixed-size layer_in buffer; ternary layers declare a padded n_input
        buf = np.zeros(max(li.n_input, len(act)) + 16, np.int8)
        buf[:len(act)] = act
        logits = f.processfclayer(buf, model.layer_weights(i), li.bits_per_weight, li.n_input, li.n_output)
        act, cls = f.relunorm(logits)
        acts.append(act)
    return cls, logits, np.concatenate(acts)


sys.path.insert(0, os.path.join(REPO, "oracle"))
from checker import load_oracle, OrcFcLayer, OrcCnnFront, OracleModel  # noqa: E402,F401  (the checker lives under oracle/)


def load_golden_model(name):
    from bitnetmcu_amd import Model
    return Model.from_zoo(name)


def parse_c_int8_arrays(text):
    """`int8_t input_data_k[256] = {...}; uint8_t label_k = v;` blocks (BitNetMCU_MNIST_test_data.h,
    mcu/BitNetMCUdemo.c:23-28; commented-out blocks are skipped).  Returns (images int8 [k,256], labels)."""
    text = re.sub(r"//[^\n]*", "", text)
    imgs, labels = [], []
    for m in re.finditer(r"int8_t\s+input_data_(\d+)\s*\[256\]\s*=\s*\{([^}]*)\}", text):
        vals = [int(float(t)) if "." in t else int(t, 0)
                for t in (s.strip() for s in m.group(2).split(",")) if t]
        assert len(vals) == 256
        imgs.append(np.array(vals, dtype=np.int64).astype(np.uint8).view(np.int8) if max(vals) > 127
                    else np.array(vals, dtype=np.int8))
        lm = re.search(r"label_%s\s*=\s*(\d+)" % m.group(1), text)
        labels.append(int(lm.group(1)))
    return np.stack(imgs), np.array(labels, dtype=np.uint32)


def test_data_header(images, labels):
    """`int8_t input_data_<k>[256]` / `uint8_t label_<k>` blocks in the layout of the reference's BitNetMCU_MNIST_test_data.h."""
    out = ["#include <stdint.h>"]
    for k, (img, lab) in enumerate(zip(images, labels)):
        out.append(f"int8_t input_data_{k}[256] = {{" + ", ".join(str(int(v)) for v in img) + "};")
        out.append(f"uint8_t label_{k} = {int(lab)};")
    return "\n".join(out) + "\n"


test_data_header.__test__ = False


def compile_c_host(source, include_dir):
    """gcc a C host under examples/ against include/ + `include_dir` and link it with the GPU library; returns the executable."""
    import subprocess
    exe = os.path.join(str(include_dir), os.path.splitext(source)[0])
    libdir = os.path.join(REPO, "bitnetmcu_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", str(include_dir), "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "examples", source), "-L", libdir, "-lbitnetmcu_hip", f"-Wl,-rpath,{libdir}",
                           "-o", exe])
    return exe


def qat_model_golden():
    """The whole-model QAT fixtures (tests/golden/make_qat_model_golden.py) as one mapping: qat_fc_model.npz + the six-tile class's
    qat_fc_model_wide.npz."""
    import numpy as np
    out = {}
    for name in ("qat_fc_model.npz", "qat_fc_model_wide.npz"):
        with np.load(os.path.join(GOLDEN, name)) as f:
            out.update({k: f[k] for k in f.files})
    return out
