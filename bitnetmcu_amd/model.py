"""Model (host, parsed header) and Context (model resident on one GPU)."""
import ctypes as C
import os

import numpy as np

from . import _lib as L

# Model zoo: the weight data of every header in the reference tree (and of the generated ternary / 12 KB-family headers under
# tests/golden/headers) as BNMBLOB1 blobs, written by tests/golden/make_golden.py after a word-for-word check against the
# compiled reference's own Lk_weights symbols.  The reference's headers themselves do not travel to the GPU box.
ZOO_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zoo")


class Model:
    """A BitNetMCU model parsed from exporter-written header text or from a BNMBLOB1 blob.

    Parsing is done by the native library (bitnetmcu_amd/csrc/bnm_model.cpp) so that C and Python
    hosts share one loader; no GPU is needed for this class.
    """

    def __init__(self, handle, lib):
        self._h, self._lib = handle, lib

    @classmethod
    def from_header_text(cls, text, lib=None):
        lib = lib or L.load()
        if isinstance(text, str):
            text = text.encode()
        h = C.c_void_p()
        L.check(lib, lib.bnm_model_from_header_text(text, len(text), C.byref(h)), "bnm_model_from_header_text")
        return cls(h, lib)

    @classmethod
    def from_header(cls, path, lib=None):
        with open(path, "rb") as f:
            return cls.from_header_text(f.read(), lib)

    @classmethod
    def from_blob(cls, blob, lib=None):
        lib = lib or L.load()
        blob = bytes(blob)
        h = C.c_void_p()
        L.check(lib, lib.bnm_model_from_blob(blob, len(blob), C.byref(h)), "bnm_model_from_blob")
        return cls(h, lib)

    @classmethod
    def from_zoo(cls, name, lib=None):
        """One of the committed zoo models (bitnetmcu_amd/zoo/<name>.bnm); zoo_names() lists them."""
        with open(os.path.join(ZOO_DIR, name + ".bnm"), "rb") as f:
            return cls.from_blob(f.read(), lib)

    @staticmethod
    def zoo_names():
        return sorted(f[:-4] for f in os.listdir(ZOO_DIR) if f.endswith(".bnm"))

    def to_header_text(self, dialect="exporter"):
        """BitNetMCU_model.h text of this model in the exporter's dialect (headerwriter.write_header)."""
        from .headerwriter import write_header
        return write_header(self, dialect)

    def to_blob(self):
        n = self._lib.bnm_model_blob_size(self._h)
        buf = C.create_string_buffer(n)
        L.check(self._lib, self._lib.bnm_model_to_blob(self._h, buf, n), "bnm_model_to_blob")
        return buf.raw

    def __del__(self):
        try:
            if self._h:
                self._lib.bnm_model_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def kind(self):
        return self._lib.bnm_model_kind(self._h)

    @property
    def num_classes(self):
        return self._lib.bnm_model_num_classes(self._h)

    @property
    def num_layers(self):
        return self._lib.bnm_model_num_layers(self._h)

    def layer(self, i):
        info = L.LayerInfo()
        L.check(self._lib, self._lib.bnm_model_layer(self._h, i, C.byref(info)), "bnm_model_layer")
        return info

    def layer_weights(self, i):
        """The layer's array exactly as the C compiler lays out ``Lk_weights[]`` (numpy copy)."""
        info = self.layer(i)
        if info.weight_count == 0:
            return np.zeros(0, np.uint32)
        dt = {4: np.uint32, 2: np.uint16, 1: np.int8}[info.weight_elem_bytes]
        p = self._lib.bnm_model_layer_weights(self._h, i)
        raw = C.string_at(p, info.weight_count * info.weight_elem_bytes)
        return np.frombuffer(raw, dtype=dt).copy()

    def layers(self):
        return [self.layer(i) for i in range(self.num_layers)]

    def fc_layers(self):
        return [(i, li) for i, li in enumerate(self.layers()) if li.type == L.LAYER_FC]


class Context:
    """A model uploaded to and unpacked on one GPU (bnm_ctx)."""

    def __init__(self, model, device=-1):
        self._lib = model._lib
        self.model = model
        h = C.c_void_p()
        L.check(self._lib, self._lib.bnm_ctx_create(model._h, device, C.byref(h)), "bnm_ctx_create")
        self._h = h
        self.cnn_variant = self._lib.bnm_ctx_get_cnn_variant(self._h)      # 3 lane = image kernel, 1 channel kernel (see set_cnn_variant)
        self.ternary_variant = 2
        # diagnostic library only (BNM_LIBRARY=.../libbitnetmcu_hip_diag.so, build.py --diag): cache-resident source for
        # compute-side timing.  The product library does not export the symbol and ignores the variable.
        import os
        wrap = os.environ.get("BNM_DIAG_SRC_WRAP")
        if wrap and hasattr(self._lib, "bnm_diag_set_src_wrap"):
            L.check(self._lib, self._lib.bnm_diag_set_src_wrap(self._h, int(wrap)), "bnm_diag_set_src_wrap")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bnm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def path(self):
        return self._lib.bnm_ctx_get_path(self._h)

    @property
    def variant(self):
        """fused-kernel variant in use (-1 when the resolved path is not the fused kernel)"""
        return self._lib.bnm_ctx_get_variant(self._h)

    def set_path(self, path):
        L.check(self._lib, self._lib.bnm_ctx_set_path(self._h, path), "bnm_ctx_set_path")

    def set_tuning(self, variant=-1, grid_blocks=0):
        L.check(self._lib, self._lib.bnm_ctx_set_tuning(self._h, variant, grid_blocks), "bnm_ctx_set_tuning")

    def set_work_batch(self, tiles):
        """generic fused kernel: tiles per take from the device-wide work counter (0 = default)"""
        L.check(self._lib, self._lib.bnm_ctx_set_work_batch(self._h, tiles), "bnm_ctx_set_work_batch")

    def set_ternary_variant(self, variant):
        """Ternary ALU kernel: 2 streamed weights + two images per lane (default), 1 one image per lane, 0 round 1's."""
        L.check(self._lib, self._lib.bnm_ctx_set_ternary_variant(self._h, variant), "bnm_ctx_set_ternary_variant")
        self.ternary_variant = variant

    def set_cnn_variant(self, variant):
        """1: conv1 on the matrix cores, a lane = a channel, dynamic image batches; 2: fixed share per wave; 100 + g: batches of g
        images; 0: the all-VALU front end of round 1; 3: the lane = image kernel (all three convolutions on the matrix cores;
        300 + g: g tiles per take; the FC tail runs in the same wave where it fits - 4 / 400 + g: the tail as its own launch; 5: conv3's
        third plane kept; 6: the four-waves-per-SIMD form instead of the pipelined one - A/B measurements)"""
        L.check(self._lib, self._lib.bnm_ctx_set_cnn_variant(self._h, variant), "bnm_ctx_set_cnn_variant")
        self.cnn_variant = 0 if variant == 0 else 3 if (variant in (3, 4, 5, 6) or variant > 300) else 1

    @property
    def cnn_planes(self):
        """conv3 operand planes of the lane = image CNN kernels: 2 when the weights bound the pooled conv2 outputs below 2^16, else 3."""
        return self._lib.bnm_ctx_cnn_planes(self._h)

    @property
    def cnn_pipelined(self):
        """The one-kernel CNN form runs as cnn_li_fused_pipe_kernel (conv1 sums below 2^16: every CNN of the reference's zoo)."""
        return self._lib.bnm_ctx_cnn_pipelined(self._h) == 1

    @property
    def cnn_tail_fused(self):
        """True when calls that take the lane = image front end run ONE kernel (front end + FC tail in the same wave)."""
        return self._lib.bnm_ctx_cnn_tail_fused(self._h) == 1

    def release_stream(self, stream):
        """Drop the scratch buffers and the counter block the context keeps for `stream` (a torch.cuda.Stream); synchronises it."""
        L.check(self._lib, self._lib.bnm_ctx_release_stream(self._h, stream.cuda_stream), "bnm_ctx_release_stream")

    def set_persistent(self, on=True, idle_us=0):
        """One-image host calls (infer_host with n = 1: what Inference() runs) through a resident single-wave kernel and a page-locked
        mailbox instead of a launch per call (bnm_ctx_set_persistent); the kernel leaves by itself after idle_us without a call."""
        L.check(self._lib, self._lib.bnm_ctx_set_persistent(self._h, 1 if on else 0, idle_us), "bnm_ctx_set_persistent")

    def set_host_tuning(self, mode=0, copy_threads=0, spin=True):
        L.check(self._lib, self._lib.bnm_ctx_set_host_tuning(self._h, mode, copy_threads, 1 if spin else 0), "bnm_ctx_set_host_tuning")

    # ---- host-pointer API (numpy) -------------------------------------------------------------
    def infer(self, images, logits=False):
        """images: int8 array [n,256] (or [n,16,16]).  Returns class ids (uint32[n]) and, if asked,
        the last layer's int32 outputs [n,num_classes]."""
        x = np.ascontiguousarray(images, dtype=np.int8).reshape(-1, 256)
        n = x.shape[0]
        cls = np.empty(n, np.uint32)
        lg = np.empty((n, self.model.num_classes), np.int32) if logits else None
        L.check(self._lib, self._lib.bnm_infer_host(self._h, x.ctypes.data, n, cls.ctypes.data,
                                                    lg.ctypes.data if logits else None), "bnm_infer_host")
        return (cls, lg) if logits else cls

    def activations(self, images):
        """int8 activations after every ReLUNorm, concatenated per image (parity tap)."""
        x = np.ascontiguousarray(images, dtype=np.int8).reshape(-1, 256)
        n = x.shape[0]
        width = (self.model.layer(0).out_channels * 4 if self.model.kind == L.KIND_CNN else 0) + \
            sum(li.n_output for _, li in self.model.fc_layers())
        out = np.zeros((n, width), np.int8)
        L.check(self._lib, self._lib.bnm_infer_host_activations(self._h, x.ctypes.data, n, out.ctypes.data, width),
                "bnm_infer_host_activations")
        return out

    # ---- device-pointer API (torch tensors on the context's GPU) --------------------------------
    def infer_device(self, images, cls, logits=None, stream=None):
        """Asynchronous launch on torch's current stream (or `stream`).  images: torch.int8 [n,256]
        cuda tensor; cls: torch.int32 [n]; logits: torch.int32 [n,num_classes] or None."""
        import torch
        n = images.shape[0]
        assert images.is_cuda and images.is_contiguous() and images.dtype == torch.int8
        assert cls.is_cuda and cls.is_contiguous() and cls.element_size() == 4 and cls.numel() >= n
        s = stream if stream is not None else torch.cuda.current_stream(images.device)
        L.check(self._lib, self._lib.bnm_infer_device(self._h, images.data_ptr(), n, cls.data_ptr(),
                                                      logits.data_ptr() if logits is not None else None,
                                                      s.cuda_stream), "bnm_infer_device")
        return cls

    def infer_float_device(self, x, cls, logits=None, stream=None):
        """float32 cuda tensor [n,256] -> class ids (torch.int32 [n]) and optionally the int32 logits: the reference's per-image
        flow (quantise as test_inference.py:140-141, then Inference()) for a whole batch, asynchronous on the stream."""
        import torch
        n = x.numel() // 256
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
        assert cls.is_cuda and cls.is_contiguous() and cls.element_size() == 4 and cls.numel() >= n
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        L.check(self._lib, self._lib.bnm_infer_float_device(self._h, x.data_ptr(), n, cls.data_ptr(),
                                                            logits.data_ptr() if logits is not None else None, s.cuda_stream),
                "bnm_infer_float_device")
        return cls

    @property
    def last_kernel(self):
        """Names of the kernels the context's last inference call launched, joined by '+' (bnm_ctx_last_kernel)."""
        return self._lib.bnm_ctx_last_kernel(self._h).decode()

    def set_float_mode(self, mode=0, groups=0):
        """How infer_float_device runs: 0 the fused float-input kernel where it exists, 1 fused or an error, 2 always
        quantise + infer (two kernels); groups: 8-image groups in flight per wave of the fused kernel (0 = default; 1, 2, 4: what the tile class has)."""
        L.check(self._lib, self._lib.bnm_ctx_set_float_mode(self._h, mode, groups), "bnm_ctx_set_float_mode")

    @property
    def float_nonfinite(self):
        """Images, over all float calls of this context so far, that held a NaN or an infinity: each was classified as the all-zero
        image (what the reference's numpy quantisation makes of it on x86).  Synchronises the device (bnm_ctx_float_nonfinite)."""
        import ctypes as C
        v = C.c_uint64(0)
        L.check(self._lib, self._lib.bnm_ctx_float_nonfinite(self._h, C.byref(v)), "bnm_ctx_float_nonfinite")
        return int(v.value)

    @property
    def float_fused(self):
        """True when float calls of this context run the one-kernel path now."""
        return self._lib.bnm_ctx_float_fused(self._h) == 1

    def quantize_device(self, x, out=None, stream=None):
        """float32 cuda tensor [n,256] -> int8 [n,256] on the GPU with the reference's input quantisation
        (test_inference.py:140-141), bit-identical to harness.quantize_input."""
        import torch
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
        n = x.numel() // 256
        if out is None:
            out = torch.empty((n, 256), dtype=torch.int8, device=x.device)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        L.check(self._lib, self._lib.bnm_quantize_input_device(x.data_ptr(), n, out.data_ptr(), s.cuda_stream),
                "bnm_quantize_input_device")
        return out
