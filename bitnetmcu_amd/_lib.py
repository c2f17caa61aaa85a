"""ctypes binding of libbitnetmcu_hip.so (C ABI: include/bitnetmcu_hip.h).

The library is the product; this module only declares prototypes.  It fails loudly when the shared
object is missing — there is no Python or CPU fallback for any compute entry point.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# BNM_LIBRARY: load another build of the same library instead (diagnostic builds, see build.py --diag-timing)
LIB_PATH = os.environ.get("BNM_LIBRARY") or os.path.join(HERE, "libbitnetmcu_hip.so")

BNM_OK = 0
KIND_FC, KIND_CNN = 0, 1
LAYER_FC, LAYER_CONV, LAYER_POOL = 1, 2, 3
PATH_AUTO, PATH_FUSED_MFMA, PATH_LAYERWISE_ALU, PATH_TERNARY_ALU, PATH_LAYERWISE_MFMA = 0, 1, 2, 3, 4
DIST_U, DIST_M = 0, 1
SEED_DIST_U, SEED_DIST_M = 0xB17E7001, 0xB17E7002


class LayerInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("type", "order")] + [("bits_per_weight", C.c_int32)] + [
        (n, C.c_uint32) for n in ("n_input", "n_output", "in_channels", "out_channels", "groups", "kernel_size",
                                  "incoming_x", "outgoing_x", "pool_size", "weight_elem_bytes", "weight_count")]


# every symbol include/bitnetmcu_hip.h declares: name -> (restype, argtypes)
_u8p, _i8p, _u32p, _i32p, _u64p, _vp = (C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.c_void_p)
PROTOTYPES = {
    # (A) reference ABI
    "Inference": (C.c_uint32, [_i8p]),
    "BitMnistInference": (C.c_uint32, [_i8p]),
    "processfclayer": (None, [_i8p, _u32p, C.c_int32, C.c_uint32, C.c_uint32, _i32p]),
    "ReLUNorm": (C.c_uint32, [_i32p, _i8p, C.c_uint32]),
    "processconv33ReLU": (_i32p, [_i32p, _i8p, C.c_uint32, C.c_uint32, _i32p]),
    "processmaxpool22": (_i32p, [_i32p, C.c_uint32, _i32p]),
    # (B) additive ABI
    "bnm_last_error": (C.c_char_p, []),
    "bnm_version": (C.c_char_p, []),
    "bnm_model_from_header_text": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(_vp)]),
    "bnm_model_from_blob": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "bnm_model_blob_size": (C.c_size_t, [_vp]),
    "bnm_model_to_blob": (C.c_int, [_vp, _vp, C.c_size_t]),
    "bnm_model_free": (None, [_vp]),
    "bnm_model_kind": (C.c_uint32, [_vp]),
    "bnm_model_num_layers": (C.c_uint32, [_vp]),
    "bnm_model_num_classes": (C.c_uint32, [_vp]),
    "bnm_model_input_bytes": (C.c_uint32, [_vp]),
    "bnm_model_layer": (C.c_int, [_vp, C.c_uint32, C.POINTER(LayerInfo)]),
    "bnm_model_layer_weights": (_vp, [_vp, C.c_uint32]),
    "bnm_ctx_create": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "bnm_ctx_destroy": (None, [_vp]),
    "bnm_ctx_device": (C.c_int, [_vp]),
    "bnm_ctx_set_path": (C.c_int, [_vp, C.c_int]),
    "bnm_ctx_get_path": (C.c_int, [_vp]),
    "bnm_ctx_get_variant": (C.c_int, [_vp]),
    "bnm_ctx_set_tuning": (C.c_int, [_vp, C.c_int, C.c_int]),
    "bnm_ctx_set_cnn_variant": (C.c_int, [_vp, C.c_int]),
    "bnm_ctx_get_cnn_variant": (C.c_int, [_vp]),
    "bnm_ctx_set_ternary_variant": (C.c_int, [_vp, C.c_int]),
    "bnm_ctx_set_work_batch": (C.c_int, [_vp, C.c_int]),
    "bnm_ctx_set_host_tuning": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "bnm_infer_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "bnm_ctx_release_stream": (C.c_int, [_vp, _vp]),
    "bnm_infer_host": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "bnm_infer_host_activations": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint32]),
    "bnm_fc_layer_device": (C.c_int, [_vp, C.c_uint32, _vp, C.c_int32, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp]),
    "bnm_relunorm_device": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint64, _vp]),
    "bnm_unpack_layer_host": (C.c_int, [_vp, C.c_int32, C.c_uint32, C.c_uint32, _vp, _vp, C.c_uint32]),
    "bnm_quantize_input_device": (C.c_int, [_vp, C.c_uint64, _vp, _vp]),
    "bnm_infer_float_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "bnm_ctx_cnn_tail_fused": (C.c_int, [_vp]),
    "bnm_ctx_cnn_planes": (C.c_int, [_vp]),
    "bnm_ctx_cnn_pipelined": (C.c_int, [_vp]),
    "bnm_ctx_last_kernel": (C.c_char_p, [_vp]),
    "bnm_ctx_set_float_mode": (C.c_int, [_vp, C.c_int, C.c_int]),
    "bnm_ctx_float_fused": (C.c_int, [_vp]),
    "bnm_ctx_float_nonfinite": (C.c_int, [_vp, _u64p]),
    "bnm_ctx_set_persistent": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "bnm_ctx_persistent_last_call": (C.c_int, [_vp, _u32p, _u32p]),
    "bnm_quantize_input_counted_device": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp]),
    "bnm_qat_workspace_bytes": (C.c_uint64, [C.c_uint32, C.c_uint32]),
    "bnm_qat_bitconv2d_forward_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_uint32,
                                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_int, C.c_int, _vp, _vp,
                                                   C.c_uint64, _vp]),
    "bnm_qat_bitlinear_forward_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint32, C.c_int,
                                                   C.c_int, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "bnm_qat_model_workspace_bytes": (C.c_uint64, [C.c_uint32, _vp]),
    "bnm_qat_model_supported": (C.c_int, [C.c_uint32, _vp, _vp, C.c_int]),
    "bnm_qat_model_forward_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp,
                                               C.c_uint64, _vp]),
    "bnm_qat_cnn_front_supported": (C.c_int, [C.c_uint32, _vp, _vp]),
    "bnm_qat_cnn_front_workspace_bytes": (C.c_uint64, [C.c_uint32]),
    "bnm_qat_cnn_front_forward_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "bnm_qat_cnn_front_forward_train_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "bnm_synth_fill_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _vp]),
    "bnm_class_digest_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, C.c_uint32, _vp]),
    "bnm_stream_read_device": (C.c_int, [_vp, C.c_uint64, _vp, _vp]),
    "bnm_stream_rw_device": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint32, C.c_uint32, _vp]),
    "bnm_run_synth_multi_gpu": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_int, C.c_uint64, _vp, C.c_uint32, C.POINTER(C.c_double)]),
    "bnm_multi_gpu_transport": (C.c_char_p, []),
    "bnm_bind_default_model": (C.c_int, [_vp]),
    "bnm_device_count": (C.c_int, []),
    "bnm_device_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "bnm_device_free": (C.c_int, [_vp]),
    "bnm_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "bnm_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "bnm_device_synchronize": (C.c_int, []),
}


# diagnostic library only (bitnetmcu_amd/build.py --diag, csrc/bnm_diag.h): bound when present, never required
DIAG_PROTOTYPES = {
    "bnm_diag_stream_device": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_int, _vp, _vp]),
    "bnm_diag_set_src_wrap": (C.c_int, [_vp, C.c_uint64]),
    "bnm_diag_cnn_set_record": (C.c_int, [_vp]),
}


class BnmError(RuntimeError):
    pass


def bind(lib, strict=True):
    """Attach prototypes to a loaded CDLL.  strict: every declared symbol must be exported."""
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and strict:
        raise BnmError(f"{lib._name} does not export: {missing}")
    for name, (res, args) in DIAG_PROTOTYPES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib


_lib = None


def load(path=None):
    """Load the native library (cached).  Raises if it has not been built — no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # PyTorch-ROCm wheels bundle their own HIP runtime.  If this library pulled in the system libamdhip64 first, a
    # later `import torch` would find two runtimes in the process and report "No HIP GPUs are available"; importing
    # torch first makes both share torch's copy.  (Pure C hosts never see this: they have one runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise BnmError(f"{p} not found: build it with `python bitnetmcu_amd/build.py` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = bind(C.CDLL(p))
    if path is None:
        _lib = lib
    return lib


def check(lib, rc, what=""):
    if rc != BNM_OK:
        raise BnmError(f"{what} failed ({rc}): {lib.bnm_last_error().decode(errors='replace')}")
