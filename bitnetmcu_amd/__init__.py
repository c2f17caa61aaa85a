"""bitnetmcu_amd — MI355X-native batch inference for BitNetMCU models.

Host-side mirror of the reference's Python boundary (test_inference.py:134-150): the native library
does the work, this package loads exporter-written ``BitNetMCU_model.h`` files, moves buffers and
launches.  PyTorch is used only for device memory, streams and torch.distributed.
"""
from ._lib import (BnmError, LayerInfo, load, LIB_PATH, PATH_AUTO, PATH_FUSED_MFMA, PATH_LAYERWISE_ALU,
                   PATH_TERNARY_ALU, PATH_LAYERWISE_MFMA, DIST_U, DIST_M, SEED_DIST_U, SEED_DIST_M, KIND_FC, KIND_CNN)
from .model import Model, Context
from . import harness, synth, dist, evaluate

__all__ = ["BnmError", "LayerInfo", "load", "LIB_PATH", "Model", "Context", "harness", "synth", "dist", "evaluate",
           "PATH_AUTO", "PATH_FUSED_MFMA", "PATH_LAYERWISE_ALU", "PATH_TERNARY_ALU", "PATH_LAYERWISE_MFMA", "DIST_U", "DIST_M",
           "SEED_DIST_U", "SEED_DIST_M", "KIND_FC", "KIND_CNN"]
