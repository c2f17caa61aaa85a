"""The C-ABI shared library loads without a GPU and exports every symbol include/bitnetmcu_hip.h declares
(no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import bitnetmcu_amd
from bitnetmcu_amd import _lib
from util import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "bitnetmcu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"BNM_API\s+[\w\s\*]+?\b(\w+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.PROTOTYPES)


def test_library_exports_all_symbols(bnm):
    for name in declared_symbols():
        assert hasattr(bnm, name), name
    assert b"gfx950" in bnm.bnm_version()


def test_reference_symbol_set_of_the_dropin_dll():
    """A model-bound Bitnet_inf.dll exports what the reference DLL exports."""
    dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", "tern_96", "Bitnet_inf.dll")
    if not os.path.isfile(dll):
        pytest.skip("run __graft_entry__.build() first")
    out = subprocess.run(["nm", "-D", "--defined-only", dll], capture_output=True, text=True).stdout
    syms = {l.split()[-1] for l in out.splitlines() if l.strip()}
    for s in ("Inference", "BitMnistInference", "processfclayer", "ReLUNorm", "processconv33ReLU", "processmaxpool22",
              "L1_weights", "L2_weights", "L3_weights", "L4_weights"):
        assert s in syms, s
    lib = C.CDLL(dll)   # loads on a machine without a GPU; calling Inference there aborts by design
    assert lib.Inference is not None


def test_no_gpu_means_error_not_fallback(bnm):
    """On a box without a HIP device the additive API reports BNM_EHIP; it never computes on the CPU."""
    if bnm.bnm_device_count() > 0:
        pytest.skip("a GPU is present")
    import util
    m = util.load_golden_model("mcu_1k")
    with pytest.raises(bitnetmcu_amd.BnmError) as e:
        bitnetmcu_amd.Context(m)
    assert "(-4)" in str(e.value)
