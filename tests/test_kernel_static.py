"""Static checks of the built gfx950 code objects (no GPU): no kernel spills, and the streamed ternary kernels' compiler-invisible
scalar loads are never touched while in flight (profiles/check_inflight_sgprs.py)."""
import os
import subprocess
import sys

import pytest

import util

BUILD = os.path.join(util.REPO, "bitnetmcu_amd", "_build")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(BUILD, "bnm_ternary.o")), reason="objects not built")


def test_no_kernel_spills_or_scratch():
    r = subprocess.run([sys.executable, os.path.join(util.REPO, "profiles", "kernel_resources.py"), "--spills-only"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 with VGPR spills or scratch" in r.stdout


def test_ternary_chunk_loads_untouched_in_flight():
    r = subprocess.run([sys.executable, os.path.join(util.REPO, "profiles", "check_inflight_sgprs.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 instructions touch registers of a load in flight" in r.stdout
    assert " 0 instructions touch a result register in flight" in r.stdout and " 0 scalar work-counter takes" not in r.stdout
