import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import util
    return util.load_oracle()


@pytest.fixture(scope="session")
def orc_funcs(orc):
    import util
    return util.Funcs(orc, "orc_")


@pytest.fixture(scope="session")
def bnm():
    """The product library; building it is __graft_entry__.build()'s job."""
    import bitnetmcu_amd
    return bitnetmcu_amd.load()


@pytest.fixture(scope="session")
def gpu_ok(bnm):
    if bnm.bnm_device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (no CPU fallback exists)")
    return True
