import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """build the oracle (test infrastructure) and the product library once per session"""
    import __graft_entry__ as g
    g.build(quiet=True)


@pytest.fixture
def dev_switches():
    """the LMGPU_* development switches (A/B launch forms) are read by liblmgpu_test.so only: tests that set them run on that build"""
    from gtsam_personal_amd import _lib
    _lib.use_test_library(True)
    yield
    _lib.use_test_library(False)
