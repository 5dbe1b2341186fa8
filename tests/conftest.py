import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_api
    return oracle_api.load()


@pytest.fixture(scope="session")
def gpu_ctx():
    """One vpl_ctx on device 0 for the GPU parity tests (calls go through the C ABI)."""
    import vplines_slam_amd as v
    ctx = v.Context(device=0, max_windows=64)
    yield ctx
    ctx.close()
