import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gr():
    """gr_ctx on cuda:0 through the C ABI. No fallback: the HIP library must be present."""
    from granite_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
