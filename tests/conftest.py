import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One GPU context for the whole session.  Fails loudly (no skip, no fallback) when the HIP
    library or the GPU is missing: -m gpu tests are only meaningful on the GPU box."""
    from freddie_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
