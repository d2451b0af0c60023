import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One libksgpu context on cuda:0 for the whole GPU session (fails loudly without a gfx950 device)."""
    import slepc_amd as ks
    c = ks.Context(0)
    yield c
    c.close()


@pytest.fixture
def debug(ctx):
    """Test hooks of the session's context (ks_ctx_set_debug) for ONE test: every hook touched is back at its default afterwards."""
    touched = []

    def set_(key, value=1):
        touched.append(key)
        ctx.set_debug(key, value)
    yield set_
    for key in touched:
        ctx.set_debug(key, 1 if key == "halo_overlap" else 0)
