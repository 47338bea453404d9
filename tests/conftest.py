import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure both shared libraries exist (CPU cross-compile works without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def orc(built):
    import oracle_bind
    return oracle_bind


@pytest.fixture(scope="session")
def vislam(built):
    import vislam
    return vislam


@pytest.fixture(scope="session")
def ctx(vislam):
    c = vislam.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def canvas(vislam):
    return vislam.synth_canvas(2048, 0xE0C00001)
