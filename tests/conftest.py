import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def ctx():
    """HIP context on cuda:0 -- GPU tests only.  Fails loudly when libmer.so or the GPU is missing."""
    from mitsubaer_amd import capi
    c = capi.Context(0)
    yield c
    c.close()
