import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored: on a fresh checkout compile them first (hipcc cross-compiles gfx950 without a GPU;
    the same thing __graft_entry__.build() does).  A failed build is reported by the tests that need the library."""
    need = [os.path.join(ROOT, "mitsubaer_amd", "libmer.so"), os.path.join(ROOT, "mitsubaer_amd", "libmer_check.so"), os.path.join(ROOT, "mitsubaer_amd", "libmer_host.so"),
            os.path.join(ROOT, "oracle", "libmer_oracle.so")]
    if not all(os.path.exists(f) for f in need):
        try:
            import __graft_entry__
            __graft_entry__.build()
        except Exception as e:          # noqa: BLE001
            print("conftest: build failed: %s" % e, file=sys.stderr)


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
