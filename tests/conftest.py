import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, in tree, before any test
    imports them.  A tree that went through __graft_entry__.build() already has them and nothing happens here."""
    import glob
    have = (os.path.exists(os.path.join(ROOT, "canvas_amd", "libcanvas_hip.so"))
            and glob.glob(os.path.join(ROOT, "fluggo", "media", "process*.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle_fma.so")))
    if not have:
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure only)."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="module")
def cvs():
    """The library, initialised on device 0 (GPU tests; modules with their own `cvs` fixture shadow this one)."""
    from canvas_amd import _lib
    lib = _lib.load()
    assert lib.cvs_init(0) == 0, _lib.last_error()
    lib.init_half()
    return lib
