import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "separate_only: a GPU parity test that holds for the default arithmetic flavour only "
                                       "(it pins a kernel, a frozen answer or a model that exists for the gcc build of the reference)")


# The reference has two builds (SConstruct:46-48,75-83): gcc -std=c99 rounds every multiply and add on its own, clang fuses
# a * b + c inside an expression.  The library has a flavour for each (cvs_set_arithmetic) and the oracle a build for each
# (liboracle.so / liboracle_fma.so): every GPU parity test of these modules runs twice, the library in one flavour against the
# oracle build of the same flavour, bit for bit.
_BOTH_FLAVOURS = ("test_gpu_parity.py", "test_process_module_gpu.py")


def pytest_generate_tests(metafunc):
    if os.path.basename(str(metafunc.definition.fspath)) not in _BOTH_FLAVOURS:
        return
    if metafunc.definition.get_closest_marker("gpu") is None:
        return
    flavours = ["separate"] if metafunc.definition.get_closest_marker("separate_only") else ["separate", "contracted"]
    metafunc.parametrize("arithmetic", flavours, indirect=True)


@pytest.fixture(autouse=True)
def arithmetic(request):
    """Puts the library AND the oracle into the named arithmetic flavour for the duration of one test (autouse, so that the
    parametrisation above reaches tests that do not name it; a test without the parameter is left alone)."""
    name = getattr(request, "param", None)
    if name is None:
        yield "separate"
        return
    import oracle
    from canvas_amd import _lib
    lib = _lib.load()
    before = lib.cvs_set_arithmetic(_lib.ARITH_CONTRACTED if name == "contracted" else _lib.ARITH_SEPARATE)
    ctx = oracle.flavour("fma" if name == "contracted" else "gcc")
    ctx.__enter__()
    try:
        yield name
    finally:
        ctx.__exit__(None, None, None)
        lib.cvs_set_arithmetic(before if before >= 0 else _lib.ARITH_SEPARATE)


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
