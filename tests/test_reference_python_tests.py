"""The reference's own Python tests that need no pixels, run unmodified where its tree is mounted (this container; the GPU
box has no /root/reference), and the pixel expectations of tests/canvas/VideoSourceRefConnector.py restated for the GPU box."""
import importlib.util
import os
import unittest

import pytest

REF = "/root/reference/tests"


def _run_reference_unittest(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    import sys
    keep, sys.dont_write_bytecode = sys.dont_write_bytecode, True     # the reference tree is read-only: leave no __pycache__ in it
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = keep
    result = unittest.TextTestRunner(stream=open(os.devnull, "w")).run(unittest.defaultTestLoader.loadTestsFromModule(mod))
    return mod, result


def test_reference_basetypes_tests_run_unmodified():
    """/root/reference/tests/basetypes.py imports fluggo.media.basetypes -- the one of THIS repo (the extension module
    imports it at init for v2i / v2f / box2i / box2f / rgba, src/process/basetypes.c:117-148)."""
    path = os.path.join(REF, "basetypes.py")
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    import fluggo.media.basetypes as ours
    mod, result = _run_reference_unittest(path, "reference_basetypes_tests")
    assert mod.v2f is ours.v2f and os.path.dirname(ours.__file__).startswith(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    assert result.testsRun >= 2 and result.wasSuccessful(), (result.failures, result.errors)


@pytest.fixture(scope="module")
def process():
    from canvas_amd import _lib
    lib = _lib.load()
    if lib.cvs_init(0) != 0:
        pytest.skip("no HIP device")
    from fluggo.media import process as p
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("channel", [0, 1, 2])
def test_ref_connector_expectations_through_a_pass_through(process, channel):
    """tests/canvas/VideoSourceRefConnector.py:37-83: a solid whose colour ramps (0,0,0,1) -> (100,0,0,1) over 100 frames,
    behind the editor's stream object -- plugins.VideoStream, a SUBCLASS of process.VideoPassThroughFilter
    (fluggo/editor/plugins/_source.py:399) -- pulled as f32 at box2i(0,0,0,0): frame i has the ramping channel == i to 6
    places, the others 0, alpha 1 (check_red / check_green).  The connector re-targets its source while it lives
    (set_source): both orders are pulled."""
    from fluggo.media.basetypes import box2i

    class VideoStream(process.VideoPassThroughFilter):          # what the editor's plugin layer does
        def __init__(self, source):
            process.VideoPassThroughFilter.__init__(self, source)

    hi = [0.0, 0.0, 0.0, 1.0]
    hi[channel] = 100.0
    solid = process.SolidColorVideoSource(process.LerpFunc((0, 0, 0, 1), tuple(hi), 100))
    other = process.SolidColorVideoSource((9.0, 9.0, 9.0, 1.0))
    conn = VideoStream(other)
    conn.set_source(solid)
    for i in range(5):
        c = conn.get_frame_f32(i, box2i(0, 0, 0, 0)).pixel(0, 0)
        vals = [c.r, c.g, c.b]
        for k in range(3):
            assert vals[k] == pytest.approx(float(i) if k == channel else 0.0, abs=5e-7), (i, vals)
        assert c.a == pytest.approx(1.0, abs=5e-7)
    conn.set_source(None)                                       # an offline source: no data, not an exception
    assert conn.get_frame_f32(0, box2i(0, 0, 0, 0)).pixel(0, 0) is None
