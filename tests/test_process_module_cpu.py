"""fluggo.media.process, CPU side: the module imports, exposes the reference's Python surface
(SURVEY.md section 8b), and its parameter / bookkeeping logic is right.  No pixels are produced here:
without a GPU every pull reports an empty window (and says why)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def process():
    try:
        from fluggo.media import process
    except ImportError:
        import __graft_entry__
        __graft_entry__.build()
        from fluggo.media import process
    return process


def test_surface_is_complete(process):
    for name in ["VideoSource", "RgbaFrameF16", "RgbaFrameF32", "SolidColorVideoSource", "EmptyVideoSource",
                 "VideoGainOffsetFilter", "VideoMixFilter", "VideoScaler", "VideoPassThroughFilter", "VideoSequence",
                 "VideoWorkspace", "VideoPullQueue", "FrameFunction", "LerpFunc", "LinearFrameFunc",
                 "AnimationFunc", "AnimationPoint", "POINT_HOLD", "POINT_LINEAR", "FrameFuncPassThroughFilter", "Pulldown23RemovalFilter",
                 "CodedImageSource", "CodedImage", "DVReconstructionFilter", "DVSubsampleFilter",
                 "get_frame_time", "get_time_frame", "time_get_frame", "enable_glib_logging",
                 "create_offscreen_gl_context", "set_current_gl_context", "check_context_supported"]:
        assert hasattr(process, name), name
    for t in [process.SolidColorVideoSource, process.VideoMixFilter, process.VideoWorkspace, process.VideoSequence,
              process.VideoPassThroughFilter, process.RgbaFrameF16]:
        assert issubclass(t, process.VideoSource)
        assert hasattr(t, "get_frame_f16") and hasattr(t, "get_frame_f32")


def test_every_source_carries_the_named_capsule(process):
    from fluggo.media.basetypes import box2i
    solid = process.SolidColorVideoSource((0, 0, 0, 1))
    objs = [solid, process.EmptyVideoSource(), process.VideoGainOffsetFilter(solid), process.VideoMixFilter(solid, solid, 0.5),
            process.VideoScaler(solid, (0, 0), (0, 0), (2, 2), box2i(0, 0, 9, 9)), process.VideoPassThroughFilter(solid),
            process.VideoSequence(), process.VideoWorkspace()]
    for o in objs:
        cap = o._video_frame_source_funcs
        assert type(cap).__name__ == "PyCapsule" and '"_video_frame_source_funcs"' in repr(cap)
    with pytest.raises(Exception):
        process.VideoPassThroughFilter(object())            # no capsule -> refused (src/process/main.c:52-58)
    process.VideoPassThroughFilter(None)                    # None is "no source"


def test_lerp_func_kat(process):
    """tests/process/frame_func.py:13-29 of the reference."""
    func = process.LerpFunc((1.0, 2.0, 3.0, 4.0), (-1.0, -2.0, -3.0, -4.0), 4)
    want = {0: (1.0, 2.0, 3.0, 4.0), 1: (0.5, 1.0, 1.5, 2.0), 2: (0.0, 0.0, 0.0, 0.0), 3: (-0.5, -1.0, -1.5, -2.0), 4: (-1.0, -2.0, -3.0, -4.0)}
    for frame, values in want.items():
        assert func.get_values(frame)[0] == pytest.approx(values)
    got = func.get_values([4, 1, 2, 0, 3])
    for g, f in zip(got, [4, 1, 2, 0, 3]):
        assert g == pytest.approx(want[f])
    lin = process.LinearFrameFunc(2.0, -1.0)
    assert lin.get_values([0, 1.5])[1] == pytest.approx((2.0, 0, 0, 0))
    assert isinstance(func, process.FrameFunction)
    with pytest.raises(Exception):
        process.LerpFunc((0,), (1,), 0)


def test_basetypes(process):
    from fluggo.media.basetypes import box2f, box2i, rgba, v2f, v2i
    assert box2i(0, 0, 3, 3) == box2i((0, 0), (3, 3)) == ((0, 0), (3, 3))
    assert box2i(0, 0, 3, 3).width == 4 and box2i(0, 0, 3, 2).height == 3 and box2i(0, 0, 3, 3).size() == v2i(4, 4)
    assert box2i().empty() and not box2i() and box2i(2, 2, 1, 5).empty() and bool(box2i(0, 0, 0, 0))
    assert v2i(1, 2) + v2i(3, 4) == (4, 6) and v2f((1, 2)) - (0.5, 0.5) == (0.5, 1.5)
    assert rgba(1, 0.5, 0.25) == (1.0, 0.5, 0.25, 1.0)
    assert repr(box2i(0, 0, 3, 3)) == "box2i(v2i(0, 0), v2i(3, 3))"
    assert box2f(0, 0, 1.5, 2).width == 2.5


def test_frame_time_functions(process):
    from fractions import Fraction
    ntsc = Fraction(30000, 1001)
    assert process.get_frame_time(ntsc, 0) == 1
    for f in (0, 1, 29, 30, 1000):
        assert process.get_time_frame(ntsc, process.get_frame_time(ntsc, f)) == f
    assert process.get_frame_time(24, 24) == 10 ** 9 + 1


def test_sequence_list_protocol(process):
    red = process.SolidColorVideoSource((1, 0, 0, 1))
    seq = process.VideoSequence()
    seq.append((red, 1, 10))
    seq.append((red, 1, 5))
    seq.insert(1, (red, 0, 3))
    assert len(seq) == 3
    assert [seq.get_start_frame(i) for i in range(3)] == [0, 10, 13]
    assert seq[1][2] == 3 and seq[1][0] is red
    seq[0] = (red, 2, 4)
    assert [seq.get_start_frame(i) for i in range(3)] == [0, 4, 7]
    del seq[1]
    assert len(seq) == 2 and seq.get_start_frame(1) == 4
    with pytest.raises(ValueError):
        seq.append((red, 0, -1))
    with pytest.raises(IndexError):
        seq[5]


def test_workspace_items(process):
    red = process.SolidColorVideoSource((1, 0, 0, 1))
    green = process.SolidColorVideoSource((0, 1, 0, 1))
    ws = process.VideoWorkspace()
    a = ws.add(source=red, x=10, length=5, z=3, offset=0, tag="a")
    b = ws.add(source=green, x=0, length=20, z=9, offset=7, tag="b")
    assert len(ws) == 2 and ws[0] is b and ws[1] is a
    assert (a.x, a.length, a.z, a.offset, a.tag) == (10, 5, 3, 0, "a") and a.source is red
    a.update(x=-4, source=green, tag=None)
    assert a.x == -4 and a.source is green and a.tag is None and ws[0] is a
    ws.remove(b)
    assert len(ws) == 1
    with pytest.raises(Exception):
        b.x


def test_passthrough_is_subclassable_and_has_properties(process):
    class Stream(process.VideoPassThroughFilter):            # fluggo/editor/plugins/_source.py:399 does this
        def __init__(self, source):
            process.VideoPassThroughFilter.__init__(self, source, offset=3, start_frame=None, end_frame=10)
            self.extra = "x"

    red = process.SolidColorVideoSource((1, 0, 0, 1))
    s = Stream(red)
    assert (s.offset, s.start_frame, s.end_frame, s.extra) == (3, None, 10, "x") and s.source is red
    s.offset, s.start_frame = 5, 2
    s.set_source(None)
    assert (s.offset, s.start_frame, s.source) == (5, 2, None)
    g = process.VideoGainOffsetFilter(red)
    assert (g.gain, g.offset) == (1.0, 0.0) and g.source is red
    g.gain = process.LinearFrameFunc(1, 0)
    assert isinstance(g.gain, process.LinearFrameFunc)


def test_animation_func_key_points(process):
    """The curve of tests/process/frame_func.py:33-66 (hold 4 @0, linear 2 @1, linear 6 @2), asked in order and
    out of order -- lookups keep no cursor here, so order cannot matter -- plus the edge rules of
    AnimationFunc.c:408-463."""
    f = process.AnimationFunc()
    assert len(f) == 0 and f.get_values(7.0)[0] == (0.0, 0.0, 0.0, 0.0)
    f.add(process.AnimationPoint(process.POINT_HOLD, 0.0, 4.0))
    f.add(process.AnimationPoint(process.POINT_LINEAR, 1.0, 2.0))
    f.add(process.AnimationPoint(process.POINT_LINEAR, 2.0, 6.0))
    want = {-0.5: 4.0, 0.0: 4.0, 0.25: 4.0, 0.5: 4.0, 0.75: 4.0, 1.0: 2.0, 1.25: 3.0, 1.5: 4.0, 1.75: 5.0, 2.0: 6.0, 2.5: 6.0}
    for frame in list(want) + [-0.5, 0.75, 1.25, 2.5, 0.0, 2.0, 0.25, 1.75, 0.5, 1.0, 1.5]:
        assert f.get_values(frame)[0][0] == pytest.approx(want[frame])
    assert [v[0] for v in f.get_values([2.5, 1.25, -0.5])] == pytest.approx([6.0, 3.0, 4.0])
    # four slots, tuples shorter than four are zero-filled
    g = process.AnimationFunc()
    g.add(process.POINT_LINEAR, 10, (1.0, 2.0))
    g.add(process.POINT_LINEAR, 20, (5.0, 0.0, 0.0, 8.0))
    assert g.get_values(15)[0] == pytest.approx((3.0, 1.0, 0.0, 4.0))
    with pytest.raises(ValueError):
        g.add(process.POINT_HOLD, 1, ())
    with pytest.raises(ValueError):
        g.add(process.POINT_HOLD, 1, (1, 2, 3, 4, 5))
    with pytest.raises(Exception):
        process.AnimationPoint(7, 0.0, 1.0)


def test_animation_points_move_and_leave(process):
    """tests/process/frame_func.py:68-83: points stay sorted when a frame is reassigned."""
    f = process.AnimationFunc()
    p1 = f.add(process.POINT_HOLD, 0.0, 4.0)
    p2 = f.add(process.POINT_LINEAR, 2.0, 6.0)
    p3 = f.add(process.POINT_LINEAR, 1.0, 2.0)
    assert (f[0], f[1], f[2]) == (p1, p3, p2)
    p3.frame = 3.0
    assert (f[0], f[1], f[2]) == (p1, p2, p3) and p3.frame == 3.0 and p3.type == process.POINT_LINEAR
    assert p3.value == (2.0, 0.0, 0.0, 0.0)
    with pytest.raises(Exception):
        process.AnimationFunc().add(p3)            # already owned
    f.remove(p2)
    assert len(f) == 2 and f.get_values(2.0)[0][0] == 4.0          # p1 holds until p3
    f.remove(p2)                                   # not ours any more: ignored
    other = process.AnimationFunc()
    assert other.add(p2) is p2 and other[0] is p2
    with pytest.raises(IndexError):
        f[2]


def test_animation_func_drives_filter_parameters(process):
    """The editor's crossfade curve (fluggo/editor/graph/video.py:154-159) plugs into VideoMixFilter / gain."""
    mix = process.AnimationFunc()
    mix.add(process.POINT_HOLD, -1.0, 0.0)
    fade = mix.add(process.POINT_LINEAR, 0.0, 0.0)
    out = mix.add(process.POINT_HOLD, 0.0, 1.0)
    fade.frame, out.frame = 10.0, 20.0
    assert [v[0] for v in mix.get_values([0, 10, 15, 20, 30])] == pytest.approx([0.0, 0.0, 0.5, 1.0, 1.0])
    a, b = process.SolidColorVideoSource((1, 0, 0, 1)), process.SolidColorVideoSource((0, 1, 0, 1))
    process.VideoMixFilter(a, b, mix)
    process.VideoGainOffsetFilter(a, gain=mix, offset=0.0)
    assert hasattr(mix, "_frame_function_funcs")


def test_frame_func_pass_through_filter(process):
    """src/process/FrameFuncPassThroughFilter.c:59-88: the upstream function at frame + offset; constants pass as they are."""
    inner = process.LerpFunc((0.0, 10.0, 0.0, 0.0), (10.0, 20.0, 0.0, 0.0), 10.0)
    f = process.FrameFuncPassThroughFilter(inner, offset=2.5)
    assert isinstance(f, process.FrameFunction) and f.source() is inner and f.offset == 2.5
    assert f.get_values([0, 1]) == inner.get_values([2.5, 3.5])
    f.offset = 0.0
    assert f.get_values(4) == inner.get_values(4)
    const = process.FrameFuncPassThroughFilter((1.0, 2.0, 3.0, 4.0), offset=9.0)
    assert const.source() is None and const.get_values([0, 5]) == [(1.0, 2.0, 3.0, 4.0)] * 2
    const.set_source(inner)
    assert const.get_values(1) == inner.get_values(10.0)
    # it is itself a frame function: a filter parameter can be driven through it
    gain = process.VideoGainOffsetFilter(process.SolidColorVideoSource((0, 0, 0, 1)), gain=process.FrameFuncPassThroughFilter(inner, 1.0))
    assert gain.gain.source() is inner

    class Shifted(process.FrameFuncPassThroughFilter):       # Py_TPFLAGS_BASETYPE there (:176)
        pass
    assert Shifted(inner, 1.0).get_values(0) == inner.get_values(1.0)


def test_pulldown_cadence_table(process):
    """Pulldown23RemovalFilter.c:51-71.  The comment table there lists, per cadence offset, which source frame each of
    the four output frames of a cycle comes from; offsets 0-3 follow it.  For offset 4 the CODE sends the whole-frame
    case `frameOffset == 3` to base + 4 with base already a cycle ahead, so output 0 maps to source 5 where the comment
    says 0 -- what the code does is what is reproduced."""
    import ctypes as C
    import oracle
    from canvas_amd import _lib
    lib, orc = _lib.load(), oracle.lib()
    a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()

    def frames(fn, offset, i, x, y):
        mixed = fn(offset, i, C.byref(x), C.byref(y))
        return (x.value, y.value) if mixed else (x.value,)
    table = {0: [(0,), (1,), (2, 3), (4,)], 1: [(0,), (1, 2), (3,), (4,)], 2: [(0, 1), (2,), (3,), (4,)], 3: [(1,), (2,), (3,), (4, 5)]}
    for offset, cycle in table.items():
        for k in range(3):                                    # three cycles: +5 source frames per 4 output frames
            for i, want in enumerate(cycle):
                assert frames(lib.cvs_pulldown23_frames, offset, 4 * k + i, a, b) == tuple(w + 5 * k for w in want), (offset, k, i)
    assert [frames(lib.cvs_pulldown23_frames, 4, i, a, b) for i in range(5)] == [(5,), (1,), (2,), (3, 4), (10,)]
    for offset in range(5):
        for i in range(-9, 40):
            assert frames(lib.cvs_pulldown23_frames, offset, i, a, b) == frames(orc.orc_pulldown23_frames, offset, i, c, d), (offset, i)


def test_diagnostics_reach_python_logging(process):
    """enable_glib_logging (main.c:171-191, 272-329): library diagnostics arrive at logging.getLogger(domain)."""
    import io
    import logging
    from fluggo.media.basetypes import box2i
    stream = io.StringIO()
    handler = logging.StreamHandler(stream)
    logger = logging.getLogger("fluggo.media.cprocess")
    logger.addHandler(handler)
    try:
        process.enable_glib_logging(True)
        if process.check_context_supported():
            pytest.skip("a GPU is present: nothing fails here")
        frame = process.SolidColorVideoSource((1, 0, 0, 1)).get_frame_f16(0, box2i(0, 0, 3, 3))
        assert frame.current_window.empty()
        assert "no CPU path" in stream.getvalue()
    finally:
        process.enable_glib_logging(False)
        logger.removeHandler(handler)


def test_pull_queue_worker_count(process):
    process.VideoPullQueue(workers=4)
    for bad in (0, 17):
        with pytest.raises(ValueError):
            process.VideoPullQueue(workers=bad)


def test_coded_image_sources(process):
    """CodedImageSource.c:53-102,118-223: a Python subclass feeds planes through the capsule; the base has nothing."""
    class Planes(process.CodedImageSource):
        def get_frame(self, frame):
            if frame < 0:
                return None
            return [process.CodedImage(bytearray([frame % 256]) * (720 * 480), 720, 480),
                    process.CodedImage(bytearray(180 * 480), 180, 480), process.CodedImage(bytearray(180 * 480), 180, 480)]

    class Short(process.CodedImageSource):
        def get_frame(self, frame):
            return [process.CodedImage(bytearray(10), 720, 480)]          # wrong size: refused, like the reference's warning

    src = Planes()
    out = process.CodedImageSource.get_frame(src, 7)                       # base method -> capsule -> the override
    assert [(p.stride, p.line_count) for p in out] == [(720, 480), (180, 480), (180, 480)] and out[0].data[:2] == b"\x07\x07"
    assert process.CodedImageSource.get_frame(src, -1) is None
    assert process.CodedImageSource().get_frame(0) is None
    assert process.CodedImageSource.get_frame(Short(), 0) is None
    recon = process.DVReconstructionFilter(src)
    assert isinstance(recon, process.VideoSource) and hasattr(recon, "_video_frame_source_funcs")
    sub = process.DVSubsampleFilter(process.SolidColorVideoSource((0.2, 0.3, 0.4, 1.0)))
    assert isinstance(sub, process.CodedImageSource) and hasattr(sub, "_coded_image_source_funcs")
    with pytest.raises(Exception):
        process.DVReconstructionFilter(process.EmptyVideoSource())
    with pytest.raises(Exception):
        process.DVSubsampleFilter(src)


def test_without_a_gpu_pulls_are_empty_and_loud(process):
    from canvas_amd import _lib
    if _lib.load().cvs_device_count() > 0:
        pytest.skip("a GPU is present")
    from fluggo.media.basetypes import box2i
    frame = process.SolidColorVideoSource((1, 0.5, 0.25, 1)).get_frame_f16(0, box2i(0, 0, 3, 3))
    assert frame.current_window.empty() and frame.full_window == box2i(0, 0, 3, 3)
    assert frame.pixel(0, 0) is None
    assert "no CPU path" in process.last_error()
    assert process.check_context_supported() is False


def test_extension_links_only_the_library_not_the_oracle():
    import glob
    so = glob.glob(os.path.join(ROOT, "fluggo", "media", "process*.so"))
    assert so
    out = subprocess.run(["ldd", so[0]], stdout=subprocess.PIPE, text=True).stdout
    assert "libcanvas_hip.so" in out and "oracle" not in out
    for f in glob.glob(os.path.join(ROOT, "canvas_amd", "pyext", "*.[ch]")):
        assert "orc_" not in open(f).read()



def test_reference_own_frame_function_tests_run_unmodified(process):
    """Where the reference tree is mounted (this container, not the GPU box): its own unittest file for frame
    functions, loaded as it is and run against THIS module -- LerpFunc and AnimationFunc need no pixels."""
    import importlib.util
    import unittest
    path = "/root/reference/tests/process/frame_func.py"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    spec = importlib.util.spec_from_file_location("reference_frame_func_tests", path)
    mod = importlib.util.module_from_spec(spec)
    import sys
    keep, sys.dont_write_bytecode = sys.dont_write_bytecode, True     # the reference tree is read-only: leave no __pycache__ in it
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = keep
    assert mod.process is process                       # it imported fluggo.media.process from this repo
    result = unittest.TextTestRunner(stream=open(os.devnull, "w")).run(unittest.defaultTestLoader.loadTestsFromModule(mod))
    assert result.testsRun >= 3 and result.wasSuccessful(), (result.failures, result.errors)


def test_frame_owner_and_arithmetic_functions(process):
    """Added beside the reference's surface in round 4: the frame-to-device rule and the arithmetic flavour switch."""
    assert [process.frame_owner(g, 4) for g in range(-2, 9)] == [g % 4 for g in range(-2, 9)]
    with pytest.raises(ValueError):
        process.frame_owner(3, 0)
    assert isinstance(process.device_count(), int)
    before = process.get_arithmetic()
    assert before in ("separate", "contracted")
    try:
        assert process.set_arithmetic("contracted") == before
        assert process.get_arithmetic() == "contracted"
        with pytest.raises(ValueError):
            process.set_arithmetic("fast")
    finally:
        process.set_arithmetic(before)
