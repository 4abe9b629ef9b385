"""fluggo.media.process on the GPU: the reference's own Python tests re-expressed (same inputs, same
assertions, force_gl accepted and ignored), plus graphs checked against the CPU oracle.

Reference tests mirrored:
  tests/process/video/RgbaFrameF16.py:6-23            test_solid
  tests/process/video/SolidColorVideoSource.py:13-55   const colour / const window / moving colour / moving window
  tests/process/video/VideoWorkspace.py:12-38          10 000 random add / move / remove / pull operations
  tests/canvas/sequence.py:58-100                      check1: cut, cut, crossfade -- built by hand from the
                                                       same process.* objects the editor's graph code creates
"""
import ctypes as C
import random
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def process():
    from fluggo.media import process
    assert process.check_context_supported(), process.last_error()
    return process


@pytest.fixture(scope="module")
def bt():
    from fluggo.media import basetypes
    return basetypes


def almost(a, b, places):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert round(x - y, places) == 0, (a, b)


def getcolor(source, frame, bt):
    return source.get_frame_f32(frame, bt.box2i(0, 0, 0, 0)).pixel(0, 0)


# ---------------------------------------------------------------- RgbaFrameF16.py

def test_solid_frame_and_rewindow(process, bt):
    color = (1.0, 0.5, 0.333333, 0.2)
    solid = process.SolidColorVideoSource(color, bt.box2i((0, 0), (2, 2)))
    frame = solid.get_frame_f16(0, bt.box2i((0, 0), (3, 3)))
    assert frame.current_window == bt.box2i(0, 0, 2, 2)
    assert frame.full_window == bt.box2i(0, 0, 3, 3)
    almost(frame.pixel(0, 0), color, 3)
    assert frame.pixel(3, 3) is None and len(frame) == 16
    frame2 = frame.get_frame_f16(0, bt.box2i(-1, -1, 1, 1))
    assert frame2.current_window == bt.box2i(0, 0, 1, 1)
    assert frame2.full_window == bt.box2i(-1, -1, 1, 1)
    almost(frame2.pixel(0, 0), color, 3)
    raw = frame.to_argb32_bytes()
    assert len(raw) == 9 * 4


# ---------------------------------------------------------------- SolidColorVideoSource.py

@pytest.mark.parametrize("force_gl", [True, False])
def test_solid_const_color_and_window(process, bt, force_gl):
    color = (1.0, 0.5, 0.333333, 0.2)
    solid = process.SolidColorVideoSource(color)
    frame = solid.get_frame_f32(0, bt.box2i(0, 0, 3, 3), force_gl=force_gl)
    assert frame.current_window == bt.box2i(0, 0, 3, 3)
    almost(frame.pixel(0, 0), color, 6)
    solid = process.SolidColorVideoSource(color, bt.box2i(0, 0, 2, 2))
    frame = solid.get_frame_f32(0, bt.box2i(0, 0, 3, 3), force_gl=force_gl)
    assert frame.current_window == bt.box2i(0, 0, 2, 2)
    almost(frame.pixel(0, 0), color, 6)
    frame2 = frame.get_frame_f32(0, bt.box2i(-1, -1, 1, 1), force_gl=force_gl)
    assert frame2.current_window == bt.box2i(0, 0, 1, 1)
    almost(frame2.pixel(0, 0), color, 6)


def test_solid_moving_color_and_window(process, bt):
    solid = process.SolidColorVideoSource(process.LerpFunc((0.5, 0.25, 2.0, 1.0), (-0.5, -0.25, -2.0, 0.0), 2))
    almost(getcolor(solid, 0, bt), (0.5, 0.25, 2.0, 1.0), 6)
    almost(getcolor(solid, 1, bt), (0.0, 0.0, 0.0, 0.5), 6)
    almost(getcolor(solid, 2, bt), (-0.5, -0.25, -2.0, 0.0), 6)
    solid = process.SolidColorVideoSource(bt.rgba(0.0, 0.0, 1.0, 1.0), process.LerpFunc((-2, -2, 2, 2), (-4, -4, 0, 6), 2))
    for i, want in enumerate([(-2, -2, 2, 2), (-3, -3, 1, 4), (-4, -4, 0, 6)]):
        frame = solid.get_frame_f32(i, bt.box2i(-5, -5, 5, 6))
        assert frame.current_window == bt.box2i(*want)


# ---------------------------------------------------------------- sequence.py check1

def ramp(process, channel):
    hi = [0, 0, 0, 1]
    hi[channel] = 100
    return process.SolidColorVideoSource(process.LerpFunc((0, 0, 0, 1), tuple(hi), 100))


def test_sequence_cut_cut_crossfade(process, bt):
    red, green, blue = ramp(process, 0), ramp(process, 1), ramp(process, 2)
    fade = process.VideoMixFilter(src_a=process.VideoPassThroughFilter(green, offset=6),
                                  src_b=process.VideoPassThroughFilter(blue, offset=1),
                                  mix_b=process.LerpFunc((0,), (1,), 5))
    seq = process.VideoSequence()
    seq.append((red, 1, 10))
    seq.append((green, 1, 5))
    seq.append((fade, 0, 5))
    seq.append((blue, 6, 5))
    colors = [getcolor(seq, i, bt) for i in range(30)]
    for i in range(0, 10):
        almost(colors[i], (i + 1.0, 0, 0, 1), 6)
    for i in range(10, 15):
        almost(colors[i], (0, i - 10 + 1.0, 0, 1), 6)
    for i in range(15, 20):
        m = (i - 15) / 5.0
        almost(colors[i], (0.0, (i - 10 + 1.0) * (1.0 - m), (i - 15 + 1.0) * m, 1.0), 6)
    for i in range(20, 25):
        almost(colors[i], (0, 0, i - 15 + 1.0, 1), 6)
    for i in range(25, 30):
        assert colors[i] is None
    assert getcolor(seq, -1, bt) is None


def test_sequence_crossfade_driven_by_animation_points(process, bt):
    """The same cut/cut/crossfade, with mix_b built the way the editor builds it
    (fluggo/editor/graph/video.py:154-159: hold 0 until the fade point, linear to the out point, hold 1)."""
    red, green, blue = ramp(process, 0), ramp(process, 1), ramp(process, 2)
    mix_b = process.AnimationFunc()
    mix_b.add(process.POINT_HOLD, -1.0, 0.0)
    fade_point = mix_b.add(process.POINT_LINEAR, 0.0, 0.0)
    out_point = mix_b.add(process.POINT_HOLD, 0.0, 1.0)
    out_point.frame = 5.0
    assert (mix_b[1], mix_b[2]) == (fade_point, out_point)
    fade = process.VideoMixFilter(src_a=process.VideoPassThroughFilter(green, offset=6),
                                  src_b=process.VideoPassThroughFilter(blue, offset=1), mix_b=mix_b)
    seq = process.VideoSequence()
    seq.append((red, 1, 10))
    seq.append((green, 1, 5))
    seq.append((fade, 0, 5))
    seq.append((blue, 6, 5))
    for i in range(15, 20):
        m = (i - 15) / 5.0
        almost(getcolor(seq, i, bt), (0.0, (i - 10 + 1.0) * (1.0 - m), (i - 15 + 1.0) * m, 1.0), 6)
    # moving the fade point re-times the transition without touching the graph
    fade_point.frame = 2.0
    almost(getcolor(seq, 16, bt), (0.0, 7.0, 0.0, 1.0), 6)            # still holding source a
    m = (4 - 2.0) / 3.0
    almost(getcolor(seq, 19, bt), (0.0, 10.0 * (1.0 - m), 5.0 * m, 1.0), 5)


# ---------------------------------------------------------------- VideoWorkspace.py

def test_workspace_random_operations(process, bt):
    red, green, blue = ramp(process, 0), ramp(process, 1), ramp(process, 2)
    rnd = random.Random(5)
    workspace = process.VideoWorkspace()
    for _ in range(10000):
        action = rnd.randint(1, 7)
        n = len(workspace)
        if action == 1 and n:
            workspace[rnd.randrange(n)].update(x=rnd.randint(0, 1000))
        elif action == 2 and n:
            workspace[rnd.randrange(n)].update(z=rnd.randint(-10, 10))
        elif action == 3 and n:
            workspace[rnd.randrange(n)].update(length=rnd.randint(1, 100))
        elif action == 4 and n:
            workspace[rnd.randrange(n)].update(offset=rnd.randint(-20, 20))
        elif action == 5 and n:
            workspace.remove(workspace[rnd.randrange(n)])
        elif action == 6:
            for _ in range(3):
                getcolor(workspace, rnd.randint(-100, 1100), bt)
        else:
            workspace.add(source=rnd.choice((red, green, blue)), x=rnd.randint(0, 1000), z=rnd.randint(-10, 10),
                          length=rnd.randint(1, 100), offset=rnd.randint(-20, 20))


def test_workspace_stack_values(process, bt, orc):
    """Three windowed translucent solids over an opaque one, through the Python API, against the oracle's
    over applied bottom-to-top (workspace.c:530-544)."""
    from canvas_amd.abi import HostFrame
    full = (0, 0, 31, 17)
    spec = [((0.2, 0.4, 0.6, 1.0), full, 0), ((0.9, 0.1, 0.3, 0.5), (4, 2, 20, 12), 5), ((0.3, 0.8, 0.1, 0.25), (10, 3, 31, 9), 2)]
    ws = process.VideoWorkspace()
    for color, win, z in spec:
        ws.add(source=process.SolidColorVideoSource(color, bt.box2i(*win)), x=0, length=10, z=z, offset=0)
    got = ws.get_frame_f32(3, bt.box2i(*full))
    assert got.current_window == bt.box2i(*full)
    acc = None
    for color, win, z in sorted(spec, key=lambda s: s[2]):
        arr = np.zeros((18, 32, 4), np.float32)
        arr[win[1]:win[3] + 1, win[0]:win[2] + 1] = np.array(color, np.float32)
        layer = HostFrame(full, np.float32, arr, win)
        if acc is None:
            acc = layer
        else:
            orc.lib().orc_mix_over_f32(acc.ref(), layer.ref(), C.c_float(1.0))
    for y in range(18):
        for x in (0, 5, 12, 25, 31):
            almost(got.pixel(x, y), tuple(float(v) for v in acc.array[y, x]), 7)


# ---------------------------------------------------------------- config 1: SolidColor -> gain -> pull

def test_config1_solid_gain_pull(process, bt, orc):
    solid = process.SolidColorVideoSource((0.25, 0.5, 0.75, 1.0))
    gain = process.VideoGainOffsetFilter(solid, gain=1.5, offset=0.0625)
    frame = gain.get_frame_f16(0, bt.box2i(0, 0, 1919, 1079))
    assert frame.current_window == bt.box2i(0, 0, 1919, 1079)
    # expected per the repo's definition of gain/offset: truncate(color) -> widen -> *1.5 + 0.0625 -> truncate
    c = orc.half_to_float(orc.float_to_half(np.array([0.25, 0.5, 0.75, 1.0], np.float32)))
    want = c.copy()
    want[:3] = (c[:3] * np.float32(1.5)).astype(np.float32) + np.float32(0.0625)
    want = orc.half_to_float(orc.float_to_half(want))
    for xy in [(0, 0), (1919, 1079), (960, 540)]:
        assert tuple(frame.pixel(*xy)) == tuple(float(v) for v in want)
    ns = process.time_get_frame(gain, 0, 99, (0, 0, 1919, 1079))        # the reference's timing primitive
    assert ns > 0
    print("config 1: 100 x 1920x1080 SolidColor->gain->pull: %.1f ms, %.0f Mpx/s (host frames, PCIe included)" % (
        ns / 1e6, 100 * 1920 * 1080 / (ns / 1e9) / 1e6))


def test_scaler_node_matches_oracle(process, bt, orc):
    from canvas_amd.abi import HostFrame, v2f
    solid = process.SolidColorVideoSource((0.2, 0.4, 0.6, 0.8), bt.box2i(2, 1, 13, 7))
    scaler = process.VideoScaler(solid, target_point=(0, 0), source_point=(0, 0), scale_factors=(2.0, 2.0), source_rect=bt.box2i(0, 0, 15, 8))
    got = scaler.get_frame_f32(0, bt.box2i(0, 0, 31, 17))
    src = HostFrame((0, 0, 15, 8), np.float32, current_window=(2, 1, 13, 7))
    src.array[1:8, 2:14] = np.array([0.2, 0.4, 0.6, 0.8], np.float32)
    # the pull rectangle of video_scale.c:303-309 for this target is (-1,-1,16,9) clipped to source_rect
    want = HostFrame((0, 0, 31, 17), np.float32)
    orc.lib().orc_scale_bilinear_f32(want.ref(), v2f(0, 0), src.ref(), v2f(0, 0), v2f(2.0, 2.0))
    assert got.current_window == bt.box2i(*want.current_window.tuple())
    x0, y0, x1, y1 = want.current_window.tuple()
    for y in range(y0, y1 + 1, 3):
        for x in range(x0, x1 + 1, 5):
            almost(got.pixel(x, y), tuple(float(v) for v in want.array[y, x]), 7)


# ---------------------------------------------------------------- plugin protocol + pull queue

def test_foreign_host_only_source_plugs_in(process, bt):
    """A third-party source: a Python object whose capsule wraps a vtable with only get_frame_32 filled
    (no device slot) -- pulled through device-resident nodes."""
    from canvas_amd.abi import GET_FRAME_F32, video_frame_source_funcs

    def fill(self_ptr, idx, fp):
        f = fp.contents
        n = f.full_window.width * f.full_window.height
        arr = np.ctypeslib.as_array(C.cast(f.data, C.POINTER(C.c_float)), shape=(n, 4))
        arr[:] = (idx, 0.5, 0.25, 1.0)
        f.current_window = f.full_window

    cb = GET_FRAME_F32(fill)
    funcs = video_frame_source_funcs(0, C.cast(None, type(video_frame_source_funcs().get_frame)), cb, None)
    C.pythonapi.PyCapsule_New.restype = C.py_object
    C.pythonapi.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]

    class Foreign:
        _video_frame_source_funcs = C.pythonapi.PyCapsule_New(C.addressof(funcs), b"_video_frame_source_funcs", None)

    src = Foreign()
    chain = process.VideoGainOffsetFilter(process.VideoPassThroughFilter(src, offset=2), gain=2.0, offset=0.0)
    px = chain.get_frame_f32(5, bt.box2i(0, 0, 3, 3)).pixel(1, 1)
    almost(px, (14.0, 1.0, 0.5, 1.0), 6)


@pytest.mark.parametrize("workers", [None, 1, 6])
def test_pull_queue_delivers_frames(process, bt, workers):
    q = process.VideoPullQueue() if workers is None else process.VideoPullQueue(workers=workers)
    solid = ramp(process, 0)
    done, seen = threading.Event(), {}

    def callback(frame_index, frame, user_data):
        seen[frame_index] = (frame.pixel(0, 0), user_data)
        if len(seen) == 8:
            done.set()

    items = [q.enqueue(source=solid, frame_index=i, window=bt.box2i(0, 0, 1, 1), callback=callback, user_data="u%d" % i) for i in range(8)]
    assert done.wait(30), "callbacks did not arrive"
    for i in range(8):
        almost(seen[i][0], (float(i), 0, 0, 1), 6)
        assert seen[i][1] == "u%d" % i
    cancelled = q.enqueue(source=solid, frame_index=99, window=bt.box2i(0, 0, 1, 1), callback=callback, user_data=None)
    cancelled.cancel()
    del items


def test_pull_queue_may_lose_its_last_reference_on_its_own_worker(process, bt):
    """The last Python reference to a queue can go away while a request is pending: the pending item keeps the queue
    alive, the worker drops that reference after the callback, and the queue is then deallocated ON the worker thread --
    which must neither join itself nor touch the freed object on its way out."""
    import gc
    import threading
    solid = process.SolidColorVideoSource((0.25, 0.5, 0.75, 1.0), bt.box2i(0, 0, 31, 17))
    for workers in (1, 3):
        done, seen = threading.Event(), []

        def cb(i, frame, user):
            seen.append((i, frame.pixel(3, 3)))
            done.set()
        q = process.VideoPullQueue(workers=workers)
        item = q.enqueue(solid, 4, bt.box2i(0, 0, 31, 17), cb, None)
        del q                                               # the item holds the only reference now
        assert done.wait(30)
        assert seen[0][0] == 4 and abs(seen[0][1].g - 0.5) < 1e-3
        del item
        gc.collect()
    # the process is still healthy: another queue works
    done = threading.Event()
    q = process.VideoPullQueue()
    q.enqueue(solid, 0, bt.box2i(0, 0, 3, 3), lambda i, f, u: done.set(), None)
    assert done.wait(30)


def test_writer_worker_and_callback_on_one_animation_do_not_deadlock(process, bt):
    """Three parties on one AnimationFunc's lock: the main thread keeps adding and removing points (writer, called with
    the GIL), pull-queue workers read it through a solid's colour (readers without the GIL), and the queue's callbacks
    call get_values() on it (readers WITH the GIL, on the worker thread).  A writer that came back from its wait holding
    the lock and wanting the GIL used to meet a callback holding the GIL and wanting the lock."""
    anim = process.AnimationFunc()
    anim.add(process.POINT_LINEAR, 0.0, (0.0, 0.0, 0.0, 1.0))
    anim.add(process.POINT_LINEAR, 1000.0, (1000.0, 0.0, 0.0, 1.0))
    solid = process.SolidColorVideoSource(anim, bt.box2i(0, 0, 63, 63))
    passing = process.FrameFuncPassThroughFilter(anim, offset=1.0)
    q = process.VideoPullQueue(workers=3)
    stop, total, got, errors = threading.Event(), 600, [], []
    all_done = threading.Event()

    def callback(frame_index, frame, user_data):
        try:
            v = anim.get_values(float(frame_index))[0]
            w = passing.get_values([float(frame_index)])[0]
            got.append((frame_index, frame.pixel(1, 1), v, w))
        except Exception as e:                              # noqa: BLE001
            errors.append(e)
        if len(got) + len(errors) == total:
            all_done.set()

    def feeder():
        for i in range(total):
            q.enqueue(solid, i % 900, bt.box2i(0, 0, 63, 63), callback, None)

    t = threading.Thread(target=feeder, daemon=True)
    t.start()
    writes = 0
    while not all_done.is_set() and writes < 200000:
        p = anim.add(process.POINT_LINEAR, 2000.0 + writes % 7, (0.0, 0.0, 0.0, 1.0))      # beyond every frame pulled
        p.frame = 3000.0 + writes % 5
        anim.remove(p)
        passing.set_source(anim)
        writes += 1
    assert all_done.wait(60), "deadlock: %d of %d callbacks after %d writes" % (len(got), total, writes)
    stop.set()
    t.join(30)
    assert not errors, errors[:3]
    for i, px, v, w in got[::37]:
        assert abs(px.r - i) <= max(1.0, i) * 2e-3 and abs(v[0] - i) < 1e-9 and abs(w[0] - (i + 1)) < 1e-9, (i, px, v, w)


def test_preview_pulls_convert_on_the_device(process, bt, orc):
    """get_frame_argb32 equals get_frame_f16(...).to_argb32_bytes() (RgbaFrameF16.c:114-149 against the oracle);
    get_frame_rgba8 is the software widget's sRGB bytes (widget_gl.c:291-307)."""
    from canvas_amd.abi import HostFrame
    ws = process.VideoWorkspace()
    ws.add(source=process.SolidColorVideoSource((0.25, 0.5, 0.75, 1.0), bt.box2i(0, 0, 30, 20)), x=0, length=10, z=0)
    ws.add(source=process.SolidColorVideoSource((0.9, 0.1, 0.3, 0.4), bt.box2i(8, 5, 40, 30)), x=0, length=10, z=1)
    window = bt.box2i(-2, -2, 35, 25)
    frame = ws.get_frame_f16(3, window)
    raw, cur = ws.get_frame_argb32(3, window)
    assert cur == frame.current_window and bytes(raw) == bytes(frame.to_argb32_bytes())
    # against the oracle, from the pulled halfs
    cw = frame.current_window
    w, h = cw.max.x - cw.min.x + 1, cw.max.y - cw.min.y + 1
    host = HostFrame((cw.min.x, cw.min.y, cw.max.x, cw.max.y), np.uint16)
    for y in range(h):
        for x in range(w):
            host.array[y, x] = orc.float_to_half(np.array(frame.pixel(cw.min.x + x, cw.min.y + y), np.float32))
    want = np.zeros((h, w), np.uint32)
    orc.lib().orc_frame_to_bytes(want.ctypes.data_as(C.POINTER(C.c_uint32)), host.ref(), None, 1)
    assert bytes(raw) == want.tobytes()
    rgba, cur2 = ws.get_frame_rgba8(3, window)
    table = orc.transfer_table(3)
    orc.lib().orc_frame_to_rgba8_intent(want.ctypes.data_as(C.POINTER(C.c_uint32)), host.ref(), table.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_float(1.25))
    assert cur2 == cur and bytes(rgba) == want.tobytes()           # the widget's default rendering intent
    rgba, cur2 = ws.get_frame_rgba8(3, window, rendering_intent=1.0)
    orc.lib().orc_frame_to_rgba8_intent(want.ctypes.data_as(C.POINTER(C.c_uint32)), host.ref(), table.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_float(1.0))
    assert cur2 == cur and bytes(rgba) == want.tobytes()
    nothing, cur3 = process.EmptyVideoSource().get_frame_argb32(0, window)
    assert nothing is None and cur3.empty()


def test_dv_nodes_against_oracle(process, bt, orc):
    """Solid colour + window -> DVSubsampleFilter -> planes (oracle: orc_subsample_dv) -> a Python coded-image
    source -> DVReconstructionFilter -> frame (oracle: orc_reconstruct_dv)."""
    from canvas_amd.abi import HostFrame
    solid = process.SolidColorVideoSource((0.2, 0.45, 0.7, 1.0), bt.box2i(40, 30, 650, 400))
    planes = process.DVSubsampleFilter(solid).get_frame(0)
    assert [(p.stride, p.line_count) for p in planes] == [(720, 480), (180, 480), (180, 480)]
    full = (0, -1, 719, 478)
    host = HostFrame(full, np.uint16, current_window=(40, 30, 650, 400))
    host.array[...] = orc.float_to_half(np.array([0.2, 0.45, 0.7, 1.0], np.float32))
    want = [np.zeros((480, s), np.uint8) for s in (720, 180, 180)]
    orc.lib().orc_subsample_dv((C.c_void_p * 3)(*[w.ctypes.data for w in want]), (C.c_int * 3)(720, 180, 180), host.ref())
    for p in range(3):
        assert bytes(planes[p].data) == want[p].tobytes(), "plane %d" % p

    class Fixed(process.CodedImageSource):
        def get_frame(self, frame):
            return planes

    recon = process.DVReconstructionFilter(Fixed())
    window = bt.box2i(-4, -3, 725, 481)
    frame = recon.get_frame_f16(0, window)
    assert frame.current_window == bt.box2i(0, -1, 719, 478)
    theirs = HostFrame((-4, -3, 725, 481), np.uint16)
    orc.lib().orc_reconstruct_dv(theirs.ref(), (C.c_void_p * 3)(*[w.ctypes.data for w in want]), (C.c_int * 3)(720, 180, 180))
    for x, y in [(0, -1), (40, 30), (41, 30), (43, 31), (300, 200), (650, 400), (651, 401), (719, 478)]:
        got = orc.float_to_half(np.array(frame.pixel(x, y), np.float32))
        assert np.array_equal(got, theirs.array[y + 3, x + 4]), (x, y)
    # inside the solid's window the round trip lands near the colour it started from
    almost(frame.pixel(300, 200), (0.2, 0.45, 0.7, 1.0), 1)


def test_dv_footage_pipeline_against_oracle(process, bt, orc):
    """The chain real DV footage goes through: a coded-image source with different planes per frame ->
    DVReconstructionFilter -> Pulldown23RemovalFilter -> preview bytes.  Every stage against the oracle: reconstruct
    both source frames (orc_reconstruct_dv), weave them (orc_weave_fields_f16), convert (orc_frame_to_bytes)."""
    from canvas_amd import _lib
    from canvas_amd.abi import HostFrame
    lib = _lib.load()
    rng = np.random.default_rng(31)
    coded = {}

    def planes_of(frame):
        if frame not in coded:
            coded[frame] = [rng.integers(16, 236, (480, s), dtype=np.uint8) for s in (720, 180, 180)]
        return coded[frame]

    class Tape(process.CodedImageSource):
        def get_frame(self, frame):
            return [process.CodedImage(bytearray(p.tobytes()), p.shape[1], 480) for p in planes_of(frame)]

    film = process.Pulldown23RemovalFilter(process.DVReconstructionFilter(Tape()), 2)
    window = bt.box2i(0, -1, 719, 478)
    a, b = C.c_int(), C.c_int()
    seen = set()
    for i in (0, 1, 4, 5):
        mixed = lib.cvs_pulldown23_frames(2, i, C.byref(a), C.byref(b))
        seen.add(mixed)
        want = HostFrame((0, -1, 719, 478), np.uint16)
        pl = planes_of(a.value)
        orc.lib().orc_reconstruct_dv(want.ref(), (C.c_void_p * 3)(*[p.ctypes.data for p in pl]), (C.c_int * 3)(720, 180, 180))
        if mixed:
            other = HostFrame((0, -1, 719, 478), np.uint16)
            pl2 = planes_of(b.value)
            orc.lib().orc_reconstruct_dv(other.ref(), (C.c_void_p * 3)(*[p.ctypes.data for p in pl2]), (C.c_int * 3)(720, 180, 180))
            orc.lib().orc_weave_fields_f16(want.ref(), other.ref())
        raw, cur = film.get_frame_argb32(i, window)
        assert cur == window
        packed = np.zeros((480, 720), np.uint32)
        orc.lib().orc_frame_to_bytes(packed.ctypes.data_as(C.POINTER(C.c_uint32)), want.ref(), None, 1)
        assert bytes(raw) == packed.tobytes(), i
        got = film.get_frame_f16(i, window)
        for x, y in [(0, -1), (1, 0), (359, 240), (719, 477), (718, 478)]:
            assert np.array_equal(orc.float_to_half(np.array(got.pixel(x, y), np.float32)), want.array[y + 1, x]), (i, x, y)
    assert seen == {0, 1}


@pytest.mark.parametrize("offset", [0, 1, 2, 3, 4])
def test_pulldown_removal_node(process, bt, offset):
    """Pulldown23RemovalFilter over a source whose every frame is a different colour: whole frames come through as they
    are, the woven frame has the odd rows of one source frame and the even rows of the next (Pulldown23RemovalFilter.c:51-104)."""
    import ctypes as C
    from canvas_amd import _lib
    lib = _lib.load()
    source = process.SolidColorVideoSource(process.LerpFunc((0.0, 1.0, 0.25, 1.0), (1.0, 0.0, 0.75, 1.0), 32.0), bt.box2i(0, -1, 23, 12))
    node = process.Pulldown23RemovalFilter(source, offset)
    window = bt.box2i(0, -1, 23, 10)
    a, b = C.c_int(), C.c_int()
    seen_mixed = 0
    for i in range(0, 9):
        got = node.get_frame_f16(i, window)
        mixed = lib.cvs_pulldown23_frames(offset, i, C.byref(a), C.byref(b))
        first, second = source.get_frame_f16(a.value, window), source.get_frame_f16(b.value, window)
        assert got.current_window == first.current_window == window
        for y in range(window.min.y, window.max.y + 1):
            want = second if (mixed and y % 2 == 0) else first
            for x in (0, 11, 23):
                assert got.pixel(x, y) == want.pixel(x, y), (i, x, y)
        seen_mixed += mixed
        # pulled as f32 through the f16-native node (main.c:105-144): the same halfs, widened
        assert node.get_frame_f32(i, window).pixel(5, 0) == got.pixel(5, 0)
    assert seen_mixed >= 2
    assert process.Pulldown23RemovalFilter(None, 0).get_frame_f16(0, window).current_window.empty()


@pytest.mark.parametrize("nlayers", [1, 3, 5])
def test_workspace_of_half_native_clips_takes_the_fused_stack(process, bt, orc, nlayers):
    """Half-native items (here gain/offset nodes over solids: f16 slot only) pulled as f16: the workspace hands the
    stack to the chain kernel; results equal the oracle's widen -> over -> truncate, every pixel."""
    from canvas_amd import _lib
    from canvas_amd.abi import HostFrame
    full = (0, 0, 95, 53)
    colours = [(0.2, 0.4, 0.6, 1.0), (0.9, 0.1, 0.3, 0.5), (0.3, 0.8, 0.1, 0.25), (0.05, 0.5, 0.95, 0.75), (0.6, 0.6, 0.1, 0.1)]
    ws = process.VideoWorkspace()
    layers = []
    for z in range(nlayers):
        clip = process.VideoGainOffsetFilter(process.SolidColorVideoSource(colours[z]), gain=1.25, offset=0.03125)
        ws.add(source=clip, x=0, length=4, z=z, offset=0)
        # what the clip produces: truncate(colour) -> widen -> c * gain + offset (two rounded f32 ops) -> truncate
        c = orc.half_to_float(orc.float_to_half(np.array(colours[z], np.float32)))
        v = np.array([np.float32(np.float32(c[k] * np.float32(1.25)) + np.float32(0.03125)) for k in range(3)] + [c[3]], np.float32)
        layers.append(HostFrame(full, np.uint16, np.broadcast_to(orc.float_to_half(v), (54, 96, 4)).copy()))
    got = ws.get_frame_f16(1, bt.box2i(*full))
    assert _lib.load().cvs_chain_last_was_fused() == 1
    want = orc.chain_color_over(layers, None)
    assert got.current_window == bt.box2i(*full)
    for y in (0, 17, 53):
        for x in (0, 40, 95):
            assert np.array_equal(orc.float_to_half(np.array(got.pixel(x, y), np.float32)), want.array[y, x]), (x, y)
    # a windowed item sends the same call node by node, with the reference's window behaviour
    ws.add(source=process.VideoGainOffsetFilter(process.SolidColorVideoSource((1, 1, 1, 0.5), bt.box2i(10, 10, 30, 30))), x=0, length=4, z=99, offset=0)
    got2 = ws.get_frame_f16(1, bt.box2i(*full))
    assert _lib.load().cvs_chain_last_was_fused() == 0 and got2.current_window == bt.box2i(*full)
    assert got2.pixel(5, 5) == got.pixel(5, 5) and got2.pixel(20, 20) != got.pixel(20, 20)


def test_crossfade_of_half_native_clips_is_one_launch(process, bt, orc):
    """VideoMixFilter over two half-native clips pulled as f16: fused crossfade; same values as the f32 pull truncated."""
    from canvas_amd import _lib
    clip_a = process.VideoGainOffsetFilter(process.SolidColorVideoSource((0.2, 0.4, 0.6, 1.0)), gain=1.25, offset=0.03125)
    clip_b = process.VideoGainOffsetFilter(process.SolidColorVideoSource((0.9, 0.1, 0.3, 0.5)), gain=0.75, offset=0.0)
    fade = process.VideoMixFilter(src_a=clip_a, src_b=clip_b, mix_b=process.LerpFunc((0,), (1,), 10))
    window = bt.box2i(0, 0, 63, 35)
    for frame in (3, 5, 9):
        got16 = fade.get_frame_f16(frame, window)
        assert _lib.load().cvs_chain_last_was_fused() == 1
        got32 = fade.get_frame_f32(frame, window)
        for x, y in [(0, 0), (17, 9), (63, 35)]:
            want = orc.float_to_half(np.array(got32.pixel(x, y), np.float32))
            assert np.array_equal(orc.float_to_half(np.array(got16.pixel(x, y), np.float32)), want), (frame, x, y)
    # the ends of the fade are single pulls (video_mix.c:46-71)
    almost(fade.get_frame_f16(0, window).pixel(5, 5), clip_a.get_frame_f16(0, window).pixel(5, 5), 7)
    almost(fade.get_frame_f16(10, window).pixel(5, 5), clip_b.get_frame_f16(10, window).pixel(5, 5), 7)


def test_scaler_over_a_half_native_clip_pulled_as_f16(process, bt, orc):
    """VideoScaler over a half-native input, f16 pull: the scaler's own passes widen and truncate; values equal the
    f32 pull truncated."""
    clip = process.VideoGainOffsetFilter(process.SolidColorVideoSource(process.LerpFunc((0.1, 0.2, 0.3, 1.0), (0.9, 0.8, 0.7, 0.5), 10),
                                                                        bt.box2i(2, 1, 40, 30)), gain=1.25, offset=0.03125)
    scaler = process.VideoScaler(clip, target_point=(0, 0), source_point=(0, 0), scale_factors=(2.0, 1.5), source_rect=bt.box2i(0, 0, 63, 35))
    window = bt.box2i(0, 0, 99, 59)
    for frame in (0, 4):
        got16, got32 = scaler.get_frame_f16(frame, window), scaler.get_frame_f32(frame, window)
        assert got16.current_window == got32.current_window and not got16.current_window.empty()
        cw = got16.current_window
        for x, y in [(cw.min.x, cw.min.y), (cw.max.x, cw.max.y), ((cw.min.x + cw.max.x) // 2, (cw.min.y + cw.max.y) // 2), (cw.min.x + 1, cw.max.y - 1)]:
            want = orc.float_to_half(np.array(got32.pixel(x, y), np.float32))
            assert np.array_equal(orc.float_to_half(np.array(got16.pixel(x, y), np.float32)), want), (frame, x, y)


def test_random_graphs_f16_pull_equals_truncated_f32_pull(process, bt, orc):
    """Differential fuzz of the fused f16 paths (crossfade, workspace stack, gain) against the node-by-node f32 path:
    for 60 random graphs of full-window nodes, an f16 pull must equal the f32 pull truncated -- that is what
    video_get_frame_f16 does to an f32 node's output (main.c:43-71), whichever route the pixels took."""
    rng = random.Random(20261010)
    window = bt.box2i(0, 0, 47, 26)

    def colour():
        return (rng.random(), rng.random(), rng.random(), rng.choice([1.0, 1.0, rng.random(), 0.0]))

    def leaf():
        solid = process.SolidColorVideoSource(process.LerpFunc(colour(), colour(), 10))
        return process.VideoGainOffsetFilter(solid, gain=rng.choice([1.0, 0.5, 1.5]), offset=rng.choice([0.0, 0.0625])) if rng.random() < 0.7 else solid

    def node(depth):
        if depth == 0 or rng.random() < 0.25:
            return leaf()
        kind = rng.choice(["gain", "mix", "workspace", "pass", "sequence"])
        if kind == "gain":
            return process.VideoGainOffsetFilter(node(depth - 1), gain=rng.uniform(0.2, 2.0), offset=rng.uniform(-0.1, 0.1))
        if kind == "mix":
            return process.VideoMixFilter(node(depth - 1), node(depth - 1), rng.choice([0.0, 1.0, rng.random(), process.LerpFunc((0,), (1,), 10)]))
        if kind == "workspace":
            ws = process.VideoWorkspace()
            for z in range(rng.randint(1, 4)):
                ws.add(source=node(depth - 1), x=0, length=20, z=rng.randint(-3, 3), offset=rng.randint(0, 3))
            return ws
        if kind == "pass":
            return process.VideoPassThroughFilter(node(depth - 1), offset=rng.randint(-2, 2))
        seq = process.VideoSequence()
        for _ in range(rng.randint(1, 3)):
            seq.append((node(depth - 1), rng.randint(0, 3), rng.randint(3, 8)))
        return seq

    for case in range(60):
        graph = node(3)
        for frame in (0, 4):
            f16, f32 = graph.get_frame_f16(frame, window), graph.get_frame_f32(frame, window)
            assert f16.current_window == f32.current_window, (case, frame)
            if f16.current_window.empty():
                continue
            cw = f16.current_window
            for _ in range(25):
                x, y = rng.randint(cw.min.x, cw.max.x), rng.randint(cw.min.y, cw.max.y)
                want = orc.float_to_half(np.array(f32.pixel(x, y), np.float32))
                got = orc.float_to_half(np.array(f16.pixel(x, y), np.float32))
                # fold zero signs and NaNs the way tests/util.py does
                same = np.array_equal(got, want) or all((g == w) or ((g & 0x7FFF) == 0 and (w & 0x7FFF) == 0) or ((g & 0x7FFF) > 0x7C00 and (w & 0x7FFF) > 0x7C00) for g, w in zip(got, want))
                assert same, (case, frame, x, y, got, want)


def test_pull_queue_over_device_contexts(process, bt):
    """VideoPullQueue(workers=N, devices=[...]): one device context per entry (two entries naming device 0 = two contexts on
    the one GPU of this box), worker w bound to context w % G, frame g rendered by context g % G -- and the frames are the ones
    a direct pull gives."""
    assert process.device_count() >= 1
    assert [process.frame_owner(g, 2) for g in range(5)] == [0, 1, 0, 1, 0]
    with pytest.raises(ValueError):
        process.VideoPullQueue(workers=1, devices=[0, 0])       # fewer workers than devices
    q = process.VideoPullQueue(workers=4, devices=[0, 0])
    assert q.devices == (0, 0) and len(set(q.contexts)) == 2 and min(q.contexts) >= 0
    assert process.VideoPullQueue().devices == ()
    red, green = ramp(process, 0), ramp(process, 1)
    graph = process.VideoMixFilter(src_a=red, src_b=process.VideoGainOffsetFilter(green, gain=0.5, offset=0.25), mix_b=0.25)
    window = bt.box2i(0, 0, 15, 8)
    want = {g: graph.get_frame_f16(g, window) for g in range(16)}
    done, got = threading.Event(), {}

    def callback(frame_index, frame, user_data):
        got[frame_index] = frame
        if len(got) == 16:
            done.set()

    items = [q.enqueue(source=graph, frame_index=g, window=window, callback=callback, user_data=None) for g in range(16)]
    assert [it.owner for it in items] == [g % 2 for g in range(16)]
    assert done.wait(60), "callbacks did not arrive"
    for g in range(16):
        assert got[g].current_window == want[g].current_window
        for y in (0, 4, 8):
            for x in (0, 7, 15):
                almost(got[g].pixel(x, y), want[g].pixel(x, y), 7)
    del items
