"""Pins for the CPU oracle: the reference's own known-answer tests, facts the survey measured
on the compiled reference, and independent first-principles models.

CPU only; the oracle is test infrastructure and is the thing under test here.
"""
import ctypes as C

import numpy as np
import pytest

from canvas_amd.abi import HostFrame, box2i, rgba_frame_f32, v2f
from tests.models import f2h_rz_model, h2f_ieee, over_model

ALL_CODES = np.arange(65536, dtype=np.uint16)


# ---------------------------------------------------------------- half.c / genhalf.py

def test_h2f_is_ieee_for_every_code(orc):
    got = orc.half_to_float(ALL_CODES)
    want = h2f_ieee(ALL_CODES)
    nan = np.isnan(want)
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))
    assert np.isnan(got[nan]).all()
    # the tables keep the payload and do NOT quiet signalling NaNs (half.c:31-37)
    bits = got.view(np.uint32)
    h = ALL_CODES.astype(np.uint32)
    expect_nan_bits = ((h & 0x8000) << 16) | 0x7F800000 | ((h & 0x3FF) << 13)
    assert np.array_equal(bits[nan], expect_nan_bits[nan])


def test_f2h_truncates_matches_integer_model(orc):
    rng = np.random.default_rng(7)
    probes = [
        rng.uniform(-4, 4, 200000).astype(np.float32),
        rng.uniform(-70000, 70000, 50000).astype(np.float32),
        (rng.uniform(-1, 1, 50000) * 2.0 ** rng.integers(-30, -10, 50000)).astype(np.float32),
        rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 65535.9, 65536.0, -65536.0, 1e30, -1e30, np.inf, -np.inf,
                  2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -24, 2.0 ** -14, np.nextafter(np.float32(2.0 ** -14), np.float32(0))],
                 np.float32),
        np.array([0x7F800001, 0x7F801FFF, 0x7F802000, 0x7FC00000, 0xFFC00000, 0xFF800001], np.uint32).view(np.float32),
    ]
    for p in probes:
        assert np.array_equal(orc.float_to_half(p), f2h_rz_model(p))


def test_f2h_vs_round_to_nearest_statistics(orc):
    # on 1e6 U(-4,4) samples the reference's f2h equals RNE in 50.0 % and is one
    # code lower (in magnitude) in 50.0 %.
    rng = np.random.default_rng(0)
    x = rng.uniform(-4, 4, 1_000_000).astype(np.float32)
    got = orc.float_to_half(x).astype(np.int64)
    rne = x.astype(np.float16).view(np.uint16).astype(np.int64)
    d = rne - got                       # same sign => magnitude codes compare directly
    assert set(np.unique(d)) <= {0, 1}
    assert abs((d == 0).mean() - 0.5) < 0.005
    back = orc.half_to_float(got.astype(np.uint16))
    assert (np.abs(back) <= np.abs(x)).all()


def test_fast_variants(orc):
    # half.c:39-45,53-59: exponent re-bias only; exact for normal halfs
    normal = ALL_CODES[((ALL_CODES >> 10) & 0x1F != 0) & ((ALL_CODES >> 10) & 0x1F != 31)]
    out = np.empty(normal.shape, np.float32)
    lib = orc.lib()
    lib.orc_half_to_float_fast(out.ctypes.data_as(C.POINTER(C.c_float)),
                               normal.ctypes.data_as(C.POINTER(C.c_uint16)), normal.size)
    assert np.array_equal(out, h2f_ieee(normal))
    back = np.empty(normal.shape, np.uint16)
    lib.orc_float_to_half_fast(back.ctypes.data_as(C.POINTER(C.c_uint16)),
                               out.ctypes.data_as(C.POINTER(C.c_float)), normal.size)
    assert np.array_equal(back, normal)


def test_half_lookup(orc):
    rng = np.random.default_rng(1)
    table = rng.integers(0, 65536, 65536).astype(np.uint16)
    codes = rng.integers(0, 65536, 1000).astype(np.uint16)
    assert np.array_equal(orc.half_lookup(table, codes), table[codes])


# ---------------------------------------------------------------- gammatab.c

def _rz(f):
    return f2h_rz_model(np.asarray(f, np.float32))


def test_transfer_tables_against_float64_model(orc):
    x = h2f_ieee(ALL_CODES).astype(np.float64)
    finite = np.isfinite(x)
    with np.errstate(all="ignore"):
        models = {
            0: np.where(x < np.float32(4.5) * np.float32(0.018), x / 4.5, ((x + np.float32(0.099)) / np.float32(1.099)) ** (1 / 0.45)),
            1: np.where(x < 0, 0.0, x ** 2.5),
            2: np.where(x < np.float32(0.018), x * 4.5, np.float32(1.099) * x ** 0.45 - np.float32(0.099)),
            3: np.where(x <= np.float32(0.0031308), x * np.float32(12.92), 1.055 * x ** (1 / 2.4) - 0.055),
        }
    for which, m in models.items():
        tab = orc.transfer_table(which)
        want = _rz(m.astype(np.float32))
        ok = finite & np.isfinite(m)
        # powf in f32 vs pow in f64: allow one half code either way, require >= 99 % exact
        d = np.abs(tab[ok].astype(np.int64) - want[ok].astype(np.int64))
        assert d.max() <= 1, (which, d.max())
        assert (d == 0).mean() > 0.99
    # exact anchors
    assert orc.transfer_table(0)[0x3C00] == 0x3C00        # rec709->linear(1.0) = 1.0
    assert orc.transfer_table(1)[0x8400] == 0             # display: negatives -> 0 (gammatab.c:145-147)
    assert orc.transfer_table(3)[0] == 0


def test_gamma45_ramp(orc):
    ramp = orc.gamma45_ramp()
    assert ramp[0] == 0 and ramp[0x3C00] == 255 and ramp[0x7C00] == 255     # 0, 1.0, +inf (clamped)
    assert ramp[0x3800] == int(0.5 ** 0.45 * 255)                            # 0.5
    pos = ramp[: 0x7C00 + 1].astype(int)
    assert (np.diff(pos) >= 0).all()


# ---------------------------------------------------------------- filter.c

def test_triangle_taps(orc):
    # factor 0.5 => 3 taps [.25,.5,.25] (measured on the compiled reference)
    taps, centre = orc.fir_triangle(0.5, 0.0)
    assert centre == 1 and np.array_equal(taps, np.array([0.25, 0.5, 0.25], np.float32))
    taps, centre = orc.fir_triangle(2.0, 0.0)           # upsample: not normalised, edge taps dropped
    assert centre == 1 and np.array_equal(taps, np.array([0.5, 1.0, 0.5], np.float32))
    taps, centre = orc.fir_triangle(0.25, 0.5)
    assert abs(taps.sum() - 1) < 1e-6 and len(taps) == 8 and centre == 3


def test_triangle_small_buffer_protocol(orc):
    from canvas_amd.abi import fir_filter
    buf = (C.c_float * 2)()
    f = fir_filter(C.cast(buf, C.POINTER(C.c_float)), 2, 0)
    orc.lib().orc_fir_triangle(C.c_float(0.25), C.c_float(0.0), C.byref(f))
    assert f.center == -1 and f.width == 7              # filter.c:51-55


def test_lanczos_taps(orc):
    # Lanczos3 at sub=0.5 => width 11, centre 5, sum 1 (measured)
    taps, centre = orc.fir_lanczos(0.5, 3, 0.0)
    assert len(taps) == 11 and centre == 5
    assert abs(float(taps.astype(np.float64).sum()) - 1.0) < 1e-6
    assert np.allclose(taps, taps[::-1], atol=1e-7) and taps[5] == taps.max()
    x = (np.arange(11) - 5) * 0.5
    with np.errstate(all="ignore"):
        model = np.where(x == 0, 1.0, 3 * np.sin(np.pi * x) * np.sin(np.pi * x / 3) / (np.pi ** 2 * x ** 2))
    model = model.astype(np.float32)
    assert np.allclose(taps, model / model.sum(dtype=np.float32), atol=2e-7)


# ---------------------------------------------------------------- reference KATs (tests/ in the reference)

def _px(r, g, b, a):
    return HostFrame((0, 0, 0, 0), np.float32, np.array([[[r, g, b, a]]], np.float32))


def test_crossfade_kat_sequence_py(orc):
    """tests/canvas/sequence.py:58-100 check1, frames 15..19: green (i-9) fades to blue (i-14)
    with mix (i-15)/5; expected g=(i-9)(1-(i-15)/5), b=(i-14)(i-15)/5, a=1, to 6 places."""
    for i in range(15, 20):
        mix = float(i - 15) / 5.0
        a, b, out = _px(0, float(i - 9), 0, 1), _px(0, 0, float(i - 14), 1), _px(9, 9, 9, 9)
        orc.lib().orc_mix_cross_f32(out.ref(), a.ref(), b.ref(), C.c_float(mix))
        r, g, bl, al = [float(v) for v in out.array[0, 0]]
        assert out.current_window.tuple() == (0, 0, 0, 0)
        assert round(r - 0.0, 6) == 0
        assert round(g - float(i - 9) * (1.0 - mix), 6) == 0
        assert round(bl - float(i - 14) * mix, 6) == 0
        assert round(al - 1.0, 6) == 0


def test_solid_kat_rgbaframef16_py(orc):
    """tests/process/video/RgbaFrameF16.py:6-23 and SolidColorVideoSource.py:21-33."""
    color = np.array([1.0, 0.5, 0.333333, 0.2], np.float32)
    frame = HostFrame((0, 0, 3, 3), np.uint16)
    win = box2i.of(0, 0, 2, 2)
    orc.lib().orc_solid_f16(frame.ref(), C.byref(win), color.ctypes.data_as(C.POINTER(C.c_float)))
    assert frame.current_window.tuple() == (0, 0, 2, 2)
    px = orc.half_to_float(frame.array[0, 0])
    assert np.allclose(px, color, atol=5e-4)            # assertAlmostEqual(..., 3)
    frame2 = HostFrame((-1, -1, 1, 1), np.uint16)
    orc.lib().orc_copy_frame_f16(frame2.ref(), frame.ref())
    assert frame2.current_window.tuple() == (0, 0, 1, 1)
    assert np.array_equal(frame2.window_view(), frame.array[0:2, 0:2])

    f32 = HostFrame((0, 0, 3, 3), np.float32)
    orc.lib().orc_solid_f32(f32.ref(), C.byref(win), color.ctypes.data_as(C.POINTER(C.c_float)))
    assert f32.current_window.tuple() == (0, 0, 2, 2)
    assert np.array_equal(f32.array[0, 0], color)       # 6 places in the reference; exact here


def test_solid_moving_window_kat(orc):
    """SolidColorVideoSource.py:46-55: windows (-2,-2,2,2) .. (-4,-4,0,6) clipped to (-5,-5,5,6)."""
    color = np.array([0, 0, 1, 1], np.float32)
    for win in [(-2, -2, 2, 2), (-3, -3, 1, 4), (-4, -4, 0, 6)]:
        f = HostFrame((-5, -5, 5, 6), np.float32)
        w = box2i.of(*win)
        orc.lib().orc_solid_f32(f.ref(), C.byref(w), color.ctypes.data_as(C.POINTER(C.c_float)))
        assert f.current_window.tuple() == win
        assert (f.window_view() == color).all()


# ---------------------------------------------------------------- over / copy vs the per-pixel model

def _rand_frame(rng, full, win, alpha="rand"):
    fw = box2i.of(*full)
    arr = rng.uniform(0, 1, (fw.height, fw.width, 4)).astype(np.float32)
    if alpha == "one":
        arr[..., 3] = 1
    elif alpha == "zero":
        arr[..., 3] = 0
    return HostFrame(full, np.float32, arr, win)


@pytest.mark.parametrize("lower_win,upper_win", [
    ((0, 0, 15, 8), (0, 0, 15, 8)),        # full overlap
    ((0, 0, 15, 8), (3, 2, 10, 6)),        # upper nested in lower
    ((3, 2, 10, 6), (0, 0, 15, 8)),        # lower nested in upper
    ((0, 0, 15, 8), (0, 0, -1, -1)),       # empty upper
    ((0, 0, -1, -1), (2, 1, 9, 7)),        # empty lower
    ((0, 0, 7, 8), (0, 0, 15, 8)),         # same origin, upper wider
    ((1, 1, 6, 3), (1, 5, 6, 8)),          # disjoint vertically, same x span
])
@pytest.mark.parametrize("mix", [1.0, 0.3, 0.0, 1.7])
def test_over_matches_model(orc, lower_win, upper_win, mix):
    rng = np.random.default_rng(hash((lower_win, upper_win)) & 0xFFFF)
    full = (0, 0, 15, 8)
    lower, upper = _rand_frame(rng, full, lower_win), _rand_frame(rng, full, upper_win)
    want, want_win = over_model(lower.array, lower_win, upper.array, upper_win, full, mix)
    orc.lib().orc_mix_over_f32(lower.ref(), upper.ref(), C.c_float(mix))
    assert lower.current_window.tuple() == tuple(want_win)
    if not lower.current_window.is_empty():
        x0, y0, x1, y1 = want_win
        assert np.array_equal(lower.array[y0:y1 + 1, x0:x1 + 1].view(np.uint32), want[y0:y1 + 1, x0:x1 + 1].view(np.uint32))


def test_over_zero_alpha_gives_zero_pixel(orc):
    rng = np.random.default_rng(3)
    full = (0, 0, 7, 3)
    lower, upper = _rand_frame(rng, full, full, "zero"), _rand_frame(rng, full, full, "zero")
    orc.lib().orc_mix_over_f32(lower.ref(), upper.ref(), C.c_float(1.0))
    assert (lower.array == 0).all()                       # video_mix.c:334-336


def test_workspace_stack_order_and_membership(orc):
    """workspace.c:243-307,494-550: items live while x <= i < x+length; lowest z at the bottom."""
    from canvas_amd.abi import GET_FRAME_F32, video_frame_source_funcs, video_source
    colours = {1: (1, 0, 0, 1), 2: (0, 1, 0, 0.5), 3: (0, 0, 1, 0.25)}
    seen = []

    def make(tag):
        def get(self, idx, fp):
            seen.append((tag, idx))
            f = fp.contents
            f.current_window = f.full_window
            n = f.full_window.width * f.full_window.height
            arr = np.ctypeslib.as_array(C.cast(f.data, C.POINTER(C.c_float)), shape=(n, 4))
            arr[:] = colours[tag]
        return GET_FRAME_F32(get)

    cbs = {t: make(t) for t in colours}
    funcs = {t: video_frame_source_funcs(0, C.cast(None, type(video_frame_source_funcs().get_frame)), cbs[t], None) for t in colours}
    srcs = {t: video_source(None, C.pointer(funcs[t])) for t in colours}
    items = (orc.ws_item * 3)(
        orc.ws_item(0, 10, 5, 100, C.pointer(srcs[2])),     # z=5, middle
        orc.ws_item(5, 10, 9, 0, C.pointer(srcs[3])),       # z=9, top, starts at 5
        orc.ws_item(-3, 20, -1, 7, C.pointer(srcs[1])),     # z=-1, bottom
    )
    out = HostFrame((0, 0, 1, 0), np.float32)
    orc.lib().orc_workspace_get_frame_f32(items, 3, 4, out.ref())
    assert seen == [(1, 4 + 3 + 7), (2, 4 + 100)]
    # green(0.5) over red(1): a = 1*(1-0.5)+0.5 = 1; rgb = (red*0.5 + green*0.5)/1
    assert np.allclose(out.array[0, 0], [0.5, 0.5, 0, 1])
    seen.clear()
    orc.lib().orc_workspace_get_frame_f32(items, 3, 12, out.ref())
    assert [t for t, _ in seen] == [1, 3]
    seen.clear()
    orc.lib().orc_workspace_get_frame_f32(items, 3, 40, out.ref())
    assert seen == [] and out.current_window.is_empty()


# ---------------------------------------------------------------- colour matrix structure

def test_color_matrix_structure(orc):
    """color.c:104-165: LUT hits all four channels; matrix in f32, left-to-right; alpha copied;
    every f32->f16 step truncates."""
    rng = np.random.default_rng(11)
    full = (0, 0, 9, 4)
    codes = orc.float_to_half(rng.uniform(0, 1, (5, 10, 4)).astype(np.float32))
    frame = HostFrame(full, np.uint16, codes, (1, 1, 8, 3))
    before = frame.array.copy()
    orc.lib().orc_color_rgb_to_xyz_sdtv(frame.ref())
    lut = orc.transfer_table(0)
    m = np.array([0.3936, 0.2124, 0.0187, 0.3652, 0.7010, 0.1119, 0.1916, 0.0865, 0.9582], np.float32)
    v = h2f_ieee(lut[before])
    f32 = np.float32
    out = np.empty_like(v)
    for c in range(3):
        out[..., c] = ((v[..., 0] * m[c]).astype(f32) + (v[..., 1] * m[3 + c]).astype(f32)).astype(f32)
        out[..., c] = (out[..., c] + (v[..., 2] * m[6 + c]).astype(f32)).astype(f32)
    out[..., 3] = v[..., 3]
    want = f2h_rz_model(out)
    assert np.array_equal(frame.window_view(), want[1:4, 1:9])
    # outside current_window untouched
    mask = np.ones((5, 10), bool)
    mask[1:4, 1:9] = False
    assert np.array_equal(frame.array[mask], before[mask])


def test_chain_equals_node_by_node(orc):
    rng = np.random.default_rng(5)
    full = (0, 0, 31, 17)
    layers = []
    for k in range(3):
        px = rng.uniform(0, 1, (18, 32, 4)).astype(np.float32)
        if k == 0:
            px[..., 3] = 1
        layers.append(HostFrame(full, np.uint16, orc.float_to_half(px)))
    m = np.array([0.2126, -0.114572, 0.5, 0.7152, -0.385428, -0.454153, 0.0722, 0.5, -0.045847], np.float32)
    lut = orc.transfer_table(0)
    got = orc.chain_color_over(layers, m, lut, None)
    # node by node with the public pieces
    graded = []
    for l in layers:
        g = l.copy()
        orc.lib().orc_color_matrix_f16(g.ref(), m.ctypes.data_as(C.POINTER(C.c_float)),
                                       lut.ctypes.data_as(C.POINTER(C.c_uint16)), None)
        graded.append(HostFrame(full, np.float32, orc.half_to_float(g.array)))
    acc = graded[0]
    for g in graded[1:]:
        orc.lib().orc_mix_over_f32(acc.ref(), g.ref(), C.c_float(1.0))
    assert np.array_equal(got.array, orc.float_to_half(acc.array))
    assert got.current_window.tuple() == full


# ---------------------------------------------------------------- second statements of the remaining operations
# The reference holds no test for these; the C oracle and the numpy models in tests/models.py were
# written separately from the same source lines, in different forms (row loops against whole-array expressions).

def _canon(a):
    from tests.util import canon_f16, canon_f32
    return canon_f32(a) if a.dtype == np.float32 else canon_f16(a)


def test_crossfade_matches_model(orc):
    from tests.models import cross_model
    rng = np.random.default_rng(301)
    full = (0, 0, 40, 21)
    for mix in (0.0, 0.3, 0.5, 1.0, 1.7):
        a = rng.uniform(-0.5, 1.5, (22, 41, 4)).astype(np.float32)
        b = rng.uniform(-0.5, 1.5, (22, 41, 4)).astype(np.float32)
        a[..., 3] = rng.choice([0.0, 1.0, 0.4], (22, 41)).astype(np.float32)
        b[..., 3] = rng.choice([0.0, 1.0, 0.7], (22, 41)).astype(np.float32)
        out = HostFrame(full, np.float32)
        orc.lib().orc_mix_cross_f32(out.ref(), HostFrame(full, np.float32, a).ref(), HostFrame(full, np.float32, b).ref(), C.c_float(mix))
        assert np.array_equal(_canon(out.array), _canon(cross_model(a, b, mix))), mix


def test_gain_offset_and_bytes_match_models(orc):
    from tests.models import bytes_model, gain_offset_model
    rng = np.random.default_rng(302)
    full = (0, 0, 63, 35)
    codes = rng.integers(0, 65536, (36, 64, 4), dtype=np.uint16)
    src, out = HostFrame(full, np.uint16, codes), HostFrame(full, np.uint16)
    orc.lib().orc_gain_offset_f16(out.ref(), src.ref(), C.c_float(1.75), C.c_float(-0.125))
    assert np.array_equal(_canon(out.array), _canon(gain_offset_model(codes, 1.75, -0.125)))
    ramp = orc.gamma45_ramp()
    for table, mode in [(None, 0), (orc.transfer_table(3), 0), (None, 1), (orc.transfer_table(3), 1)]:
        packed = np.zeros((36, 64), np.uint32)
        orc.lib().orc_frame_to_bytes(packed.ctypes.data_as(C.POINTER(C.c_uint32)), src.ref(),
                                     None if table is None else table.ctypes.data_as(C.POINTER(C.c_uint16)), mode)
        assert np.array_equal(packed, bytes_model(codes, ramp, table, mode == 1))
    # the software widget's ramp (rendering intent) and its frame conversion
    from tests.models import widget_ramp_model
    for intent in (1.25, 1.0, 0.8):
        got = np.zeros(65536, np.uint8)
        orc.lib().orc_widget_ramp(got.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_float(intent))
        want = widget_ramp_model(intent)
        # numpy's powf and libm's may differ in the last place: a byte may move only where x^intent*255 sits on a rounding tie
        diff = np.flatnonzero(got != want)
        assert diff.size <= 4 and np.all(np.abs(got[diff].astype(int) - want[diff].astype(int)) == 1), (intent, diff)
        packed = np.zeros((36, 64), np.uint32)
        orc.lib().orc_frame_to_rgba8_intent(packed.ctypes.data_as(C.POINTER(C.c_uint32)), src.ref(),
                                            orc.transfer_table(3).ctypes.data_as(C.POINTER(C.c_uint16)), C.c_float(intent))
        assert np.array_equal(packed, bytes_model(codes, got, orc.transfer_table(3), False))
    ramp125 = np.zeros(65536, np.uint8)
    orc.lib().orc_widget_ramp(ramp125.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_float(1.25))
    # known answers: 0 -> 0, 1.0 -> 255, 0.5 -> lrint(0.5^1.25 * 255) = 107, negatives and NaN -> 0, +inf -> 255
    assert ramp125[0x0000] == 0 and ramp125[0x3C00] == 255 and ramp125[0x3800] == 107
    assert ramp125[0xBC00] == 0 and ramp125[0x7E00] == 0 and ramp125[0x7C00] == 255


def test_field_weave_matches_model(orc):
    from tests.models import weave_model
    rng = np.random.default_rng(305)
    for full, cur, ocur in [((0, 0, 31, 17), (0, 0, 31, 17), (0, 0, 31, 17)), ((0, -1, 31, 16), (0, -1, 31, 16), (0, -1, 31, 16)),
                            ((-4, -3, 40, 20), (3, 2, 29, 14), (3, 2, 29, 14)), ((-8, -3, 40, 20), (-5, 1, 20, 9), (-5, 1, 20, 9)),
                            ((0, 0, 31, 17), (0, 0, 31, 17), (4, 3, 20, 9)), ((0, 0, 31, 17), (0, 0, 31, 17), (0, 0, -1, -1))]:
        fa = rng.integers(0, 65536, (full[3] - full[1] + 1, full[2] - full[0] + 1, 4), dtype=np.uint16)
        oa = rng.integers(0, 65536, (cur[3] - cur[1] + 1, cur[2] - cur[0] + 1, 4), dtype=np.uint16)
        frame, other = HostFrame(full, np.uint16, fa, cur), HostFrame(cur, np.uint16, oa, ocur)
        before = fa.copy()
        want = weave_model(fa, full, cur, oa, ocur)
        assert not np.array_equal(want, before)                  # the model does something
        orc.lib().orc_weave_fields_f16(frame.ref(), other.ref())
        assert np.array_equal(frame.array, want), (full, cur, ocur)


def test_blur_and_lanczos_match_models(orc):
    from tests.models import blur_model, lanczos_model
    rng = np.random.default_rng(303)
    src = rng.uniform(-0.5, 1.5, (23, 37, 4)).astype(np.float32)
    frame = HostFrame((0, 0, 36, 22), np.float32, src)
    for ntaps in (9, 4, 1):
        taps = rng.uniform(-0.2, 1.0, ntaps).astype(np.float32)
        out = HostFrame((0, 0, 36, 22), np.float32)
        orc.lib().orc_fir_blur_f32(out.ref(), frame.ref(), taps.ctypes.data_as(C.POINTER(C.c_float)), ntaps)
        assert np.array_equal(_canon(out.array), _canon(blur_model(src, taps))), ntaps
    for fx, fy, tsize in [(0.5, 0.5, (19, 12)), (2.0, 1.5, (60, 30)), (0.4, 1.0, (15, 23))]:
        out = HostFrame((0, 0, tsize[0] - 1, tsize[1] - 1), np.float32)
        orc.lib().orc_resample_lanczos_f32(out.ref(), frame.ref(), C.c_float(fx), C.c_float(fy), 3)
        want = lanczos_model(src, tsize, fx, fy, 3, orc.fir_lanczos)
        assert np.array_equal(_canon(out.array), _canon(want)), (fx, fy)


def test_dv_edge_matches_models(orc):
    from tests.models import dv_reconstruct_model, dv_subsample_model
    rng = np.random.default_rng(304)
    planes = [np.ascontiguousarray(rng.integers(0, 256, (480, s), dtype=np.uint8)) for s in (720, 180, 180)]
    frame = HostFrame((0, -1, 719, 478), np.uint16)
    orc.lib().orc_reconstruct_dv(frame.ref(), (C.c_void_p * 3)(*[p.ctypes.data for p in planes]), (C.c_int * 3)(720, 180, 180))
    want = dv_reconstruct_model(planes[0], planes[1], planes[2], orc.transfer_table(0), orc.fir_triangle(4.0, 0.0))
    assert np.array_equal(_canon(frame.array), _canon(want))
    codes = rng.integers(0, 0x7C00, (480, 720, 4), dtype=np.uint16)          # finite, non-negative halfs
    src = HostFrame((0, -1, 719, 478), np.uint16, codes.copy())
    back = [np.zeros((480, s), np.uint8) for s in (720, 180, 180)]
    orc.lib().orc_subsample_dv((C.c_void_p * 3)(*[p.ctypes.data for p in back]), (C.c_int * 3)(720, 180, 180), src.ref())
    yy, cb, cr = dv_subsample_model(codes, orc.transfer_table(2), orc.fir_triangle(0.25, 0.0))
    assert np.array_equal(back[0], yy) and np.array_equal(back[1], cb) and np.array_equal(back[2], cr)
    assert np.array_equal(src.array, orc.transfer_table(2)[codes])          # the input was encoded in place


def test_triangle_scaler_matches_model(orc):
    """video_scale_bilinear_f32 against the whole-array statement: 60 random set-ups plus the fixed cases of the GPU
    parity test (pass order, partial-coverage intermediate window, scatter / gather, window reported)."""
    from canvas_amd.abi import v2f
    from tests.models import scale_model
    rng = np.random.default_rng(305)
    cases = [((0, 0, 31, 17), (0, 0, 15, 8), (0, 0, 15, 8), (0, 0), (0, 0), (2.0, 2.0)),
             ((0, 0, 15, 8), (0, 0, 31, 17), (0, 0, 31, 17), (0, 0), (0, 0), (0.5, 0.5)),
             ((0, 0, 40, 30), (0, 0, 15, 8), (0, 0, 15, 8), (3.5, 2.25), (1.0, 0.5), (2.5, 3.0)),
             ((0, 0, 20, 40), (0, 0, 15, 8), (0, 0, 15, 8), (0, 0), (0, 0), (1.3, 4.0)),
             ((0, 0, 15, 8), (0, 0, 15, 8), (0, 0, 15, 8), (2.0, 0.0), (0, 0), (1.0, 1.0)),
             ((-8, -4, 23, 13), (0, 0, 15, 8), (0, 0, 15, 8), (0, 0), (8.0, 4.0), (2.0, 2.0))]
    for _ in range(60):
        sfull = (int(rng.integers(-4, 3)), int(rng.integers(-3, 3)), int(rng.integers(8, 22)), int(rng.integers(5, 14)))
        tfull = (int(rng.integers(-4, 3)), int(rng.integers(-3, 3)), int(rng.integers(8, 30)), int(rng.integers(5, 22)))
        ax, bx = sorted(int(v) for v in rng.integers(sfull[0], sfull[2] + 1, 2))
        ay, by = sorted(int(v) for v in rng.integers(sfull[1], sfull[3] + 1, 2))
        fac = (float(rng.choice([0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0])), float(rng.choice([0.25, 0.5, 0.8, 1.0, 1.25, 2.0, 4.0])))
        tp = (float(rng.choice([0.0, 0.5, 2.25])), float(rng.choice([0.0, 1.0, 3.5])))
        sp = (float(rng.choice([0.0, 0.75, 2.0])), float(rng.choice([0.0, 0.5, 1.0])))
        cases.append((tfull, sfull, (ax, ay, bx, by), tp, sp, fac))
    for tfull, sfull, scur, tp, sp, fac in cases:
        h, w = sfull[3] - sfull[1] + 1, sfull[2] - sfull[0] + 1
        src = rng.uniform(-0.5, 1.5, (h, w, 4)).astype(np.float32)
        frame = HostFrame(sfull, np.float32, src, scur)
        out = HostFrame(tfull, np.float32)
        orc.lib().orc_scale_bilinear_f32(out.ref(), v2f(*tp), frame.ref(), v2f(*sp), v2f(*fac))
        want, win = scale_model(src, sfull, scur, tfull, tp, sp, fac, orc.fir_triangle)
        got_win = out.current_window.tuple()
        both_empty = (got_win[2] < got_win[0] or got_win[3] < got_win[1]) and (win[2] < win[0] or win[3] < win[1])
        assert both_empty or got_win == tuple(win), (tfull, sfull, scur, tp, sp, fac, got_win, win)
        if not both_empty:
            x0, y0, x1, y1 = got_win
            a = out.array[y0 - tfull[1]: y1 - tfull[1] + 1, x0 - tfull[0]: x1 - tfull[0] + 1]
            b = want[y0 - tfull[1]: y1 - tfull[1] + 1, x0 - tfull[0]: x1 - tfull[0] + 1]
            assert np.array_equal(_canon(a), _canon(b)), (tfull, sfull, scur, tp, sp, fac)
