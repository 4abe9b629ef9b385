"""GPU parity: every entry point of libcanvas_hip.so, called through the C-ABI, against the CPU
oracle on the same seeded inputs.  Bar: bit-exact (sign of zero and NaN payload aside, see
tests/util.py) -- tighter than the 1-ulp-in-half bar BASELINE.md allows.

Run on the GPU box with `pytest -m gpu`.  Nothing here reads /root/reference.
"""
import ctypes as C
import os

import numpy as np
import pytest

from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth
from canvas_amd.abi import (GET_FRAME_F16, GET_FRAME_F32, HostFrame, box2i, fir_filter, v2f,
                            video_frame_source_funcs, video_source)
from canvas_amd.device import DeviceFrame, chain_color_over
from tests.util import (assert_same_f16, assert_same_f32, f32p, rand_f16_frame, rand_f32_frame, same_window, u16p)

pytestmark = pytest.mark.gpu


def test_the_arithmetic_flavour_of_this_run_is_in_force(cvs, orc, arithmetic):
    """Every test of this module runs twice (tests/conftest.py): the library in one arithmetic flavour against the checker's
    build of the same flavour.  This one holds the plumbing itself: the library reports the flavour the test was given, and the
    checker in use is the matching build (the two builds differ on this input)."""
    assert cvs.cvs_get_arithmetic() == (_lib.ARITH_CONTRACTED if arithmetic == "contracted" else _lib.ARITH_SEPARATE)
    layers = [synth.layer_frame(256, 144, k, 0) for k in range(2)]
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    mine = orc.chain_color_over(layers, m, orc.transfer_table(0), None).array
    with orc.flavour("fma" if arithmetic == "contracted" else "gcc"):
        same = orc.chain_color_over(layers, m, orc.transfer_table(0), None).array
    with orc.flavour("gcc" if arithmetic == "contracted" else "fma"):
        other = orc.chain_color_over(layers, m, orc.transfer_table(0), None).array
    assert np.array_equal(mine, same) and not np.array_equal(mine, other)

ALL_CODES = np.arange(65536, dtype=np.uint16)


@pytest.fixture(scope="module")
def cvs():
    lib = _lib.load()
    assert lib.cvs_init(0) == 0, _lib.last_error()
    lib.init_half()
    return lib


# ------------------------------------------------------------------ A1-A3 half.c

def _host_h2f(codes, name="half_convert_to_float"):
    out = np.empty(codes.shape, np.float32)
    _lib.half_pointer(name)(f32p(out), u16p(codes), codes.size)
    return out


def _host_f2h(values, name="half_convert_from_float"):
    out = np.empty(values.shape, np.uint16)
    _lib.half_pointer(name)(u16p(out), f32p(values), values.size)
    return out


def test_half_pointers_null_until_init():
    # half.c:87-91: the globals are plain pointers that init_half() fills in
    lib = _lib.load()
    lib.init_half()
    for name in _lib.HALF_POINTER_GLOBALS:
        assert _lib.half_pointer(name) is not None


def test_h2f_every_code(cvs, orc):
    got, want = _host_h2f(ALL_CODES), orc.half_to_float(ALL_CODES)
    assert_same_f32(got, want, "h2f")
    # values, not just canonical form: everything that is not a NaN must match bit for bit
    nn = ~np.isnan(want)
    assert np.array_equal(got[nn].view(np.uint32), want[nn].view(np.uint32))


def test_f2h_probe_set(cvs, orc):
    rng = np.random.default_rng(7)
    probes = [
        rng.uniform(-4, 4, 300001).astype(np.float32),                     # ragged length: vector body + tail
        rng.uniform(-70000, 70000, 50000).astype(np.float32),
        (rng.uniform(-1, 1, 50000) * 2.0 ** rng.integers(-30, -10, 50000)).astype(np.float32),
        orc.half_to_float(ALL_CODES[(ALL_CODES & 0x7C00) != 0x7C00]),      # every finite half is a fixed point
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 65535.9, 65536.0, -65536.0, 1e30, -1e30, np.inf, -np.inf,
                  2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -24, 2.0 ** -14, 2.0 ** -14 * (1 - 2.0 ** -11), 1e-45, -1e-45], np.float32),
    ]
    bits = rng.integers(0, 2 ** 32, 400000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    probes.append(bits[~np.isnan(bits)])                                   # every exponent class, both signs
    for i, p in enumerate(probes):
        got, want = _host_f2h(p), orc.float_to_half(p)
        assert np.array_equal(got, want), "probe set %d" % i


def test_f2h_nan_stays_nan_or_matches(cvs, orc):
    # NaN payload/sign is not pinned (x86 vs gfx950 default NaN); the table turns low-payload NaNs into Inf
    p = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0x7F802000, 0x7FFFFFFF], np.uint32).view(np.float32)
    got, want = _host_f2h(p), orc.float_to_half(p)
    for g, w in zip(got, want):
        assert g == w or (g & 0x7FFF) > 0x7C00


def test_round_trip_is_identity_on_all_finite_codes(cvs):
    finite = ALL_CODES[(ALL_CODES & 0x7C00) != 0x7C00]
    assert np.array_equal(_host_f2h(_host_h2f(finite)), finite)


def test_fast_variants(cvs, orc):
    normal = ALL_CODES[((ALL_CODES >> 10) & 0x1F != 0) & ((ALL_CODES >> 10) & 0x1F != 31)]
    f = _host_h2f(normal, "half_convert_to_float_fast")
    assert np.array_equal(f.view(np.uint32), orc.half_to_float(normal).view(np.uint32))
    assert np.array_equal(_host_f2h(f, "half_convert_from_float_fast"), normal)


@pytest.mark.parametrize("count", [1, 7, 8, 1000, 65536 * 3 + 5])
def test_half_lookup(cvs, orc, count):
    rng = np.random.default_rng(count)
    table = rng.integers(0, 65536, 65536).astype(np.uint16)
    codes = rng.integers(0, 65536, count).astype(np.uint16)
    out = np.empty_like(codes)
    _lib.half_pointer("half_lookup")(u16p(table), u16p(out), u16p(codes), count)
    assert np.array_equal(out, table[codes])


def test_flat_dev_entry_points_with_unaligned_buffers(cvs, orc):
    rng = np.random.default_rng(3)
    n = 4099
    codes = rng.integers(0, 0x7C00, n + 3).astype(np.uint16)
    d_in, d_out = cvs.cvs_malloc(2 * (n + 3)), cvs.cvs_malloc(4 * (n + 4))
    _lib.check(cvs.cvs_memcpy_h2d(d_in, codes.ctypes.data, codes.nbytes, None))
    _lib.check(cvs.cvs_half_to_float_dev(d_out + 4, d_in + 6, n, None))      # both 16-byte misaligned
    out = np.empty(n, np.float32)
    _lib.check(cvs.cvs_memcpy_d2h(out.ctypes.data, d_out + 4, 4 * n, None))
    assert np.array_equal(out.view(np.uint32), orc.half_to_float(codes[3:3 + n]).view(np.uint32))
    cvs.cvs_free(d_in), cvs.cvs_free(d_out)


# ------------------------------------------------------------------ A13 gammatab.c

@pytest.mark.parametrize("which", [0, 1, 2, 3])
def test_transfer_tables_equal_oracle(cvs, orc, which):
    tab = np.ctypeslib.as_array(cvs.cvs_lut_host(which), shape=(65536,)).copy()
    assert_same_f16(tab, orc.transfer_table(which), "table %d" % which)
    d = np.empty(65536, np.uint16)
    _lib.check(cvs.cvs_memcpy_d2h(d.ctypes.data, cvs.cvs_lut_device(which), 131072, None))
    assert np.array_equal(d, tab)


def test_transfer_functions_on_host_buffers(cvs, orc):
    rng = np.random.default_rng(9)
    codes = rng.integers(0, 0x7C00, 12345).astype(np.uint16)
    for which, name in enumerate(["video_transfer_rec709_to_linear_scene", "video_transfer_rec709_to_linear_display",
                                  "video_transfer_linear_to_rec709", "video_transfer_linear_to_sRGB"]):
        out = np.empty_like(codes)
        getattr(cvs, name)(u16p(out), u16p(codes), codes.size)
        assert np.array_equal(out, orc.transfer_table(which)[codes]), name


def test_gamma45_ramp(cvs, orc):
    ramp = np.ctypeslib.as_array(cvs.video_get_gamma45_ramp(), shape=(65536,))
    not_nan = (ALL_CODES & 0x7FFF) <= 0x7C00
    nonneg = not_nan & (ALL_CODES < 0x8000)               # powf of a negative is NaN; its uint8 cast is unpinned
    assert np.array_equal(ramp[nonneg], orc.gamma45_ramp()[nonneg])


# ------------------------------------------------------------------ A10 filter.c

def _taps(cvs, kind, *args):
    f = fir_filter(None, 0, 0)
    if kind == "tri":
        cvs.filter_createTriangle(C.c_float(args[0]), C.c_float(args[1]), C.byref(f))
    else:
        cvs.filter_createLanczos(C.c_float(args[0]), args[1], C.c_float(args[2]), C.byref(f))
    taps = np.ctypeslib.as_array(f.coeff, shape=(f.width,)).copy()
    centre = f.center
    cvs.filter_free(C.byref(f))
    return taps, centre


@pytest.mark.parametrize("sub", [0.25, 0.5, 0.75, 1.0, 2.0, 3.5, 4.0])
@pytest.mark.parametrize("offset", [0.0, 0.25, 0.5, 0.999])
def test_fir_taps(cvs, orc, sub, offset):
    for kind, mine, theirs in [("tri", (sub, offset), orc.fir_triangle(sub, offset)),
                               ("lan", (sub, 3, offset), orc.fir_lanczos(sub, 3, offset))]:
        taps, centre = _taps(cvs, kind, *mine)
        assert centre == theirs[1] and len(taps) == len(theirs[0])
        assert np.array_equal(taps.view(np.uint32), theirs[0].view(np.uint32)), (kind, sub, offset)


def test_fir_small_buffer_protocol(cvs):
    buf = (C.c_float * 2)(7.0, 7.0)
    f = fir_filter(C.cast(buf, C.POINTER(C.c_float)), 2, 0)
    cvs.filter_createTriangle(C.c_float(0.25), C.c_float(0.0), C.byref(f))
    assert f.center == -1 and f.width == 7 and buf[0] == 7.0


# ------------------------------------------------------------------ A5 copies

WINDOW_CASES = [
    ((0, 0, 15, 8), (0, 0, 15, 8), (0, 0, 15, 8)),      # out.full, in.full, in.current
    ((0, 0, 15, 8), (0, 0, 15, 8), (3, 2, 10, 6)),
    ((-1, -1, 1, 1), (0, 0, 3, 3), (0, 0, 2, 2)),       # the RgbaFrameF16.py re-window
    ((0, 0, 15, 8), (-4, -4, 20, 12), (-2, -3, 18, 11)),
    ((0, 0, 15, 8), (0, 0, 15, 8), (0, 0, -1, -1)),     # empty input
    ((0, 0, 15, 8), (20, 20, 30, 30), (21, 21, 29, 29)),  # disjoint
]


@pytest.mark.parametrize("out_full,in_full,in_cur", WINDOW_CASES)
def test_copy_frame_f16(cvs, orc, out_full, in_full, in_cur):
    rng = np.random.default_rng(1)
    src = rand_f16_frame(rng, in_full, in_cur)
    a, b = rand_f16_frame(rng, out_full), None
    b = a.copy()
    cvs.video_copy_frame_f16(a.ref(), src.ref())
    orc.lib().orc_copy_frame_f16(b.ref(), src.ref())
    assert same_window(a.current_window, b.current_window)
    assert np.array_equal(a.array, b.array)


@pytest.mark.parametrize("out_full,in_full,in_cur", WINDOW_CASES)
@pytest.mark.parametrize("alpha", [1.0, 0.4, 0.0, 1.5, -2.0])
def test_copy_frame_alpha_f32(cvs, orc, out_full, in_full, in_cur, alpha):
    rng = np.random.default_rng(2)
    src = rand_f32_frame(rng, in_full, in_cur)
    a = rand_f32_frame(rng, out_full)
    b = a.copy()
    cvs.video_copy_frame_alpha_f32(a.ref(), src.ref(), C.c_float(alpha))
    orc.lib().orc_copy_frame_alpha_f32(b.ref(), src.ref(), C.c_float(alpha))
    assert same_window(a.current_window, b.current_window)
    assert_same_f32(a.array, b.array, "copy_alpha")


@pytest.mark.parametrize("alpha", [1.0, 0.4, 0.0, 1.5, -2.0])
def test_attenuate_f32_is_the_in_place_copy(cvs, orc, alpha):
    """framework.h:236 (no definition in the reference): defined as video_copy_frame_alpha_f32 with out == in."""
    rng = np.random.default_rng(12)
    a = rand_f32_frame(rng, (-2, -1, 20, 9), (1, 0, 15, 7))
    b = a.copy()
    cvs.video_attenuate_f32(a.ref(), C.c_float(alpha))
    orc.lib().orc_copy_frame_alpha_f32(b.ref(), b.ref(), C.c_float(alpha))
    assert same_window(a.current_window, b.current_window)
    assert_same_f32(a.array, b.array, "attenuate")


@pytest.mark.parametrize("full,cur,ocur", [
    ((0, 0, 31, 17), (0, 0, 31, 17), (0, 0, 31, 17)),          # whole frame
    ((0, -1, 31, 16), (0, -1, 31, 16), (0, -1, 31, 16)),       # DV raster: first line at y = -1, first even row is 0
    ((-4, -3, 40, 20), (0, 1, 29, 14), (0, 1, 29, 14)),        # window inside the buffer, odd first row
    ((-4, -3, 40, 20), (3, 2, 29, 14), (3, 2, 29, 14)),        # min.x > 0: the reference's row address starts 3 pixels early
    ((-8, -3, 40, 20), (-5, 2, 20, 9), (-5, 2, 20, 9)),        # min.x < 0: it starts 5 pixels late and runs into the next row
    ((0, 0, 31, 17), (0, 0, 31, 17), (4, 3, 20, 9)),           # the second field defined on part of the window only
    ((0, 0, 31, 17), (0, 0, 31, 17), (0, 0, -1, -1)),          # the second field empty
    ((0, 0, 9, 9), (2, 5, 7, 5), (2, 5, 7, 5)),                # one odd row: nothing to weave
])
def test_weave_fields(cvs, orc, full, cur, ocur):
    """Pulldown23RemovalFilter.c:88-104 on device frames against the oracle's restatement (x = 0 addressing included)."""
    rng = np.random.default_rng(77)
    frame = rand_f16_frame(rng, full, cur)
    other = rand_f16_frame(rng, cur, ocur)
    want = frame.copy()
    orc.lib().orc_weave_fields_f16(want.ref(), other.ref())
    d_frame, d_other = DeviceFrame.from_host(frame), DeviceFrame.from_host(other)
    _lib.check(cvs.cvs_weave_fields_f16_dev(d_frame.ref(), d_other.ref(), None))
    got = d_frame.download()
    assert same_window(got.current_window, want.current_window)
    assert np.array_equal(got.array, want.array)                # whole buffer: odd rows and everything outside stay untouched
    wrong = DeviceFrame.from_host(rand_f16_frame(rng, full, cur))
    if full != cur:
        assert cvs.cvs_weave_fields_f16_dev(d_frame.ref(), wrong.ref(), None) != 0      # not allocated for the current window


# ------------------------------------------------------------------ A6 / A7 mixers

FULL = (0, 0, 23, 11)
MIX_WINDOWS = [
    (FULL, FULL),                               # full overlap
    (FULL, (3, 2, 10, 6)),                      # nested
    ((3, 2, 10, 6), FULL),
    (FULL, (0, 0, -1, -1)),                     # empty either
    ((0, 0, -1, -1), (2, 1, 9, 7)),
    ((0, 0, 11, 11), (0, 0, 23, 11)),           # shared origin
    ((1, 1, 6, 3), (1, 5, 6, 8)),               # disjoint in y, same x span
    ((1, 1, 6, 3), (9, 1, 14, 3)),              # disjoint in x
    ((1, 1, 6, 3), (9, 6, 14, 9)),              # disjoint in both
    ((2, 1, 12, 7), (6, 4, 20, 10)),            # partial overlap, p upper-left
    ((6, 4, 20, 10), (2, 1, 12, 7)),            # partial overlap, p lower-right (exercises the `left` selector quirk)
    ((0, 5, 12, 9), (5, 0, 20, 7)),             # min.x vs min.y comparison picks the other frame
    ((4, 0, 9, 11), (0, 3, 23, 8)),             # cross shape
]


@pytest.mark.parametrize("pw,qw", MIX_WINDOWS)
@pytest.mark.parametrize("mix", [1.0, 0.35, 0.0, 2.0])
def test_mix_over(cvs, orc, pw, qw, mix):
    rng = np.random.default_rng(abs(hash((pw, qw))) % 9973)
    out = rand_f32_frame(rng, FULL, pw, "mixed")
    upper = rand_f32_frame(rng, FULL, qw, "mixed")
    want = out.copy()
    cvs.video_mix_over_f32(out.ref(), upper.ref(), C.c_float(mix))
    orc.lib().orc_mix_over_f32(want.ref(), upper.ref(), C.c_float(mix))
    assert same_window(out.current_window, want.current_window)
    assert_same_f32(out.array, want.array, "over")      # the WHOLE buffer: untouched junk must stay untouched


@pytest.mark.parametrize("pw,qw", MIX_WINDOWS)
@pytest.mark.parametrize("mix", [0.5, 0.2, 0.0, 1.0])
@pytest.mark.parametrize("in_place", [False, True])
def test_mix_cross(cvs, orc, pw, qw, mix, in_place):
    rng = np.random.default_rng(abs(hash((pw, qw, in_place))) % 9973)
    a = rand_f32_frame(rng, FULL, pw, "mixed")
    b = rand_f32_frame(rng, FULL, qw, "mixed")
    if in_place:
        want = a.copy()
        cvs.video_mix_cross_f32(a.ref(), a.ref(), b.ref(), C.c_float(mix))
        orc.lib().orc_mix_cross_f32(want.ref(), want.ref(), b.ref(), C.c_float(mix))
        got = a
    else:
        got = rand_f32_frame(rng, FULL)
        want = got.copy()
        cvs.video_mix_cross_f32(got.ref(), a.ref(), b.ref(), C.c_float(mix))
        orc.lib().orc_mix_cross_f32(want.ref(), a.ref(), b.ref(), C.c_float(mix))
    assert same_window(got.current_window, want.current_window)
    assert_same_f32(got.array, want.array, "cross")


def test_mix_over_clipped_by_smaller_output(cvs, orc):
    rng = np.random.default_rng(77)
    out = rand_f32_frame(rng, (4, 2, 19, 9), (5, 3, 15, 8), "mixed")
    upper = rand_f32_frame(rng, (4, 2, 19, 9), (8, 2, 19, 6), "mixed")
    want = out.copy()
    cvs.video_mix_over_f32(out.ref(), upper.ref(), C.c_float(0.8))
    orc.lib().orc_mix_over_f32(want.ref(), upper.ref(), C.c_float(0.8))
    assert same_window(out.current_window, want.current_window)
    assert_same_f32(out.array, want.array, "over (offset origin)")


def test_crossfade_kat_through_the_library(cvs):
    """The reference's own mixing KAT (tests/canvas/sequence.py:58-100, frames 15-19)."""
    for i in range(15, 20):
        mix = float(i - 15) / 5.0
        a = HostFrame((0, 0, 0, 0), np.float32, np.array([[[0, i - 9, 0, 1]]], np.float32))
        b = HostFrame((0, 0, 0, 0), np.float32, np.array([[[0, 0, i - 14, 1]]], np.float32))
        out = HostFrame((0, 0, 0, 0), np.float32)
        cvs.video_mix_cross_f32(out.ref(), a.ref(), b.ref(), C.c_float(mix))
        r, g, bl, al = [float(v) for v in out.array[0, 0]]
        assert round(r, 6) == 0 and round(al - 1.0, 6) == 0
        assert round(g - (i - 9) * (1.0 - mix), 6) == 0 and round(bl - (i - 14) * mix, 6) == 0


# ------------------------------------------------------------------ A12 colour matrix

@pytest.mark.parametrize("cur", [(0, 0, 40, 20), (3, 2, 30, 17), (5, 5, 5, 5), (0, 0, -1, -1)])
def test_named_colour_functions(cvs, orc, cur):
    rng = np.random.default_rng(5)
    for mine, theirs in [(cvs.video_color_rgb_to_xyz_sdtv, orc.lib().orc_color_rgb_to_xyz_sdtv),
                         (cvs.video_color_xyz_to_srgb, orc.lib().orc_color_xyz_to_srgb)]:
        a = rand_f16_frame(rng, (0, 0, 40, 20), cur)
        b = a.copy()
        mine(a.ref())
        theirs(b.ref())
        assert_same_f16(a.array, b.array, "colour")


@pytest.mark.parametrize("pre,post", [(-1, -1), (0, -1), (-1, 3), (0, 2)])
def test_colour_matrix_general(cvs, orc, pre, post):
    rng = np.random.default_rng(6)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    h = rand_f16_frame(rng, (-3, -2, 60, 33), (0, 0, 57, 30))
    # a few wild codes: negatives, subnormals, large values
    h.array[5, 5] = [0x8400, 0x0001, 0x7BFF, 0xC000]
    want = h.copy()
    dev = DeviceFrame.from_host(h)
    _lib.check(cvs.cvs_color_matrix_f16_dev(dev.ref(), f32p(m), pre, post, None))
    _lib.check(cvs.cvs_stream_sync(None))
    got = dev.download()
    orc.lib().orc_color_matrix_f16(want.ref(), f32p(m),
                                   None if pre < 0 else u16p(orc.transfer_table(pre)),
                                   None if post < 0 else u16p(orc.transfer_table(post)))
    assert_same_f16(got.array, want.array, "colour matrix pre=%d post=%d" % (pre, post))


# ------------------------------------------------------------------ A15 / A16

def test_gain_offset(cvs, orc):
    rng = np.random.default_rng(8)
    for out_full, in_full, in_cur in WINDOW_CASES:
        src = rand_f16_frame(rng, in_full, in_cur)
        a = rand_f16_frame(rng, out_full)
        b = a.copy()
        cvs.video_filter_gain_offset_f16(a.ref(), src.ref(), C.c_float(1.5), C.c_float(0.0625))
        orc.lib().orc_gain_offset_f16(b.ref(), src.ref(), C.c_float(1.5), C.c_float(0.0625))
        assert same_window(a.current_window, b.current_window)
        assert_same_f16(a.array, b.array, "gain/offset")


def test_solid_fill_and_reference_kats(cvs, orc):
    color = _lib.rgba_f32(1.0, 0.5, 0.333333, 0.2)
    carr = np.array([1.0, 0.5, 0.333333, 0.2], np.float32)
    # RgbaFrameF16.py:6-23
    frame = HostFrame((0, 0, 3, 3), np.uint16, fill=0x1234)
    want = frame.copy()
    win = box2i.of(0, 0, 2, 2)
    cvs.video_fill_solid_f16(frame.ref(), C.byref(win), C.byref(color))
    orc.lib().orc_solid_f16(want.ref(), C.byref(win), f32p(carr))
    assert frame.current_window.tuple() == (0, 0, 2, 2)
    assert np.array_equal(frame.array, want.array)
    assert np.allclose(orc.half_to_float(frame.array[0, 0]), carr, atol=5e-4)
    frame2 = HostFrame((-1, -1, 1, 1), np.uint16)
    cvs.video_copy_frame_f16(frame2.ref(), frame.ref())
    assert frame2.current_window.tuple() == (0, 0, 1, 1)
    # SolidColorVideoSource.py:46-55 (moving window)
    for w in [(-2, -2, 2, 2), (-3, -3, 1, 4), (-4, -4, 0, 6)]:
        f = HostFrame((-5, -5, 5, 6), np.float32)
        bw = box2i.of(*w)
        cvs.video_fill_solid_f32(f.ref(), C.byref(bw), C.byref(color))
        assert f.current_window.tuple() == w and (f.window_view() == carr).all()


# ------------------------------------------------------------------ chain (config 2 at small size)

def _synth_layers(w, h, n, frame=0):
    return [synth.layer_frame(w, h, k, frame) for k in range(n)]


@pytest.mark.parametrize("nlayers", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("size", [(64, 36), (33, 7)])       # even and odd pixel counts
def test_chain_fused_matches_oracle(cvs, orc, nlayers, size):
    w, h = size
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    layers = _synth_layers(w, h, nlayers)
    want = orc.chain_color_over(layers, m, orc.transfer_table(0), None)
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
    chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 1
    got = out.download()
    assert got.current_window.tuple() == (0, 0, w - 1, h - 1)
    assert_same_f16(got.array, want.array, "fused chain, %d layers" % nlayers)


def test_chain_jobs_that_feed_each_other_run_in_order(cvs, orc):
    """One call, jobs that depend on each other: job 1 stacks job 0's output, job 2 overwrites a buffer job 1 reads,
    job 3 overwrites job 0's output, job 4 works in place.  The library must carry them out as if one after the other
    (it starts a new launch at every dependent job): a batch in ONE launch gives no order between jobs."""
    w, h = 640, 360                                  # 115 200 pairs: 225 chunks, nearly every workgroup has one
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    lut = orc.transfer_table(0)
    L = _synth_layers(w, h, 4)
    full = (0, 0, w - 1, h - 1)
    d = [DeviceFrame.from_host(l) for l in L]
    a, b = DeviceFrame(full, np.uint16), DeviceFrame(full, np.uint16)
    jobs = [(a, [d[0], d[1]]),          # a = chain(L0, L1)
            (b, [a, d[2]]),             # b = chain(a, L2)           reads job 0's output
            (d[2], [d[0], d[3]]),       # L2 := chain(L0, L3)        overwrites what job 1 reads
            (a, [b, d[1]]),             # a = chain(b, L1)           overwrites job 0's output, reads job 1's
            (b, [b, d[3]])]             # b = chain(b, L3)           in place
    chain_color_over(jobs, m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 1
    wa = orc.chain_color_over([L[0], L[1]], m, lut, None)
    wb = orc.chain_color_over([wa, L[2]], m, lut, None)
    w2 = orc.chain_color_over([L[0], L[3]], m, lut, None)
    wa2 = orc.chain_color_over([wb, L[1]], m, lut, None)
    wb2 = orc.chain_color_over([wb, L[3]], m, lut, None)
    assert_same_f16(a.download().array, wa2.array, "a after job 3")
    assert_same_f16(b.download().array, wb2.array, "b after job 4")
    assert_same_f16(d[2].download().array, w2.array, "L2 after job 2")


def test_chain_shifted_overlap_of_output_and_layer_is_not_fused(cvs, orc):
    """out overlapping one of its own layers at a shifted address cannot go through the fused kernel (a lane would
    overwrite pixels another lane still has to read); it takes the node-by-node path, which copies the layer first."""
    w, h = 64, 36
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    L = _synth_layers(w, h, 2)
    arena = cvs.cvs_malloc(w * h * 8 + w * 8 * 4)
    full = (0, 0, w - 1, h - 1)
    lay0 = DeviceFrame(full, np.uint16, ptr=arena + w * 8 * 4)          # four rows into the arena
    out = DeviceFrame(full, np.uint16, ptr=arena)                        # overlaps lay0, shifted by four rows
    lay0.upload(L[0].array)
    lay1 = DeviceFrame.from_host(L[1])
    want = orc.chain_color_over(L, m, orc.transfer_table(0), None)
    chain_color_over([(out, [lay0, lay1])], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 0
    assert_same_f16(out.download().array, want.array, "shifted overlap")
    cvs.cvs_free(arena)


@pytest.mark.parametrize("nlayers", [2, 3, 4, 6, 8])
def test_chain_fused_with_live_divides(cvs, orc, nlayers):
    """Every layer translucent (alpha 0, 1 and in between, per pixel): the x/1.0 shortcut of the kernel
    must not be taken where the blended alpha is not exactly 1, and alpha 0 must give the zero pixel."""
    rng = np.random.default_rng(100 + nlayers)
    full = (0, 0, 95, 53)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    from tests.util import rand_f32_frame
    from canvas_amd.synth import truncate_to_half
    layers = []
    for k in range(nlayers):
        f = rand_f32_frame(rng, full, full, alpha="mixed", lo=-0.25, hi=1.5)   # out-of-gamut values too
        layers.append(HostFrame(full, np.uint16, truncate_to_half(f.array)))
    layers[1].array[3, 5] = [0x7BFF, 0x7BFF, 0xFBFF, 0x3C00]                  # 65504: pushes sums past the half range
    layers[0].array[3, 5] = [0x7BFF, 0x7BFF, 0xFBFF, 0x3800]
    want = orc.chain_color_over(layers, m, orc.transfer_table(0), None)
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame(full, np.uint16)
    chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 1
    assert_same_f16(out.download().array, want.array, "fused chain, translucent layers")


def test_chain_batch_and_lut_variants(cvs, orc):
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    for pre, post in [(-1, -1), (-1, 2), (0, 3)]:
        jobs, wants = [], []
        for frame in range(3):
            layers = _synth_layers(48, 20, 2, frame)
            wants.append(orc.chain_color_over(layers, m, None if pre < 0 else orc.transfer_table(pre),
                                              None if post < 0 else orc.transfer_table(post)))
            jobs.append((DeviceFrame((0, 0, 47, 19), np.uint16), [DeviceFrame.from_host(l) for l in layers]))
        chain_color_over(jobs, m, pre, post)
        _lib.check(cvs.cvs_stream_sync(None))
        for (out, _), want in zip(jobs, wants):
            assert_same_f16(out.download().array, want.array, "batch pre=%d post=%d" % (pre, post))


@pytest.mark.parametrize("nlayers,plain", [(1, False), (2, False), (3, True), (4, False), (5, True), (6, False), (7, False), (8, True)])
def test_chain_batch_of_frames_of_different_sizes(cvs, orc, nlayers, plain):
    """One call, frames of very different sizes, odd and even pixel counts, some smaller than one 8 KiB chunk: the
    workgroups walk the whole batch as one run of chunks, crossing from frame to frame with partial last chunks, and the
    last pixel of the odd-sized frames goes through the tail kernel.  70 jobs: more than one launch's job records."""
    m = None if plain else np.array(REC709_RGB_TO_YPBPR, np.float32)
    lut = None if plain else orc.transfer_table(0)
    sizes = [(64, 36), (33, 7), (640, 360), (1, 2), (513, 1), (2, 1), (1023, 3), (96, 54), (3, 1), (1280, 719)]
    jobs, wants = [], []
    for i in range(70):
        w, h = sizes[i % len(sizes)]
        layers = _synth_layers(w, h, nlayers, i)
        wants.append(orc.chain_color_over(layers, m, lut, None))
        jobs.append((DeviceFrame((0, 0, w - 1, h - 1), np.uint16), [DeviceFrame.from_host(l) for l in layers]))
    chain_color_over(jobs, m, _lib.LUT_NONE if plain else _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 1
    for i, ((out, _), want) in enumerate(zip(jobs, wants)):
        assert_same_f16(out.download().array, want.array, "job %d, %r" % (i, sizes[i % len(sizes)]))


def test_chain_ragged_windows_take_the_node_by_node_path(cvs, orc):
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    rng = np.random.default_rng(12)
    full = (0, 0, 47, 19)
    # Windows chosen so that video_mix.c:265's `left` selector (min.x compared with the other frame's
    # min.Y) picks the geometrically left frame: otherwise the reference copies pixels from outside
    # the upper layer's current_window, i.e. whatever its malloc'd temp happened to hold -- not
    # reproducible by anyone.  (test_mix_over covers those configurations with shared buffers.)
    layers = [rand_f16_frame(rng, full, full, alpha="one"), rand_f16_frame(rng, full, (5, 3, 30, 15)),
              rand_f16_frame(rng, full, (20, 2, 47, 10))]
    want = orc.chain_color_over(layers, m, orc.transfer_table(0), None)
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame(full, np.uint16)
    chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    assert cvs.cvs_chain_last_was_fused() == 0
    got = out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.window_view(), want.window_view(), "unfused chain")


# ------------------------------------------------------------------ A4 / A8 pulls and the workspace

def _py_source(fill16=None, fill32=None):
    """A video_source whose vtable is implemented in Python; returns (source, keepalive)."""
    keep = []

    def g16(self, idx, fp):
        fill16(idx, fp.contents)

    def g32(self, idx, fp):
        fill32(idx, fp.contents)

    cb16 = GET_FRAME_F16(g16) if fill16 else C.cast(None, GET_FRAME_F16)
    cb32 = GET_FRAME_F32(g32) if fill32 else C.cast(None, GET_FRAME_F32)
    funcs = video_frame_source_funcs(0, cb16, cb32, None)
    src = video_source(None, C.pointer(funcs))
    keep += [cb16, cb32, funcs]
    return src, keep


def _fill_from(host_frames):
    def fill(idx, f):
        src = host_frames[idx % len(host_frames)]
        n = f.full_window.width * f.full_window.height
        if src.dtype == np.uint16:
            dst = np.ctypeslib.as_array(C.cast(f.data, C.POINTER(C.c_uint16)), shape=(f.full_window.height, f.full_window.width, 4))
        else:
            dst = np.ctypeslib.as_array(C.cast(f.data, C.POINTER(C.c_float)), shape=(f.full_window.height, f.full_window.width, 4))
        assert n == src.array.shape[0] * src.array.shape[1]
        dst[:] = src.array
        f.current_window = src.current_window
    return fill


def test_pull_dispatch_converts_between_formats(cvs, orc):
    rng = np.random.default_rng(21)
    full = (0, 0, 20, 9)
    h = rand_f16_frame(rng, full, (2, 1, 18, 8))
    f = rand_f32_frame(rng, full, (1, 2, 19, 7), lo=-2, hi=2)
    s16, k1 = _py_source(fill16=_fill_from([h]))
    s32, k2 = _py_source(fill32=_fill_from([f]))
    # f16 source pulled as f32 (main.c:115-139)
    out32 = HostFrame(full, np.float32, fill=9.0)
    cvs.video_get_frame_f32(C.byref(s16), 0, out32.ref())
    assert out32.current_window.tuple() == (2, 1, 18, 8)
    assert np.array_equal(out32.window_view().view(np.uint32), orc.half_to_float(h.window_view()).view(np.uint32))
    assert (out32.array[0] == 9.0).all()                                  # outside current_window: untouched
    # f32 source pulled as f16 (main.c:43-71)
    out16 = HostFrame(full, np.uint16, fill=0x1111)
    cvs.video_get_frame_f16(C.byref(s32), 0, out16.ref())
    assert out16.current_window.tuple() == (1, 2, 19, 7)
    assert np.array_equal(out16.window_view(), orc.float_to_half(f.window_view()))
    assert (out16.array[0] == 0x1111).all()
    # NULL source => empty window (main.c:35-38)
    cvs.video_get_frame_f16(None, 0, out16.ref())
    assert out16.current_window.is_empty()


def test_forced_pull_goes_through_the_device_slot(cvs, orc):
    """video_get_frame_f16_gl / _f32_gl (main.c:78-103,146-172; what force_gl=True calls, RgbaFrameF16.c:247-249): the
    pull through vtable slot 3 even when the source also fills a host slot.  Slot 3 is the device slot here: a vtable
    whose host slots would paint junk, with the workspace's device entry in slot 3, must yield the workspace's pixels;
    a source with no device slot yields an empty window, as the reference's does without a get_frame_gl (main.c:99-102)."""
    rng = np.random.default_rng(77)
    full = (0, 0, 23, 11)
    frames = [rand_f16_frame(rng, full, full, alpha="one"), rand_f16_frame(rng, full, (3, 2, 20, 9))]
    sources = [_py_source(fill16=_fill_from([fr])) for fr in frames]
    ws = cvs.workspace_create()
    for s, z in zip(sources, (0, 1)):
        cvs.workspace_add_item(ws, C.cast(C.pointer(s[0]), C.c_void_p), 0, 10, 0, z, None)
    vs = video_source()
    cvs.workspace_as_video_source(ws, C.byref(vs))
    assert vs.funcs.contents.flags & 1 and vs.funcs.contents.get_frame_dev          # VIDEO_SOURCE_FLAG_DEVICE

    want16, want32 = HostFrame(full, np.uint16), HostFrame(full, np.float32)
    cvs.video_get_frame_f16(C.byref(vs), 2, want16.ref())
    cvs.video_get_frame_f32(C.byref(vs), 2, want32.ref())

    junk = rand_f16_frame(rng, full, (0, 0, 5, 5))
    junk32 = rand_f32_frame(rng, full, (0, 0, 5, 5))
    decoy, keep = _py_source(fill16=_fill_from([junk]), fill32=_fill_from([junk32]))
    keep[2].flags = vs.funcs.contents.flags
    keep[2].get_frame_dev = vs.funcs.contents.get_frame_dev
    decoy.obj = vs.obj
    got16, got32 = HostFrame(full, np.uint16), HostFrame(full, np.float32)
    cvs.video_get_frame_f16(C.byref(decoy), 2, got16.ref())
    assert got16.current_window.tuple() == (0, 0, 5, 5)                              # ordinary pull: the host slot
    cvs.video_get_frame_f16_gl(C.byref(decoy), 2, got16.ref())
    cvs.video_get_frame_f32_gl(C.byref(decoy), 2, got32.ref())
    assert same_window(got16.current_window, want16.current_window) and same_window(got32.current_window, want32.current_window)
    assert_same_f16(got16.window_view(), want16.window_view(), "forced f16")
    assert_same_f32(got32.window_view(), want32.window_view(), "forced f32")

    plain = HostFrame(full, np.uint16)
    cvs.video_get_frame_f16_gl(C.byref(sources[1][0]), 0, plain.ref())              # no device slot: nothing, as in the reference
    assert plain.current_window.is_empty()
    cvs.video_get_frame_f16_gl(None, 0, plain.ref())
    assert plain.current_window.is_empty()
    cvs.workspace_free(ws)


def test_workspace_stack_host_and_device(cvs, orc):
    rng = np.random.default_rng(31)
    full = (0, 0, 31, 15)
    frames = [rand_f16_frame(rng, full, full, alpha="one"), rand_f16_frame(rng, full, (4, 2, 20, 12)),
              rand_f16_frame(rng, full, (10, 1, 31, 9))]
    sources = [_py_source(fill16=_fill_from([fr])) for fr in frames]
    ws = cvs.workspace_create()
    zs = [0, 7, 3]
    items = [cvs.workspace_add_item(ws, C.cast(C.pointer(s[0]), C.c_void_p), 0, 10, 0, z, None) for s, z in zip(sources, zs)]
    assert cvs.workspace_get_length(ws) == 3
    vs = video_source()
    cvs.workspace_as_video_source(ws, C.byref(vs))

    # the oracle's stack over the same sources
    oitems = (orc.ws_item * 3)(*[orc.ws_item(0, 10, z, 0, C.pointer(s[0])) for s, z in zip(sources, zs)])
    want32 = HostFrame(full, np.float32)
    orc.lib().orc_workspace_get_frame_f32(oitems, 3, 4, want32.ref())
    want16 = orc.float_to_half(want32.array)

    got32 = HostFrame(full, np.float32)
    cvs.video_get_frame_f32(C.byref(vs), 4, got32.ref())                 # host path
    assert same_window(got32.current_window, want32.current_window)
    assert_same_f32(got32.window_view(), want32.window_view(), "workspace host")

    got16 = HostFrame(full, np.uint16)                                   # through slot 3: whole stack in HBM
    dev = _lib.rgba_frame_dev(cvs.cvs_malloc(got16.array.nbytes), 1, got16.full_window, got16.full_window, None)
    cvs.video_get_frame_dev(C.byref(vs), 4, C.byref(dev))
    _lib.check(cvs.cvs_memcpy_d2h(got16.array.ctypes.data, dev.data, got16.array.nbytes, None))
    assert dev.current_window.tuple() == want32.current_window.tuple()
    x0, y0, x1, y1 = dev.current_window.tuple()
    assert_same_f16(got16.array[y0:y1 + 1, x0:x1 + 1], want16[y0:y1 + 1, x0:x1 + 1], "workspace device")
    cvs.cvs_free(dev.data)

    # membership: outside every item => empty (workspace.c:504-508)
    cvs.video_get_frame_f32(C.byref(vs), 10, got32.ref())
    assert got32.current_window.is_empty()
    # item API round trip
    x, ln, z = C.c_int64(), C.c_int64(), C.c_int64()
    cvs.workspace_get_item_pos(items[1], C.byref(x), C.byref(ln), C.byref(z))
    assert (x.value, ln.value, z.value) == (0, 10, 7)
    nz = C.c_int64(-5)
    cvs.workspace_update_item(items[1], None, None, C.byref(nz), None, None, None)
    cvs.workspace_remove_item(items[2])
    assert cvs.workspace_get_length(ws) == 2
    cvs.workspace_free(ws)


# ------------------------------------------------------------------ A9 scaler, A11 blur, Lanczos

SCALE_CASES = [
    # target full, source full, source current, target_point, source_point, factors
    ((0, 0, 31, 17), (0, 0, 15, 8), (0, 0, 15, 8), (0, 0), (0, 0), (2.0, 2.0)),
    ((0, 0, 31, 17), (0, 0, 15, 8), (2, 1, 13, 7), (0, 0), (0, 0), (2.0, 2.0)),
    ((0, 0, 15, 8), (0, 0, 31, 17), (0, 0, 31, 17), (0, 0), (0, 0), (0.5, 0.5)),       # the partial-coverage case
    ((0, 0, 15, 17), (0, 0, 31, 17), (0, 0, 31, 17), (0, 0), (0, 0), (0.5, 1.0)),       # x only
    ((0, 0, 31, 8), (0, 0, 31, 17), (1, 1, 30, 16), (0, 0), (0, 0), (1.0, 0.5)),        # y only
    ((0, 0, 40, 30), (0, 0, 15, 8), (0, 0, 15, 8), (3.5, 2.25), (1.0, 0.5), (2.5, 3.0)),  # fractional points
    ((0, 0, 20, 40), (0, 0, 15, 8), (0, 0, 15, 8), (0, 0), (0, 0), (1.3, 4.0)),         # x first
    ((0, 0, 15, 8), (0, 0, 15, 8), (0, 0, 15, 8), (0, 0), (0, 0), (1.0, 1.0)),          # identity
    ((0, 0, 15, 8), (0, 0, 15, 8), (0, 0, 15, 8), (2.0, 0.0), (0, 0), (1.0, 1.0)),      # shift only: factor 1, points differ
    ((-8, -4, 23, 13), (0, 0, 15, 8), (0, 0, 15, 8), (0, 0), (8.0, 4.0), (2.0, 2.0)),   # negative origin
]


@pytest.mark.parametrize("tfull,sfull,scur,tp,sp,fac", SCALE_CASES)
def test_scale_bilinear(cvs, orc, tfull, sfull, scur, tp, sp, fac):
    rng = np.random.default_rng(41)
    src = rand_f32_frame(rng, sfull, scur)
    got = rand_f32_frame(rng, tfull)
    want = got.copy()
    cvs.video_scale_bilinear_f32(got.ref(), v2f(*tp), src.ref(), v2f(*sp), v2f(*fac))
    orc.lib().orc_scale_bilinear_f32(want.ref(), v2f(*tp), src.ref(), v2f(*sp), v2f(*fac))
    assert same_window(got.current_window, want.current_window), (got.current_window, want.current_window)
    assert_same_f32(got.array, want.array, "scale")


def test_scale_pull(cvs, orc):
    rng = np.random.default_rng(43)
    sfull = (0, 0, 63, 35)
    big = rand_f32_frame(rng, sfull)

    def fill(idx, f):
        fw = f.full_window
        dst = np.ctypeslib.as_array(C.cast(f.data, C.POINTER(C.c_float)), shape=(fw.height, fw.width, 4))
        dst[:] = big.array[fw.min.y:fw.max.y + 1, fw.min.x:fw.max.x + 1]
        f.current_window = fw

    src, keep = _py_source(fill32=fill)
    rect = box2i.of(*sfull)
    for fac in [(2.0, 2.0), (0.5, 1.0), (0.0, 1.0), (1.0, 1.0)]:
        got, want = HostFrame((0, 0, 31, 17), np.float32), HostFrame((0, 0, 31, 17), np.float32)
        cvs.video_scale_bilinear_f32_pull(got.ref(), v2f(4, 2), C.byref(src), 0, C.byref(rect), v2f(10, 6), v2f(*fac))
        orc.lib().orc_scale_bilinear_f32_pull(want.ref(), v2f(4, 2), C.byref(src), 0, C.byref(rect), v2f(10, 6), v2f(*fac))
        assert same_window(got.current_window, want.current_window)
        if not want.current_window.is_empty():
            assert_same_f32(got.array, want.array, "scale pull %r" % (fac,))


@pytest.mark.parametrize("scur", [(0, 0, 47, 26), (5, 3, 40, 20)])
@pytest.mark.parametrize("ntaps", [9, 5, 1, 4, 17, 21, 31, 33])
def test_fir_blur(cvs, orc, scur, ntaps):
    rng = np.random.default_rng(51)
    full = (0, 0, 47, 26)
    src = rand_f32_frame(rng, full, scur)
    taps = synth.gaussian_taps(ntaps, max(1.5, ntaps / 6.0))
    want = HostFrame(full, np.float32)
    orc.lib().orc_fir_blur_f32(want.ref(), src.ref(), f32p(taps), ntaps)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(full, np.float32)
    _lib.check(cvs.cvs_fir_blur_f32_dev(d_out.ref(), d_src.ref(), f32p(taps), ntaps, None))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f32(got.window_view(), want.window_view(), "blur")


@pytest.mark.parametrize("fx,fy,tsize", [(0.5, 0.5, (32, 18)), (0.25, 0.5, (16, 18)), (2.0, 1.5, (128, 54))])
def test_lanczos_resample(cvs, orc, fx, fy, tsize):
    rng = np.random.default_rng(61)
    src = rand_f32_frame(rng, (0, 0, 63, 35))
    tfull = (0, 0, tsize[0] - 1, tsize[1] - 1)
    want = HostFrame(tfull, np.float32)
    orc.lib().orc_resample_lanczos_f32(want.ref(), src.ref(), C.c_float(fx), C.c_float(fy), 3)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.float32)
    _lib.check(cvs.cvs_resample_lanczos_f32_dev(d_out.ref(), d_src.ref(), C.c_float(fx), C.c_float(fy), 3, None))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f32(got.array, want.array, "lanczos")


@pytest.mark.parametrize("ssize,scur,tsize,fx,fy", [
    ((128, 72), None, (64, 36), 0.5, 0.5),
    ((400, 300), (7, 5, 380, 290), (160, 120), 0.4, 0.4),
    ((300, 200), None, (225, 150), 0.75, 0.75),
    ((96, 54), None, (144, 81), 1.5, 1.5),
    ((130, 70), None, (40, 200), 0.3, 3.0),
])
def test_lanczos_resample_between_f16_frames(cvs, orc, ssize, scur, tsize, fx, fy):
    """cvs_resample_lanczos_f16_dev = widen, the two f32 passes, truncate -- with no f32 frame; and the same thing reached
    through cvs_blur_lanczos_f16_dev with the identity blur (one tap of weight 1)."""
    full = (0, 0, ssize[0] - 1, ssize[1] - 1)
    src16 = HostFrame(full, np.uint16, synth.layer_pixels(ssize[0], ssize[1], 2, 7), scur)
    src32 = HostFrame(full, np.float32, orc.half_to_float(src16.array), scur)
    tfull = (0, 0, tsize[0] - 1, tsize[1] - 1)
    want32 = HostFrame(tfull, np.float32)
    orc.lib().orc_resample_lanczos_f32(want32.ref(), src32.ref(), C.c_float(fx), C.c_float(fy), 3)
    want = orc.float_to_half(want32.array)
    d_src, d_out = DeviceFrame.from_host(src16), DeviceFrame(tfull, np.uint16)
    _lib.check(cvs.cvs_resample_lanczos_f16_dev(d_out.ref(), d_src.ref(), C.c_float(fx), C.c_float(fy), 3, None))
    got = d_out.download()
    assert same_window(got.current_window, want32.current_window)
    assert_same_f16(got.array, want, "f16 lanczos %r" % ((fx, fy),))
    one = np.array([1.0], np.float32)
    _lib.check(cvs.cvs_memset(d_out.ptr, 0x11, d_out.nbytes, None))
    _lib.check(cvs.cvs_blur_lanczos_f16_dev(d_out.ref(), d_src.ref(), f32p(one), 1, C.c_float(fx), C.c_float(fy), 3, None))
    assert_same_f16(d_out.download().array, want, "identity blur + lanczos %r" % ((fx, fy),))


@pytest.fixture
def force_fir(request):
    """cvs_fir_path_override pins the general FIR path to one of its kernels (speed only: the point of these tests is that
    the pixels do not change)."""
    lib = _lib.load()

    def pin(which):
        mode = _lib.FIR_PATH_AUTO
        if which:
            mode |= _lib.FIR_PATH_TABLES                    # blurs too: past the register-window kernel, to the table kernels
        if which == "passes":                               # no launch that does both passes: two k_fir launches through an f32 frame
            mode |= _lib.FIR_PATH_PASSES
        elif which == "tiled":
            mode |= _lib.FIR_PATH_TILED
        elif which == "hv":
            mode |= _lib.FIR_PATH_HV
        if which == "strips":                               # the fused vertical-first scaler on k_fir_vh, never its tile form
            mode = _lib.FIR_PATH_STRIPS
        elif which == "tiles":                              # ... on the tile form wherever it takes the call, whatever the size
            mode = _lib.FIR_PATH_TILES
        lib.cvs_fir_path_override(mode)
    yield pin
    pin(None)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if _FIR_SEEN and os.path.isdir(out):                       # which kernel every pinned case went to, for the record
        import json
        with open(os.path.join(out, "fir_kernels_seen.json"), "w") as f:
            json.dump(_FIR_SEEN, f, indent=0, sort_keys=True)


_KERNEL_NAMES = {_lib.FIR_KERNEL_NONE: "none", _lib.FIR_KERNEL_WINDOW: "window", _lib.FIR_KERNEL_HALVE: "halve",
                 _lib.FIR_KERNEL_VH: "vh", _lib.FIR_KERNEL_TILED: "tiled",
                 _lib.FIR_KERNEL_TWO_PASS: "two-pass", _lib.FIR_KERNEL_PASS: "pass", _lib.FIR_KERNEL_HV: "hv",
                 _lib.FIR_KERNEL_WINDOW_PAIR: "window-pair", _lib.FIR_KERNEL_HALVE_PAIR: "halve-pair", _lib.FIR_KERNEL_TILE_VH: "tile-vh"}
_FIR_SEEN = {}


def ran_on(cvs, forced, fallback=None, note=None):
    """cvs_fir_last_kernel() after a launch pinned with force_fir(forced): the pinned kernel, or -- for a table pair that
    kernel does not take -- the documented fallback named by the caller (host/scale.c fir2d_launch: hv -> tiled -> the two
    passes; "passes" pins the last).  The per-line gather (hv) exists in the default arithmetic flavour only: in the contracted
    one a table pair goes to the tiles, or to the two passes when its footprint does not fit them -- same sums, and that is
    what the test then checks."""
    got = _KERNEL_NAMES[cvs.cvs_fir_last_kernel()]
    contracted = cvs.cvs_get_arithmetic() == _lib.ARITH_CONTRACTED
    if note is not None:
        _FIR_SEEN[note + (" [contracted]" if contracted else "")] = got
    want = {"passes": "two-pass"}.get(fallback or forced, fallback or forced)
    if contracted and want == "hv":
        assert got in ("tiled", "two-pass"), "contracted flavour, pinned to %r: expected the tiles or the two passes, ran on %r" % (forced, got)
    else:
        assert got == want, "pinned to %r (expected to run on %r), ran on %r" % (forced, want, got)
    assert _lib.last_error() == "", _lib.last_error()
    assert cvs.cvs_fir_fell_through_count() == 0            # a fused kernel that was chosen and failed to launch is counted (and logged)
    return got


# cases of the tests below that the pinned kernel does NOT take, and where they go instead
_FALLBACK = {}


@pytest.mark.parametrize("kernel", ["hv", "passes", "tiled"])
@pytest.mark.parametrize("ssize,scur,tsize,fx,fy", [
    ((64, 36), None, (32, 18), 0.5, 0.5),
    ((400, 300), None, (160, 120), 0.4, 0.4),                # several strips of 128 columns, several row segments
    ((400, 300), (7, 5, 380, 290), (160, 120), 0.4, 0.4),    # source window inside its buffer: border lines have short tap lists
    ((300, 200), None, (100, 150), 1.0 / 3.0, 0.75),         # a factor that is not a power of two: every line its own taps
    ((96, 54), None, (240, 81), 2.5, 1.5),                   # enlarging: one source row feeds up to 15 target rows
    ((96, 54), None, (192, 108), 2.0, 2.0),                  # ... up to 26: the sweep's 32-slot instance
    ((64, 40), (2, 1, 60, 37), (100, 84), 1.5625, 2.1),      # the same with clipped lists at the window's edges
    ((130, 70), None, (40, 200), 0.3, 3.0),                  # more target rows than the source covers: lines without taps
])
def test_lanczos_resample_both_kernels(cvs, orc, force_fir, kernel, ssize, scur, tsize, fx, fy):
    """The general resampler has three kernels (the per-line gather, tiles in LDS, the lane-per-pixel sweep); whichever a
    table pair would get, all must give the gather's sums bit for bit."""
    rng = np.random.default_rng(62)
    sfull = (0, 0, ssize[0] - 1, ssize[1] - 1)
    src = rand_f32_frame(rng, sfull, scur, lo=-0.5, hi=1.5)
    tfull = (0, 0, tsize[0] - 1, tsize[1] - 1)
    want = HostFrame(tfull, np.float32)
    orc.lib().orc_resample_lanczos_f32(want.ref(), src.ref(), C.c_float(fx), C.c_float(fy), 3)
    force_fir(kernel)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.float32)
    cvs.cvs_clear_last_error()
    _lib.check(cvs.cvs_resample_lanczos_f32_dev(d_out.ref(), d_src.ref(), C.c_float(fx), C.c_float(fy), 3, None))
    ran_on(cvs, kernel, _FALLBACK.get((kernel, (fx, fy))), note="lanczos %s %r" % (kernel, (ssize, scur, fx, fy)))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f32(got.array, want.array, "lanczos (%s)" % kernel)
    # f16 on both sides through the same kernel
    src16 = HostFrame(sfull, np.uint16, synth.truncate_to_half(src.array), scur or sfull)
    want16 = _oracle_config3(orc, src16, tsize, np.array([1.0], np.float32), fx, fy)
    d16, o16 = DeviceFrame.from_host(src16), DeviceFrame(tfull, np.uint16)
    _lib.check(cvs.cvs_blur_lanczos_f16_dev(o16.ref(), d16.ref(), f32p(np.array([1.0], np.float32)), 1, C.c_float(fx), C.c_float(fy), 3, None))
    ran_on(cvs, kernel, _FALLBACK.get((kernel, (fx, fy))))
    assert_same_f16(o16.download().array, want16.array, "lanczos f16 (%s)" % kernel)


@pytest.mark.parametrize("kernel", ["hv", "passes", "tiled"])
@pytest.mark.parametrize("ssize,tsize,fx,fy", [((400, 300), (160, 120), 0.4, 0.4), ((96, 54), (144, 81), 1.5, 1.5), ((300, 200), (225, 150), 0.75, 0.75),
                                               ((96, 54), (192, 108), 2.0, 2.0)])
def test_lanczos_resample_with_inf_and_nan_pixels(cvs, orc, force_fir, kernel, ssize, tsize, fx, fy):
    """Inf and NaN in the source spread exactly as far as the taps that touch them: a tap the gather skips (a padded list
    entry, an accumulator slot that does not take the row) must not turn them into NaNs elsewhere (0 * Inf)."""
    rng = np.random.default_rng(64)
    sfull = (0, 0, ssize[0] - 1, ssize[1] - 1)
    src = rand_f32_frame(rng, sfull, None, lo=-0.5, hi=1.5)
    h, w = src.array.shape[:2]
    for k, v in enumerate([np.inf, -np.inf, np.nan, np.inf, np.nan, -np.inf]):
        src.array[(37 * k + 5) % h, (53 * k + 11) % w, k % 4] = v
    src.array[h - 1, w - 1, :] = np.inf                    # a frame corner, where lists are clipped
    src.array[0, 0, 1] = np.nan
    tfull = (0, 0, tsize[0] - 1, tsize[1] - 1)
    want = HostFrame(tfull, np.float32)
    orc.lib().orc_resample_lanczos_f32(want.ref(), src.ref(), C.c_float(fx), C.c_float(fy), 3)
    assert 0 < np.count_nonzero(~np.isfinite(want.array)) < want.array.size // 4
    force_fir(kernel)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.float32)
    cvs.cvs_clear_last_error()
    _lib.check(cvs.cvs_resample_lanczos_f32_dev(d_out.ref(), d_src.ref(), C.c_float(fx), C.c_float(fy), 3, None))
    ran_on(cvs, kernel, _FALLBACK.get((kernel, (fx, fy))), note="lanczos non-finite %s %r" % (kernel, (fx, fy)))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f32(got.array, want.array, "lanczos with non-finite pixels (%s)" % kernel)


@pytest.mark.parametrize("kernel", ["hv", "passes", "tiled"])
@pytest.mark.parametrize("ntaps", [1, 2, 4, 10, 16])
def test_even_and_short_blurs_both_kernels(cvs, orc, force_fir, kernel, ntaps):
    rng = np.random.default_rng(63)
    full, cur = (-3, -2, 300, 90), (5, 1, 280, 77)
    src = rand_f32_frame(rng, full, cur, lo=-0.5, hi=1.5)
    taps = rng.uniform(-0.2, 1.0, ntaps).astype(np.float32)
    want = HostFrame(full, np.float32)
    orc.lib().orc_fir_blur_f32(want.ref(), src.ref(), f32p(taps), ntaps)
    force_fir(kernel)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(full, np.float32)
    cvs.cvs_clear_last_error()
    _lib.check(cvs.cvs_fir_blur_f32_dev(d_out.ref(), d_src.ref(), f32p(taps), ntaps, None))
    ran_on(cvs, kernel, _FALLBACK.get((kernel, ntaps)), note="blur %s %d" % (kernel, ntaps))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f32(got.window_view(), want.window_view(), "blur %d taps (%s)" % (ntaps, kernel))


@pytest.mark.parametrize("f,fmt", [(0.4, "f16"), (0.4, "f32"), (0.75, "f16"), (0.75, "f32"), (1.5, "f16"), (1.5, "f32"),
                                   (1.0 / 3.0, "f16"), (1.0 / 3.0, "f32"), (2.0, "f16")])
def test_full_size_resample_agrees_between_the_kernels(cvs, force_fir, f, fmt):
    """3840x2160 Lanczos3 at factors that take different instances of the per-line gather, f16 and f32 frames: the tiled
    kernel and the gather -- independent code -- must produce the same frame bit for bit (each is checked against the oracle at
    small sizes, and bench.py proves the f16 frames at 0.4x / 0.75x / 1.5x against SHA-256 fixtures of the oracle)."""
    w, h = 3840, 2160
    tw, th = int(w * f), int(h * f)
    src16 = synth.layer_frame(w, h, 1, 0)
    if fmt == "f16":
        d_src, dtype = DeviceFrame.from_host(src16), np.uint16
    else:
        from tests.models import h2f_ieee
        d_src, dtype = DeviceFrame.from_host(HostFrame(src16.full_window, np.float32, h2f_ieee(src16.array).astype(np.float32))), np.float32
    outs = []
    cvs.cvs_clear_last_error()
    for kernel in (None, "tiled", "hv"):
        force_fir(kernel)
        d_out = DeviceFrame((0, 0, tw - 1, th - 1), dtype)
        if fmt == "f16":
            _lib.check(cvs.cvs_resample_lanczos_f16_dev(d_out.ref(), d_src.ref(), C.c_float(f), C.c_float(f), 3, None))
        else:
            _lib.check(cvs.cvs_resample_lanczos_f32_dev(d_out.ref(), d_src.ref(), C.c_float(f), C.c_float(f), 3, None))
        # the automatic choice (host/scale.c fir2d_launch): the per-line gather
        ran_on(cvs, kernel or "hv", note="full size %s %r %s" % (kernel, f, fmt))
        got = d_out.download()
        assert got.current_window.tuple() == (0, 0, tw - 1, th - 1)
        outs.append(got.array.copy())
        d_out.free()
    d_src.free()
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    assert len(np.unique(outs[0])) > 1000


@pytest.mark.parametrize("ssize,tsize,fx,fy", [((600, 400), (60, 40), 0.1, 0.1), ((500, 64), (40, 64), 0.08, 1.0), ((300, 900), (150, 60), 0.5, 1.0 / 15.0)])
def test_lanczos_far_below_one(cvs, orc, ssize, tsize, fx, fy):
    """Tap lists of 59-90: beyond the sweep kernel's registers and the tiles' LDS.  The cached tables then run as two gather
    launches through an f32 frame (no allocation, upload or wait on the way) -- same sums."""
    rng = np.random.default_rng(64)
    src = rand_f32_frame(rng, (0, 0, ssize[0] - 1, ssize[1] - 1), lo=-0.5, hi=1.5)
    tfull = (0, 0, tsize[0] - 1, tsize[1] - 1)
    want = HostFrame(tfull, np.float32)
    orc.lib().orc_resample_lanczos_f32(want.ref(), src.ref(), C.c_float(fx), C.c_float(fy), 3)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.float32)
    for _ in range(2):
        _lib.check(cvs.cvs_resample_lanczos_f32_dev(d_out.ref(), d_src.ref(), C.c_float(fx), C.c_float(fy), 3, None))
        got = d_out.download()
        assert same_window(got.current_window, want.current_window)
        assert_same_f32(got.array, want.array, "lanczos %r" % ((fx, fy),))


def _oracle_config3(orc, src16, tsize, taps, fx, fy):
    """widen -> blur -> Lanczos -> truncate with the oracle's pieces."""
    src32 = HostFrame(src16.full_window, np.float32, orc.half_to_float(src16.array), src16.current_window)
    blurred = HostFrame(src16.full_window, np.float32)
    orc.lib().orc_fir_blur_f32(blurred.ref(), src32.ref(), f32p(taps), len(taps))
    small = HostFrame((0, 0, tsize[0] - 1, tsize[1] - 1), np.float32)
    orc.lib().orc_resample_lanczos_f32(small.ref(), blurred.ref(), C.c_float(fx), C.c_float(fy), 3)
    return HostFrame(small.full_window, np.uint16, orc.float_to_half(small.array), small.current_window)


@pytest.mark.parametrize("ssize,tsize,fx,fy", [((128, 72), (64, 36), 0.5, 0.5), ((97, 55), (49, 28), 0.5, 0.5), ((64, 36), (128, 54), 2.0, 1.5),
                                               ((200, 40), (20, 40), 0.1, 1.0)])      # the last one does not fit an LDS tile: fallback path
def test_config3_pipeline_f16(cvs, orc, ssize, tsize, fx, fy):
    layer = synth.layer_frame(ssize[0], ssize[1], 1, 0)
    taps = synth.gaussian_taps(9, 1.5)
    want = _oracle_config3(orc, layer, tsize, taps, fx, fy)
    d_src = DeviceFrame.from_host(layer)
    d_out = DeviceFrame((0, 0, tsize[0] - 1, tsize[1] - 1), np.uint16)
    for _ in range(2):                                     # second call runs from the cached tap tables
        _lib.check(cvs.cvs_blur_lanczos_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), 9, C.c_float(fx), C.c_float(fy), 3, None))
        got = d_out.download()
        assert same_window(got.current_window, want.current_window)
        assert_same_f16(got.array, want.array, "config 3 pipeline %r -> %r" % (ssize, tsize))


@pytest.mark.parametrize("ntaps", [3, 5, 7, 9, 11, 13])          # 13: no fused instance, the two-launch form
@pytest.mark.parametrize("ssize,scur,tsize", [
    ((300, 170), None, (150, 85)),                     # several strips at 128 and 256 lanes, several segments
    ((300, 170), (9, 6, 280, 150), (150, 85)),         # source window inside its buffer: blurred pixels OUTSIDE it must count as skipped taps
    ((131, 77), (0, 0, 130, 76), (80, 50)),            # target larger than half the source: lines whose taps all fall outside
    ((64, 36), (20, 10, 40, 30), (32, 18)),            # a window far from the buffer's edges
])
def test_blur_then_halving_in_one_sweep(cvs, orc, ntaps, ssize, scur, tsize):
    """cvs_blur_lanczos_f16_dev at factor 1/2 with an odd blur runs blur x, blur y, resample x, resample y in ONE kernel
    (blur_halve_ops.hip): no f32 frame in between.  Same sums in the same order as the two nodes of the oracle."""
    rng = np.random.default_rng(ntaps * 1000 + ssize[0])
    taps = rng.uniform(0.02, 0.3, ntaps).astype(np.float32)
    taps = (taps / taps.sum(dtype=np.float32)).astype(np.float32)
    full = (0, 0, ssize[0] - 1, ssize[1] - 1)
    px = synth.layer_pixels(ssize[0], ssize[1], 1, 5)
    layer = HostFrame(full, np.uint16, px, scur)
    want = _oracle_config3(orc, layer, tsize, taps, 0.5, 0.5)
    d_src = DeviceFrame.from_host(layer)
    d_out = DeviceFrame((0, 0, tsize[0] - 1, tsize[1] - 1), np.uint16)
    _lib.check(cvs.cvs_memset(d_out.ptr, 0x5A, d_out.nbytes, None))
    _lib.check(cvs.cvs_blur_lanczos_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), ntaps, C.c_float(0.5), C.c_float(0.5), 3, None))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.array, want.array, "blur %d taps + halving, %r window %r" % (ntaps, ssize, scur))


@pytest.mark.parametrize("ntaps", [3, 5, 7, 9, 11])
@pytest.mark.parametrize("ssize,scur,tsize,pairs", [
    ((300, 170), None, (150, 85), True),                      # three strips of 54 target columns, several segments
    ((300, 170), (8, 6, 279, 150), (150, 85), True),          # source window inside its buffer, on pair boundaries: blurred pixels outside it are skipped taps
    ((300, 170), (9, 6, 280, 150), (150, 85), False),         # ... starting on an odd column: pairs would straddle its edge
    ((130, 77), (0, 0, 129, 76), (80, 50), True),             # target larger than half the source: lines whose taps all fall outside
    ((64, 36), (20, 10, 41, 30), (32, 18), True),             # a window far from the buffer's edges, one partly filled strip
    ((131, 40), None, (65, 20), False),                       # an odd source width
    ((560, 64), None, (280, 32), True),                       # wide enough for the two-wave workgroups (strips of 118 target columns)
    ((560, 64), (12, 4, 541, 59), (280, 32), True),           # ... with the source's window inside its buffer
])
def test_blur_then_halving_two_columns_per_lane(cvs, orc, blur_columns, ntaps, ssize, scur, tsize, pairs):
    """k_blur_halve_pair (one-wave workgroups, two source columns per lane, buffer loads, the second stage's rows in a moving
    window) against the oracle's two nodes and, code for code, against k_blur_halve; launches it cannot take say so."""
    rng = np.random.default_rng(ntaps * 1000 + ssize[0] + 7)
    taps = rng.uniform(0.02, 0.3, ntaps).astype(np.float32)
    taps = (taps / taps.sum(dtype=np.float32)).astype(np.float32)
    full = (0, 0, ssize[0] - 1, ssize[1] - 1)
    layer = HostFrame(full, np.uint16, synth.layer_pixels(ssize[0], ssize[1], 1, 5), scur)
    want = _oracle_config3(orc, layer, tsize, taps, 0.5, 0.5)
    d_src = DeviceFrame.from_host(layer)
    got = {}
    for columns in (2, 1):
        d_out = DeviceFrame((0, 0, tsize[0] - 1, tsize[1] - 1), np.uint16)
        _lib.check(cvs.cvs_memset(d_out.ptr, 0x5A, d_out.nbytes, None))
        cvs.cvs_clear_last_error()
        blur_columns(columns)
        _lib.check(cvs.cvs_blur_lanczos_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), ntaps, C.c_float(0.5), C.c_float(0.5), 3, None))
        assert _blur_kernel_seen(cvs) == ("halve-pair" if columns == 2 and pairs else "halve")
        got[columns] = d_out.download()
        assert same_window(got[columns].current_window, want.current_window)
    assert_same_f16(got[2].array, want.array, "blur %d taps + halving on two columns per lane, %r window %r" % (ntaps, ssize, scur))
    assert_same_f16(got[2].array, got[1].array, "two columns per lane against one")


@pytest.mark.parametrize("ssize,scur,tsize,pairs", [
    ((300, 170), None, (150, 85), True),
    ((300, 170), (8, 6, 279, 150), (150, 85), True),          # source window inside its buffer: taps outside it are skipped
    ((300, 170), (9, 6, 280, 150), (150, 85), False),         # ... off the pair grid: the one-column kernel
    ((560, 64), None, (280, 32), True),                       # the two-wave workgroups
    ((130, 77), (0, 0, 129, 76), (80, 50), True),             # target larger than half the source
])
def test_lanczos_halving_alone_on_two_columns_per_lane(cvs, orc, blur_columns, ssize, scur, tsize, pairs):
    """cvs_resample_lanczos_f16_dev at 1/2 on f16 frames: config 3's two-column sweep behind an identity blur (one tap of
    weight 1), against the oracle's resampler and against the decimating register-window kernel it replaces."""
    full = (0, 0, ssize[0] - 1, ssize[1] - 1)
    px = synth.layer_pixels(ssize[0], ssize[1], 1, 9)
    px[11, 40] = [0x7C00, 0xFC00, 0x7E00, 0x3C00]
    layer = HostFrame(full, np.uint16, px, scur)
    want = _oracle_config3(orc, layer, tsize, np.array([1.0], np.float32), 0.5, 0.5)
    d_src = DeviceFrame.from_host(layer)
    got = {}
    for columns in (2, 1):
        d_out = DeviceFrame((0, 0, tsize[0] - 1, tsize[1] - 1), np.uint16)
        _lib.check(cvs.cvs_memset(d_out.ptr, 0x5A, d_out.nbytes, None))
        cvs.cvs_clear_last_error()
        blur_columns(columns)
        _lib.check(cvs.cvs_resample_lanczos_f16_dev(d_out.ref(), d_src.ref(), C.c_float(0.5), C.c_float(0.5), 3, None))
        assert _blur_kernel_seen(cvs) == ("halve-pair" if columns == 2 and pairs else "window")
        got[columns] = d_out.download()
        assert same_window(got[columns].current_window, want.current_window)
    assert_same_f16(got[2].array, want.array, "Lanczos3 halving alone on two columns per lane, %r window %r" % (ssize, scur))
    assert_same_f16(got[2].array, got[1].array, "two columns per lane against one")


def test_blur_then_halving_two_columns_with_special_values(cvs, orc, blur_columns):
    """Inf, NaN, the largest halfs (sums beyond the half range), denormals and whole black regions through both stages."""
    w, h = 260, 96
    full = (0, 0, w - 1, h - 1)
    px = synth.layer_pixels(w, h, 1, 3)
    px[:, 40:90] = 0
    px[5:9, 150:170, :3] = 0x7BFF
    px[20, 200] = [0x7C00, 0xFC00, 0x7E00, 0x3C00]
    px[31, 7] = [0x0001, 0x8001, 0x03FF, 0x0001]
    px[h - 1, w - 1] = [0x7C00, 0x7C00, 0x7C00, 0x7C00]
    layer = HostFrame(full, np.uint16, px, full)
    taps = synth.gaussian_taps(9, 1.5)
    want = _oracle_config3(orc, layer, (w // 2, h // 2), taps, 0.5, 0.5)
    d_src, d_out = DeviceFrame.from_host(layer), DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
    cvs.cvs_clear_last_error()
    blur_columns(2)
    _lib.check(cvs.cvs_blur_lanczos_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
    assert _blur_kernel_seen(cvs) == "halve-pair"
    assert_same_f16(d_out.download().array, want.array, "special values through blur + halving")


def test_blur_lanczos_batch_on_two_columns_per_lane(cvs, orc, blur_columns):
    """The batched launch (grid.z = frame) of k_blur_halve_pair: five frames, each against the oracle."""
    w, h, count = 244, 90, 5
    full = (0, 0, w - 1, h - 1)
    taps = synth.gaussian_taps(9, 1.5)
    srcs = [HostFrame(full, np.uint16, synth.layer_pixels(w, h, 1, 20 + i), full) for i in range(count)]
    d_src = [DeviceFrame.from_host(f) for f in srcs]
    d_out = [DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16) for _ in range(count)]
    cvs.cvs_clear_last_error()
    blur_columns(2)
    _lib.check(cvs.cvs_blur_lanczos_f16_batch_dev(_frame_table(d_out), _frame_table(d_src), count, f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
    assert _blur_kernel_seen(cvs) == "halve-pair"
    for i in range(count):
        want = _oracle_config3(orc, srcs[i], (w // 2, h // 2), taps, 0.5, 0.5)
        assert_same_f16(d_out[i].download().array, want.array, "frame %d of the batch" % i)


def test_config3_full_size_properties(cvs, orc):
    """3840x2160 -> 1920x1080: a constant frame stays constant away from the borders (taps sum to 1 within
    float rounding), and the top-left corner block equals the oracle run on a crop that contains its footprint."""
    w, h = 3840, 2160
    taps = synth.gaussian_taps(9, 1.5)
    layer = synth.layer_frame(w, h, 1, 0)
    d_src = DeviceFrame.from_host(layer)
    d_out = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
    _lib.check(cvs.cvs_blur_lanczos_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
    got = d_out.download()
    assert got.current_window.tuple() == (0, 0, w // 2 - 1, h // 2 - 1)
    # corner: output block (0..47, 0..26) depends on source (0..~110, 0..~70); a 256x160 crop reproduces it exactly
    crop = HostFrame((0, 0, 255, 159), np.uint16, layer.array[:160, :256])
    want = _oracle_config3(orc, crop, (128, 80), taps, 0.5, 0.5)
    assert_same_f16(got.array[:27, :48], want.array[:27, :48], "4K config 3, corner block")
    const = HostFrame((0, 0, w - 1, h - 1), np.uint16, fill=0x3800)        # 0.5 everywhere
    d_src.upload(const.array)
    _lib.check(cvs.cvs_blur_lanczos_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
    inner = orc.half_to_float(d_out.download().array[16:-16, 16:-16])
    assert np.abs(inner - 0.5).max() < 2e-3


# ------------------------------------------------------------------ BASELINE sizes

def test_config2_full_4k_frame_against_oracle(cvs, orc):
    """3840x2160, 2 layers, Rec.709 LUT + RGB->Y'PbPr + over: one whole frame, every pixel."""
    w, h = 3840, 2160
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    layers = _synth_layers(w, h, 2)
    want = orc.chain_color_over(layers, m, orc.transfer_table(0), None)
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
    chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 1
    assert_same_f16(out.download().array, want.array, "4K config 2")


def test_config4_8k_three_layer_properties(cvs, orc):
    """7680x4320 3-layer over stack through size-independent properties:
    an opaque top layer wins; a fully transparent layer is the identity; the first rows equal the oracle."""
    w, h = 7680, 4320
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
    rng = np.random.default_rng(4)
    base = synth.layer_pixels(w, h, 0, 0)
    mid = synth.layer_pixels(w, h, 1, 0)
    clear = mid.copy()
    clear[..., 3] = 0
    opaque = synth.layer_pixels(w, h, 2, 0)
    opaque[..., 3] = 0x3C00
    full = (0, 0, w - 1, h - 1)

    def run(arrays):
        dl = []
        for a in arrays:
            d = DeviceFrame(full, np.uint16)
            d.upload(a)
            dl.append(d)
        out = DeviceFrame(full, np.uint16)
        chain_color_over([(out, dl)], ident)
        _lib.check(cvs.cvs_stream_sync(None))
        res = out.download().array
        for d in dl + [out]:
            d.free()
        return res

    top_wins = run([base, mid, opaque])
    assert np.array_equal(top_wins, opaque)                    # (x*0 + c*1)/1 == c exactly
    two = run([base, mid])
    three = run([base, clear, mid])
    assert np.array_equal(two, three)                          # alpha 0 layer changes nothing
    # head of the frame against the oracle
    rows = 8
    heads = [HostFrame((0, 0, w - 1, rows - 1), np.uint16, a[:rows]) for a in (base, mid)]
    want = orc.chain_color_over(heads, ident, None, None)
    assert_same_f16(two[:rows], want.array, "8K head rows")
    del rng


# ------------------------------------------------------------------ config 5 pieces: plain stack, out-of-place colour, f16 blur

@pytest.mark.parametrize("nlayers", [1, 2, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("translucent", [False, True])
def test_plain_over_stack_matches_oracle(cvs, orc, nlayers, translucent):
    """m == NULL: no colour node, just the workspace stack of f16 layers."""
    w, h = 71, 23
    full = (0, 0, w - 1, h - 1)
    if translucent:
        rng = np.random.default_rng(300 + nlayers)
        from canvas_amd.synth import truncate_to_half
        layers = [HostFrame(full, np.uint16, truncate_to_half(rand_f32_frame(rng, full, full, alpha="mixed", lo=-0.25, hi=1.5).array))
                  for _ in range(nlayers)]
    else:
        layers = _synth_layers(w, h, nlayers)
    want = orc.chain_color_over(layers, None)
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame(full, np.uint16)
    chain_color_over([(out, dl)], None)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 1
    assert_same_f16(out.download().array, want.array, "plain stack, %d layers" % nlayers)


@pytest.mark.parametrize("size", [(1, 1), (2, 1), (3, 1), (1, 5)])
def test_chain_on_tiny_frames(cvs, orc, size):
    """One pixel cannot form a pair: the call goes node by node; two and three pixels run the pair kernel + tail."""
    w, h = size
    full = (0, 0, w - 1, h - 1)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    layers = _synth_layers(w, h, 3)
    for matrix, pre, table in [(m, _lib.LUT_REC709_TO_LINEAR_SCENE, orc.transfer_table(0)), (None, _lib.LUT_NONE, None)]:
        want = orc.chain_color_over(layers, matrix, table, None)
        dl = [DeviceFrame.from_host(l) for l in layers]
        out = DeviceFrame(full, np.uint16)
        chain_color_over([(out, dl)], matrix, pre, _lib.LUT_NONE)
        _lib.check(cvs.cvs_stream_sync(None))
        assert cvs.cvs_chain_last_was_fused() == (0 if w * h < 2 else 1)
        got = out.download()
        assert same_window(got.current_window, want.current_window)
        assert_same_f16(got.array, want.array, "chain on %dx%d" % size)


def test_plain_stack_rejects_tables_and_takes_ragged_windows(cvs, orc):
    full = (0, 0, 47, 19)
    rng = np.random.default_rng(13)
    layers = [rand_f16_frame(rng, full, full, alpha="one"), rand_f16_frame(rng, full, (5, 3, 30, 15))]
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame(full, np.uint16)
    with pytest.raises(RuntimeError):
        chain_color_over([(out, dl)], None, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    want = orc.chain_color_over(layers, None)
    chain_color_over([(out, dl)], None)
    assert cvs.cvs_chain_last_was_fused() == 0
    got = out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.window_view(), want.window_view(), "plain stack, node by node")


@pytest.mark.parametrize("pre,post", [(-1, -1), (0, -1), (0, 2)])
@pytest.mark.parametrize("geom", [((0, 0, 63, 35), (0, 0, 63, 35), (0, 0, 63, 35)),          # whole frame: flat kernel
                                  ((0, 0, 62, 34), (0, 0, 62, 34), (0, 0, 62, 34)),          # odd pixel count
                                  ((-2, -2, 50, 30), (0, 0, 63, 35), (3, 1, 60, 33))])       # windows differ: rect kernel
def test_colour_matrix_out_of_place(cvs, orc, pre, post, geom):
    out_full, in_full, in_cur = geom
    rng = np.random.default_rng(7)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    src = rand_f16_frame(rng, in_full, in_cur)
    src.array[4, 4] = [0x8400, 0x0001, 0x7BFF, 0xC000]
    # reference sequence: video_copy_frame_f16 into the output, then the filter in place
    want = HostFrame(out_full, np.uint16)
    orc.lib().orc_copy_frame_f16(want.ref(), src.ref())
    orc.lib().orc_color_matrix_f16(want.ref(), f32p(m), None if pre < 0 else u16p(orc.transfer_table(pre)),
                                   None if post < 0 else u16p(orc.transfer_table(post)))
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(out_full, np.uint16)
    _lib.check(cvs.cvs_color_matrix_f16_to_dev(d_out.ref(), d_src.ref(), f32p(m), pre, post, None))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.window_view(), want.window_view(), "colour matrix out of place")
    assert np.array_equal(d_src.download().array, src.array)          # the source is left alone


@pytest.mark.parametrize("scur", [(0, 0, 47, 26), (5, 3, 40, 20)])
@pytest.mark.parametrize("ntaps", [9, 1, 4, 19, 27])
def test_fir_blur_f16(cvs, orc, scur, ntaps):
    rng = np.random.default_rng(52)
    full = (0, 0, 47, 26)
    src = rand_f16_frame(rng, full, scur)
    taps = synth.gaussian_taps(ntaps, 1.5)
    src32 = HostFrame(full, np.float32, orc.half_to_float(src.array), src.current_window)
    want32 = HostFrame(full, np.float32)
    orc.lib().orc_fir_blur_f32(want32.ref(), src32.ref(), f32p(taps), ntaps)
    want = HostFrame(full, np.uint16, orc.float_to_half(want32.array), want32.current_window)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(full, np.uint16)
    _lib.check(cvs.cvs_fir_blur_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), ntaps, None))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.window_view(), want.window_view(), "f16 blur")


@pytest.mark.parametrize("size", [(96, 54), (131, 37)])
def test_config5_graph_stream_against_oracle(cvs, orc, size):
    """4 sources, colour -> blur -> 4-step composite on three stream frames, every pixel, against the oracle's node-by-node run."""
    from canvas_amd.stream import GraphStream
    from tests.util import oracle_graph
    w, h = size
    g = GraphStream(w, h, ring=3)
    for frame in range(3):
        out = g.render(frame)
        _lib.check(cvs.cvs_stream_sync(None))
        want = oracle_graph(orc, GraphStream.host_inputs(w, h, frame), g.matrix, orc.transfer_table(0), None, g.taps)
        got = out.download()
        assert same_window(got.current_window, want.current_window)
        assert_same_f16(got.array, want.array, "config 5 graph, frame %d" % frame)


def _blur_over(cvs, out_full, src, taps, overlays):
    d_src = DeviceFrame.from_host(src)
    d_ov = [DeviceFrame.from_host(o) for o in overlays]
    refs = (C.POINTER(_lib.rgba_frame_f16_t) * max(len(d_ov), 1))(*[C.pointer(o.c) for o in d_ov])
    d_out = DeviceFrame(out_full, np.uint16)
    _lib.check(cvs.cvs_blur_over_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), len(taps), refs, len(d_ov), None))
    return d_out.download()


def _frame_table(frames):
    return (C.POINTER(_lib.rgba_frame_f16_t) * max(len(frames), 1))(*[C.pointer(f.c) for f in frames])


@pytest.mark.parametrize("count,ntaps,nover", [(5, 9, 3), (11, 5, 1), (3, 21, 4), (4, 10, 2)])
def test_blur_over_batch_is_the_single_calls(cvs, orc, count, ntaps, nover):
    """cvs_blur_over_f16_batch_dev: independent frames of one geometry, up to eight per launch (grid.z = frame, row segments
    sized for the whole batch).  The pixels must be those of `count` single calls, which are checked against the oracle
    (11 frames: a launch of eight and one of three; 21 taps: the long-list instances; 10 taps: no batched form, frame by frame)."""
    from tests.util import oracle_blur_over
    w, h = 150, 70
    full = (0, 0, w - 1, h - 1)
    rng = np.random.default_rng(9100 + count)
    taps = synth.gaussian_taps(ntaps | 1, 1.5)[:ntaps].copy()
    srcs = [rand_f16_frame(rng, full, full) for _ in range(count)]
    ovs = [[rand_f16_frame(rng, full, full) for _ in range(nover)] for _ in range(count)]
    d_src = [DeviceFrame.from_host(f) for f in srcs]
    d_ov = [[DeviceFrame.from_host(f) for f in fs] for fs in ovs]
    d_out = [DeviceFrame(full, np.uint16) for _ in range(count)]
    _lib.check(cvs.cvs_blur_over_f16_batch_dev(_frame_table(d_out), _frame_table(d_src), f32p(taps), ntaps,
                                               _frame_table([o for fs in d_ov for o in fs]), nover, count, None))
    for i in range(count):
        got = d_out[i].download()
        single = _blur_over(cvs, full, srcs[i], taps, ovs[i])
        assert same_window(got.current_window, single.current_window)
        assert np.array_equal(got.array, single.array), "frame %d of the batch differs from its single call" % i
    want = oracle_blur_over(orc, srcs[count - 1], taps, ovs[count - 1])
    assert_same_f16(d_out[count - 1].download().window_view(), want.window_view(), "last frame of the batch against the oracle")


def test_blur_over_batch_with_ragged_frames_and_frames_that_feed_each_other(cvs):
    """A batch whose frames do not share one geometry, and one whose second frame blurs the first frame's OUTPUT: both are
    carried out frame by frame, in order -- the results of the single calls made one after the other."""
    w, h = 96, 40
    full = (0, 0, w - 1, h - 1)
    rng = np.random.default_rng(9200)
    taps = synth.gaussian_taps(9, 1.5)
    # (a) the second frame's source covers only part of its buffer
    srcs = [rand_f16_frame(rng, full, full), rand_f16_frame(rng, full, (3, 2, 80, 30)), rand_f16_frame(rng, full, full)]
    ovs = [[rand_f16_frame(rng, full, full)] for _ in srcs]
    d_src = [DeviceFrame.from_host(f) for f in srcs]
    d_ov = [[DeviceFrame.from_host(f) for f in fs] for fs in ovs]
    d_out = [DeviceFrame(full, np.uint16) for _ in srcs]
    _lib.check(cvs.cvs_blur_over_f16_batch_dev(_frame_table(d_out), _frame_table(d_src), f32p(taps), 9, _frame_table([o for fs in d_ov for o in fs]), 1, 3, None))
    for i in range(3):
        single = _blur_over(cvs, full, srcs[i], taps, ovs[i])
        got = d_out[i].download()
        assert same_window(got.current_window, single.current_window)
        assert np.array_equal(got.window_view(), single.window_view()), i
    # (b) frame 1 reads what frame 0 writes
    a, b = DeviceFrame(full, np.uint16), DeviceFrame(full, np.uint16)
    _lib.check(cvs.cvs_blur_over_f16_batch_dev(_frame_table([a, b]), _frame_table([d_src[0], a]), f32p(taps), 9, _frame_table([d_ov[0][0], d_ov[2][0]]), 1, 2, None))
    first = _blur_over(cvs, full, srcs[0], taps, ovs[0])
    second = _blur_over(cvs, full, first, taps, ovs[2])
    assert np.array_equal(a.download().array, first.array)
    assert np.array_equal(b.download().array, second.array)


@pytest.mark.parametrize("count,ntaps", [(2, 9), (9, 5), (3, 13)])
def test_blur_lanczos_batch_is_the_single_calls(cvs, orc, count, ntaps):
    """cvs_blur_lanczos_f16_batch_dev at factor 1/2 (config 3's sweep, eight frames per launch; 13 taps: no one-sweep form,
    frame by frame): the single calls' pixels, the last frame also against the oracle."""
    sw, sh = 200, 90
    sfull, tfull = (0, 0, sw - 1, sh - 1), (0, 0, sw // 2 - 1, sh // 2 - 1)
    rng = np.random.default_rng(9300 + count)
    taps = synth.gaussian_taps(ntaps, 1.5)
    srcs = [rand_f16_frame(rng, sfull, sfull) for _ in range(count)]
    d_src = [DeviceFrame.from_host(f) for f in srcs]
    d_out = [DeviceFrame(tfull, np.uint16) for _ in range(count)]
    _lib.check(cvs.cvs_blur_lanczos_f16_batch_dev(_frame_table(d_out), _frame_table(d_src), count, f32p(taps), ntaps, C.c_float(0.5), C.c_float(0.5), 3, None))
    if ntaps <= 11:
        assert cvs.cvs_fir_last_kernel() == _lib.FIR_KERNEL_HALVE
    for i in range(count):
        one = DeviceFrame(tfull, np.uint16)
        _lib.check(cvs.cvs_blur_lanczos_f16_dev(one.ref(), d_src[i].ref(), f32p(taps), ntaps, C.c_float(0.5), C.c_float(0.5), 3, None))
        assert np.array_equal(d_out[i].download().array, one.download().array), i
    want = _oracle_config3(orc, srcs[count - 1], (sw // 2, sh // 2), taps, 0.5, 0.5)
    assert_same_f16(d_out[count - 1].download().array, want.array, "last frame of the batch against the oracle")


@pytest.mark.parametrize("nover", [1, 3, 4])
@pytest.mark.parametrize("ntaps", [9, 3, 15, 23])
@pytest.mark.parametrize("size", [(300, 41), (64, 36)])         # two strips / one narrow strip
def test_blur_over_fused(cvs, orc, nover, ntaps, size):
    """Blur node as the lowest workspace item, f16 layers above it, f16 pull: one launch."""
    from tests.util import oracle_blur_over
    from canvas_amd.synth import truncate_to_half
    w, h = size
    full = (0, 0, w - 1, h - 1)
    rng = np.random.default_rng(700 + nover + ntaps)
    src = rand_f16_frame(rng, full, full)
    overlays = [HostFrame(full, np.uint16, truncate_to_half(rand_f32_frame(rng, full, full, alpha="mixed", lo=-0.25, hi=1.5).array))
                for _ in range(nover)]
    overlays[0].array[2, 3] = [0x7BFF, 0xFBFF, 0x7BFF, 0x3800]
    taps = synth.gaussian_taps(ntaps, 1.5)
    want = oracle_blur_over(orc, src, taps, overlays)
    got = _blur_over(cvs, full, src, taps, overlays)
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.array, want.array, "blur+over fused")


@pytest.fixture
def blur_columns():
    """Pins the register-window blur to its one-column-per-lane or its two-columns-per-lane form (cvs_fir_path_override)."""
    lib = _lib.load()

    def pin(columns):
        lib.cvs_fir_path_override({1: _lib.FIR_PATH_ONE_COLUMN, 2: _lib.FIR_PATH_TWO_COLUMNS, None: _lib.FIR_PATH_AUTO}[columns])
    yield pin
    pin(None)


def _blur_kernel_seen(cvs):
    assert _lib.last_error() == "", _lib.last_error()
    return _KERNEL_NAMES[cvs.cvs_fir_last_kernel()]


# (full frame, source window, does the two-column form take it?)
_PAIR_GEOMETRIES = [
    ((0, 0, 299, 40), None, True),                    # three 120-column strips, the last one partly filled
    ((0, 0, 61, 19), None, True),                     # one narrow strip
    ((0, 0, 129, 39), (4, 3, 121, 35), True),         # a source window inside its buffer, on pair boundaries: skipped taps on all four sides
    ((0, 0, 129, 39), (5, 3, 120, 35), False),        # ... starting on an odd column: pairs would straddle the window's edge
    ((0, 0, 130, 39), None, False),                   # an odd width
    ((-6, -3, 123, 30), (-6, -3, 123, 30), True),     # negative coordinates
]


@pytest.mark.parametrize("full,scur,pairs", _PAIR_GEOMETRIES)
@pytest.mark.parametrize("ntaps", [3, 5, 7, 9, 13, 15])
@pytest.mark.parametrize("nover", [0, 1, 3, 4])
def test_blur_two_columns_per_lane_is_the_one_column_blur(cvs, orc, blur_columns, full, scur, pairs, ntaps, nover):
    """k_blur_pair (two neighbouring columns per lane, buffer loads with range-checked rows, blends on pixel pairs) against
    the oracle and, code for code, against k_blur on the same frames; launches it cannot take (pairs that would straddle
    an edge, more than 13 taps) must say so and run on k_blur."""
    from tests.util import oracle_blur_over
    from canvas_amd.synth import truncate_to_half
    rng = np.random.default_rng(9300 + ntaps * 10 + nover)
    src = rand_f16_frame(rng, full, scur or full)
    overlays = [HostFrame(full, np.uint16, truncate_to_half(rand_f32_frame(rng, full, full, alpha="mixed", lo=-0.25, hi=1.5).array))
                for _ in range(nover)]
    taps = synth.gaussian_taps(ntaps, 1.5)
    want = oracle_blur_over(orc, src, taps, overlays)
    cvs.cvs_clear_last_error()
    blur_columns(2)
    got2 = _blur_over(cvs, full, src, taps, overlays)
    # (layers over a blur whose source does not cover the frame go node by node: the blur then writes an f32 frame, on k_blur)
    fused = nover == 0 or scur is None or scur == full
    assert _blur_kernel_seen(cvs) == ("window-pair" if pairs and ntaps <= 13 and fused else "window")
    blur_columns(1)
    got1 = _blur_over(cvs, full, src, taps, overlays)
    assert _blur_kernel_seen(cvs) == "window"
    assert same_window(got2.current_window, want.current_window) and same_window(got1.current_window, want.current_window)
    assert_same_f16(got2.window_view(), want.window_view(), "two columns per lane against the oracle")
    assert_same_f16(got2.array, got1.array, "two columns per lane against one")


def test_blur_two_columns_per_lane_with_special_values(cvs, orc, blur_columns):
    """Zeros (whole black regions: the blend's fast reciprocal path must hand over to the IEEE divide), exact alpha 0 and 1,
    denormals, the largest halfs (sums beyond the half range must come out as infinities), Inf and NaN."""
    from tests.util import oracle_blur_over
    rng = np.random.default_rng(9400)
    full = (0, 0, 259, 47)
    src = rand_f16_frame(rng, full, full)
    src.array[:, 40:90] = 0                                   # black AND transparent
    src.array[10:30, 100:140, 3] = 0x3C00                     # opaque
    src.array[5:9, 150:170, :3] = 0x7BFF                      # 65504: nine of them overflow the half range
    src.array[20, 200] = [0x7C00, 0xFC00, 0x7E00, 0x3C00]     # Inf, -Inf, NaN
    src.array[31, 7] = [0x0001, 0x8001, 0x03FF, 0x0001]       # denormals
    overlays = [rand_f16_frame(rng, full, full) for _ in range(3)]
    overlays[0].array[:, 60:120, 3] = 0                       # transparent over black: 0 / 0
    overlays[1].array[:, 100:160, 3] = 0x3C00
    overlays[1].array[12:20, 10:30] = 0
    overlays[2].array[25, 50] = [0x7BFF, 0xFBFF, 0x7C00, 0x3800]
    overlays[2].array[26, 51] = [0x0001, 0x0002, 0x8003, 0x0001]
    taps = synth.gaussian_taps(9, 1.5)
    want = oracle_blur_over(orc, src, taps, overlays)
    for nover in (3, 1, 0):
        want = oracle_blur_over(orc, src, taps, overlays[:nover])
        cvs.cvs_clear_last_error()
        blur_columns(2)
        got = _blur_over(cvs, full, src, taps, overlays[:nover])
        assert _blur_kernel_seen(cvs) == "window-pair"
        assert_same_f16(got.array, want.array, "special values, %d layers" % nover)
        blur_columns(1)
        assert_same_f16(_blur_over(cvs, full, src, taps, overlays[:nover]).array, got.array, "special values, %d layers, against one column per lane" % nover)


def test_blur_over_batch_on_two_columns_per_lane(cvs, orc, blur_columns):
    """The batched launch (grid.z = frame) of the two-column form: five frames, each the single call's pixels."""
    from tests.util import oracle_blur_over
    w, h, count, nover = 244, 37, 5, 3
    full = (0, 0, w - 1, h - 1)
    rng = np.random.default_rng(9500)
    taps = synth.gaussian_taps(9, 1.5)
    srcs = [rand_f16_frame(rng, full, full) for _ in range(count)]
    ovs = [[rand_f16_frame(rng, full, full) for _ in range(nover)] for _ in range(count)]
    d_src = [DeviceFrame.from_host(f) for f in srcs]
    d_ov = [[DeviceFrame.from_host(f) for f in fs] for fs in ovs]
    d_out = [DeviceFrame(full, np.uint16) for _ in range(count)]
    cvs.cvs_clear_last_error()
    blur_columns(2)
    _lib.check(cvs.cvs_blur_over_f16_batch_dev(_frame_table(d_out), _frame_table(d_src), f32p(taps), 9,
                                               _frame_table([o for fs in d_ov for o in fs]), nover, count, None))
    assert _blur_kernel_seen(cvs) == "window-pair"
    for i in range(count):
        want = oracle_blur_over(orc, srcs[i], taps, ovs[i])
        assert_same_f16(d_out[i].download().array, want.array, "frame %d of the batch" % i)


@pytest.mark.parametrize("case", ["ragged", "even_taps", "five_layers", "no_layers", "small_source"])
def test_blur_over_node_by_node(cvs, orc, case):
    from tests.util import oracle_blur_over
    full = (0, 0, 59, 33)
    rng = np.random.default_rng(800)
    src_cur = (4, 2, 50, 30) if case == "small_source" else full
    src = rand_f16_frame(rng, full, src_cur)
    nover = {"five_layers": 5, "no_layers": 0}.get(case, 2)
    # windows: see test_chain_ragged_windows for why the upper layers sit where they do
    wins = [(5, 3, 40, 25), (20, 2, 59, 20)] if case == "ragged" else [full] * nover
    overlays = [rand_f16_frame(rng, full, wins[k]) for k in range(nover)]
    taps = synth.gaussian_taps(4 if case == "even_taps" else 9, 1.5)
    want = oracle_blur_over(orc, src, taps, overlays)
    got = _blur_over(cvs, full, src, taps, overlays)
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.window_view(), want.window_view(), "blur+over %s" % case)


@pytest.mark.parametrize("kernel", ["hv", "passes", "tiled"])
@pytest.mark.parametrize("ntaps", [9, 10])
def test_blur_over_node_by_node_through_the_table_kernels(cvs, orc, force_fir, kernel, ntaps):
    """The node-by-node form blurs an f16 source into an f32 frame; with the register-window kernels out of the way that is
    the table kernels' f16-in / f32-out form, which no other entry reaches."""
    from tests.util import oracle_blur_over
    full = (0, 0, 59, 33)
    rng = np.random.default_rng(810 + ntaps)
    src = rand_f16_frame(rng, full, full)
    # windows: see test_chain_ragged_windows for why the upper layers sit where they do
    overlays = [rand_f16_frame(rng, full, w) for w in [(5, 3, 40, 25), (20, 2, 59, 20)]]
    taps = synth.gaussian_taps(ntaps | 1, 1.5)[:ntaps].copy()
    want = oracle_blur_over(orc, src, taps, overlays)
    force_fir(kernel)
    cvs.cvs_clear_last_error()
    got = _blur_over(cvs, full, src, taps, overlays)
    ran_on(cvs, kernel, _FALLBACK.get((kernel, ntaps)), note="blur over %s %d" % (kernel, ntaps))
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.window_view(), want.window_view(), "blur+over node by node (%s)" % kernel)


# ------------------------------------------------------------------ display / export edge (survey N2)

def _all_codes_frame(full, cur):
    """Every half code appears in every channel position at least once."""
    f = HostFrame(full, np.uint16, current_window=cur)
    h, w = f.array.shape[:2]
    n = h * w * 4
    codes = (np.arange(n, dtype=np.uint64) * 40503 % 65536).astype(np.uint16)      # odd multiplier: a permutation, repeated
    f.array[...] = codes.reshape(h, w, 4)
    return f


@pytest.mark.parametrize("mode", [_lib.DISPLAY_RGBA8, _lib.DISPLAY_ARGB32_PREMUL])
@pytest.mark.parametrize("pre", [_lib.LUT_NONE, _lib.LUT_LINEAR_TO_SRGB])
@pytest.mark.parametrize("geom", [((0, 0, 255, 71), (0, 0, 255, 71)),        # whole frame: pair kernel
                                  ((0, 0, 254, 70), (0, 0, 254, 70)),        # odd pixel count
                                  ((-3, -2, 200, 90), (5, 1, 150, 77)),      # a window inside the buffer
                                  ((0, 0, 9, 9), (4, 4, 4, 4))])             # one pixel
def test_frame_to_bytes(cvs, orc, mode, pre, geom):
    full, cur = geom
    frame = _all_codes_frame(full, cur)
    w, h = cur[2] - cur[0] + 1, cur[3] - cur[1] + 1
    want = np.zeros((h, w), np.uint32)
    table = None if pre == _lib.LUT_NONE else orc.transfer_table(pre)
    orc.lib().orc_frame_to_bytes(want.ctypes.data_as(C.POINTER(C.c_uint32)), frame.ref(), None if table is None else u16p(table), mode)
    # device frame -> device bytes
    dev = DeviceFrame.from_host(frame)
    out = cvs.cvs_malloc(w * h * 4)
    try:
        _lib.check(cvs.cvs_frame_to_bytes_dev(out, dev.ref(), pre, mode, None))
        got = np.zeros((h, w), np.uint32)
        _lib.check(cvs.cvs_memcpy_d2h(got.ctypes.data, out, w * h * 4, None))
    finally:
        cvs.cvs_free(out)
    assert np.array_equal(got, want)
    # host frame -> host bytes
    got2 = np.zeros((h, w), np.uint32)
    _lib.check(cvs.video_frame_to_bytes(got2.ctypes.data, frame.ref(), pre, mode))
    assert np.array_equal(got2, want)


@pytest.mark.parametrize("intent", [1.25, 1.0, 0.7])
@pytest.mark.parametrize("pre", [_lib.LUT_NONE, _lib.LUT_LINEAR_TO_SRGB])
def test_frame_to_rgba8_with_the_widget_ramp(cvs, orc, pre, intent):
    """widget_gl.c:291-307: transfer table, then the ramp lrint(clamp(x^intent * 255)) -- every half code, several intents
    (each (table, intent) pair is its own cached byte table; more pairs than cache slots are exercised across the suite)."""
    full, cur = (-3, -2, 200, 90), (5, 1, 150, 77)
    frame = _all_codes_frame(full, cur)
    w, h = cur[2] - cur[0] + 1, cur[3] - cur[1] + 1
    want = np.zeros((h, w), np.uint32)
    table = None if pre == _lib.LUT_NONE else orc.transfer_table(pre)
    orc.lib().orc_frame_to_rgba8_intent(want.ctypes.data_as(C.POINTER(C.c_uint32)), frame.ref(), None if table is None else u16p(table), C.c_float(intent))
    dev = DeviceFrame.from_host(frame)
    out = cvs.cvs_malloc(w * h * 4)
    try:
        _lib.check(cvs.cvs_frame_to_rgba8_intent_dev(out, dev.ref(), pre, C.c_float(intent), None))
        got = np.zeros((h, w), np.uint32)
        _lib.check(cvs.cvs_memcpy_d2h(got.ctypes.data, out, w * h * 4, None))
    finally:
        cvs.cvs_free(out)
    assert np.array_equal(got, want)
    got2 = np.zeros((h, w), np.uint32)
    _lib.check(cvs.video_frame_to_rgba8_intent(got2.ctypes.data, frame.ref(), pre, C.c_float(intent)))
    assert np.array_equal(got2, want)


def test_display_table_cache_evicts_and_rebuilds(cvs, orc):
    """More (table, ramp) pairs than the cache holds, revisited: an evicted pair must come back right."""
    frame = _all_codes_frame((0, 0, 63, 63), (0, 0, 63, 63))
    intents = [0.5 + 0.1 * k for k in range(12)]
    for intent in intents + intents[:3]:
        want = np.zeros((64, 64), np.uint32)
        got = np.zeros((64, 64), np.uint32)
        orc.lib().orc_frame_to_rgba8_intent(want.ctypes.data_as(C.POINTER(C.c_uint32)), frame.ref(), None, C.c_float(intent))
        _lib.check(cvs.video_frame_to_rgba8_intent(got.ctypes.data, frame.ref(), _lib.LUT_NONE, C.c_float(intent)))
        assert np.array_equal(got, want), intent
    _lib.check(cvs.video_frame_to_bytes(got.ctypes.data, frame.ref(), _lib.LUT_NONE, _lib.DISPLAY_RGBA8))
    orc.lib().orc_frame_to_bytes(want.ctypes.data_as(C.POINTER(C.c_uint32)), frame.ref(), None, 0)
    assert np.array_equal(got, want)


def test_frame_to_bytes_follows_an_installed_table(cvs, orc):
    """cvs_lut_install replaces a transfer table: the byte table composed from it must follow."""
    frame = _all_codes_frame((0, 0, 63, 63), (0, 0, 63, 63))
    original = orc.transfer_table(_lib.LUT_LINEAR_TO_SRGB)
    swapped = orc.transfer_table(_lib.LUT_LINEAR_TO_REC709)
    got = np.zeros((64, 64), np.uint32)
    want = np.zeros((64, 64), np.uint32)
    try:
        _lib.check(cvs.video_frame_to_bytes(got.ctypes.data, frame.ref(), _lib.LUT_LINEAR_TO_SRGB, _lib.DISPLAY_RGBA8))
        _lib.check(cvs.cvs_lut_install(_lib.LUT_LINEAR_TO_SRGB, u16p(swapped)))
        _lib.check(cvs.video_frame_to_bytes(got.ctypes.data, frame.ref(), _lib.LUT_LINEAR_TO_SRGB, _lib.DISPLAY_RGBA8))
        orc.lib().orc_frame_to_bytes(want.ctypes.data_as(C.POINTER(C.c_uint32)), frame.ref(), u16p(swapped), 0)
        assert np.array_equal(got, want)
    finally:
        _lib.check(cvs.cvs_lut_install(_lib.LUT_LINEAR_TO_SRGB, u16p(original)))
    empty = HostFrame((0, 0, 3, 3), np.uint16, current_window=(0, 0, -1, -1))
    assert cvs.video_frame_to_bytes(got.ctypes.data, empty.ref(), _lib.LUT_NONE, _lib.DISPLAY_RGBA8) == 0
    assert cvs.video_frame_to_bytes(got.ctypes.data, frame.ref(), _lib.LUT_NONE, 7) != 0


# ------------------------------------------------------------------ DV 4:1:1 edge (survey N3)

DV_STRIDES = (720, 180, 180)


def _dv_planes(rng):
    return [np.ascontiguousarray(rng.integers(0, 256, (480, s), dtype=np.uint8)) for s in DV_STRIDES]


def _oracle_reconstruct(orc, full, planes):
    want = HostFrame(full, np.uint16, fill=0x1234)
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    strides = (C.c_int * 3)(*DV_STRIDES)
    orc.lib().orc_reconstruct_dv(want.ref(), ptrs, strides)
    return want


@pytest.mark.parametrize("full", [(0, -1, 719, 478),          # the whole raster
                                  (-8, -5, 730, 490),         # buffer larger than the raster
                                  (101, 50, 333, 99),         # a window inside it, not on a chroma boundary
                                  (715, 470, 800, 500),       # the bottom-right corner
                                  (800, 0, 900, 10)])         # beside the raster: empty
def test_dv_reconstruct(cvs, orc, full):
    rng = np.random.default_rng(901)
    planes = _dv_planes(rng)
    planes[0][7, :16] = [0, 16, 235, 255] * 4                 # studio-range edges
    planes[1][7, :4] = [0, 16, 240, 255]
    want = _oracle_reconstruct(orc, full, planes)
    img = _lib.coded_image()
    for p in range(3):
        img.data[p], img.stride[p], img.line_count[p] = planes[p].ctypes.data, DV_STRIDES[p], 480
    got = HostFrame(full, np.uint16, fill=0x1234)
    cvs.video_reconstruct_dv(got.ref(), C.byref(img))
    assert same_window(got.current_window, want.current_window)
    if not want.current_window.is_empty():
        assert_same_f16(got.window_view(), want.window_view(), "DV reconstruct %r" % (full,))


@pytest.mark.parametrize("full,cur", [((0, -1, 719, 478), (0, -1, 719, 478)),
                                      ((-10, -10, 800, 500), (-10, -10, 800, 500)),      # frame larger than the raster
                                      ((0, -1, 719, 478), (37, 20, 601, 300)),           # window not on chroma boundaries
                                      ((0, -1, 719, 478), (5, 5, 5, 5)),
                                      ((0, -1, 719, 478), (0, 0, -1, -1))])              # nothing defined: all-zero planes
def test_dv_subsample(cvs, orc, full, cur):
    rng = np.random.default_rng(902)
    frame = rand_f16_frame(rng, full, cur)
    lo, hi = rand_f32_frame(rng, full, cur, lo=-0.5, hi=2.0), None
    from canvas_amd.synth import truncate_to_half
    frame.array[::3] = truncate_to_half(lo.array[::3])        # some out-of-range rows: bytes wrap like the reference's casts
    theirs = frame.copy()
    want = [np.full((480, s), 0xAA, np.uint8) for s in DV_STRIDES]
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in want])
    orc.lib().orc_subsample_dv(ptrs, (C.c_int * 3)(*DV_STRIDES), theirs.ref())
    mine = frame.copy()
    img = cvs.video_subsample_dv(mine.ref())
    assert img
    try:
        for p in range(3):
            assert (img.contents.stride[p], img.contents.line_count[p]) == (DV_STRIDES[p], 480)
            got = np.ctypeslib.as_array(C.cast(img.contents.data[p], C.POINTER(C.c_uint8)), shape=(480, DV_STRIDES[p]))
            assert np.array_equal(got, want[p]), "plane %d" % p
    finally:
        C.CFUNCTYPE(None, C.c_void_p)(img.contents.free_func)(C.cast(img, C.c_void_p))
    # the rows that were read are left transfer-encoded in the caller's frame (video_subsample.c:144)
    if not frame.current_window.is_empty():
        assert np.array_equal(mine.window_view(), theirs.window_view())


def test_dv_device_round_trip_and_untouched_input(cvs, orc):
    """Device twins: planes -> frame -> planes on the device; with encode_input_in_place = 0 the frame is left alone."""
    rng = np.random.default_rng(903)
    planes = _dv_planes(rng)
    full = (0, -1, 719, 478)
    want_frame = _oracle_reconstruct(orc, full, planes)
    sizes = [480 * s for s in DV_STRIDES]
    dev = [cvs.cvs_malloc(n) for n in sizes]
    out = [cvs.cvs_malloc(n) for n in sizes]
    try:
        img, img2 = _lib.coded_image(), _lib.coded_image()
        for p in range(3):
            _lib.check(cvs.cvs_memcpy_h2d(dev[p], planes[p].ctypes.data, sizes[p], None))
            img.data[p], img.stride[p], img.line_count[p] = dev[p], DV_STRIDES[p], 480
            img2.data[p], img2.stride[p], img2.line_count[p] = out[p], DV_STRIDES[p], 480
        d = DeviceFrame(full, np.uint16)
        _lib.check(cvs.cvs_reconstruct_dv_dev(d.ref(), C.byref(img), None))
        got = d.download()
        assert_same_f16(got.array, want_frame.array, "DV reconstruct on device")
        _lib.check(cvs.cvs_subsample_dv_dev(C.byref(img2), d.ref(), 0, None))
        assert np.array_equal(d.download().array, got.array)
        theirs = want_frame.copy()
        want = [np.zeros((480, s), np.uint8) for s in DV_STRIDES]
        orc.lib().orc_subsample_dv((C.c_void_p * 3)(*[p.ctypes.data for p in want]), (C.c_int * 3)(*DV_STRIDES), theirs.ref())
        for p in range(3):
            back = np.zeros((480, DV_STRIDES[p]), np.uint8)
            _lib.check(cvs.cvs_memcpy_d2h(back.ctypes.data, out[p], sizes[p], None))
            assert np.array_equal(back, want[p])
    finally:
        for p in dev + out:
            cvs.cvs_free(p)


@pytest.mark.parametrize("ksize", [1, 2, 3, 4, 5])                       # 5: 19 taps, no register-window instance -> gather kernel
@pytest.mark.parametrize("geom", [((0, 0, 1099, 39), (0, 0, 1099, 39), (0, 0, 549, 19)),         # three strips wide
                                  ((-7, -3, 200, 90), (3, 5, 180, 77), (-4, -2, 110, 50)),       # windows with origins, source window inside its buffer
                                  ((0, 0, 63, 35), (0, 0, 63, 35), (0, 0, 40, 30))])             # target reaches past the source
def test_lanczos_halving_uniform_taps(cvs, orc, ksize, geom):
    sfull, scur, tfull = geom
    rng = np.random.default_rng(62 + ksize)
    src = rand_f32_frame(rng, sfull, scur, lo=-0.5, hi=1.5)
    want = HostFrame(tfull, np.float32)
    orc.lib().orc_resample_lanczos_f32(want.ref(), src.ref(), C.c_float(0.5), C.c_float(0.5), ksize)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.float32)
    _lib.check(cvs.cvs_resample_lanczos_f32_dev(d_out.ref(), d_src.ref(), C.c_float(0.5), C.c_float(0.5), ksize, None))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f32(got.array, want.array, "lanczos halving, kernel size %d" % ksize)


# ------------------------------------------------------------------ fused f16 crossfade

def _oracle_cross_f16(orc, out_full, a, b, mix):
    """widen (clipped to the output buffer, as a pull into it would be) -> video_mix_cross_f32 -> truncate"""
    def pulled(frame):
        clip = HostFrame(out_full, np.uint16)
        orc.lib().orc_copy_frame_f16(clip.ref(), frame.ref())
        return HostFrame(out_full, np.float32, orc.half_to_float(clip.array), clip.current_window)
    pa, pb = pulled(a), pulled(b)
    out = HostFrame(out_full, np.float32)
    orc.lib().orc_mix_cross_f32(out.ref(), pa.ref(), pb.ref(), C.c_float(mix))
    return HostFrame(out_full, np.uint16, orc.float_to_half(out.array), out.current_window)


@pytest.mark.parametrize("mix", [0.3, 0.5, 0.0, 1.0, 0.999])
@pytest.mark.parametrize("size", [(96, 54), (33, 7), (1, 1)])
def test_mix_cross_f16_fused(cvs, orc, mix, size):
    from canvas_amd.synth import truncate_to_half
    w, h = size
    full = (0, 0, w - 1, h - 1)
    rng = np.random.default_rng(1200 + w)
    a = HostFrame(full, np.uint16, truncate_to_half(rand_f32_frame(rng, full, full, alpha="mixed", lo=-0.25, hi=1.5).array))
    b = HostFrame(full, np.uint16, truncate_to_half(rand_f32_frame(rng, full, full, alpha="mixed", lo=-0.25, hi=1.5).array))
    if w > 8:
        a.array[1, 2] = [0x7BFF, 0xFBFF, 0x0001, 0x3C00]
        b.array[1, 2] = [0x7BFF, 0x7BFF, 0x8001, 0x3C00]
    want = _oracle_cross_f16(orc, full, a, b, mix)
    da, db, out = DeviceFrame.from_host(a), DeviceFrame.from_host(b), DeviceFrame(full, np.uint16)
    _lib.check(cvs.cvs_mix_cross_f16_dev(out.ref(), da.ref(), db.ref(), C.c_float(mix), None))
    assert cvs.cvs_chain_last_was_fused() == (1 if w * h >= 2 else 0)
    got = out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.array, want.array, "fused crossfade, mix %g" % mix)


@pytest.mark.parametrize("pw,qw", MIX_WINDOWS[:6])
def test_mix_cross_f16_windowed_goes_node_by_node(cvs, orc, pw, qw):
    rng = np.random.default_rng(1300)
    full = (0, 0, 23, 11)
    a, b = rand_f16_frame(rng, full, pw), rand_f16_frame(rng, full, qw)
    want = _oracle_cross_f16(orc, full, a, b, 0.35)
    da, db, out = DeviceFrame.from_host(a), DeviceFrame.from_host(b), DeviceFrame(full, np.uint16)
    _lib.check(cvs.cvs_mix_cross_f16_dev(out.ref(), da.ref(), db.ref(), C.c_float(0.35), None))
    got = out.download()
    assert same_window(got.current_window, want.current_window)
    if not want.current_window.is_empty():
        assert_same_f16(got.window_view(), want.window_view(), "windowed crossfade %r %r" % (pw, qw))


def test_frames_whose_window_leaves_their_buffer_are_refused(cvs):
    """A current_window reaching outside full_window would send a kernel out of bounds: every device entry point
    refuses such an input (status, message, empty output window) instead of launching."""
    full = (0, 0, 31, 17)
    good16, good32 = DeviceFrame(full, np.uint16), DeviceFrame(full, np.float32)
    bad16, bad32 = DeviceFrame(full, np.uint16, current_window=(0, 0, 40, 17)), DeviceFrame(full, np.float32, current_window=(-1, 0, 31, 17))
    out16, out32 = DeviceFrame(full, np.uint16), DeviceFrame(full, np.float32)
    taps = synth.gaussian_taps(9, 1.5)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    calls = [
        lambda: cvs.cvs_copy_frame_f16_dev(out16.ref(), bad16.ref(), None),
        lambda: cvs.cvs_copy_frame_alpha_f32_dev(out32.ref(), bad32.ref(), C.c_float(0.5), None),
        lambda: cvs.cvs_frame_f16_to_f32_dev(out32.ref(), bad16.ref(), None),
        lambda: cvs.cvs_frame_f32_to_f16_dev(out16.ref(), bad32.ref(), None),
        lambda: cvs.cvs_mix_cross_f32_dev(out32.ref(), good32.ref(), bad32.ref(), C.c_float(0.5), None),
        lambda: cvs.cvs_mix_over_f32_dev(out32.ref(), bad32.ref(), C.c_float(1.0), None),
        lambda: cvs.cvs_gain_offset_f16_dev(out16.ref(), bad16.ref(), C.c_float(1.0), C.c_float(0.0), None),
        lambda: cvs.cvs_color_matrix_f16_to_dev(out16.ref(), bad16.ref(), f32p(m), -1, -1, None),
        lambda: cvs.cvs_mix_cross_f16_dev(out16.ref(), bad16.ref(), good16.ref(), C.c_float(0.5), None),
        lambda: cvs.cvs_scale_bilinear_f32_dev(out32.ref(), v2f(0, 0), bad32.ref(), v2f(0, 0), v2f(0.5, 0.5), None),
        lambda: cvs.cvs_fir_blur_f32_dev(out32.ref(), bad32.ref(), f32p(taps), 9, None),
        lambda: cvs.cvs_fir_blur_f16_dev(out16.ref(), bad16.ref(), f32p(taps), 9, None),
        lambda: cvs.cvs_resample_lanczos_f32_dev(out32.ref(), bad32.ref(), C.c_float(0.5), C.c_float(0.5), 3, None),
        lambda: cvs.cvs_blur_lanczos_f16_dev(out16.ref(), bad16.ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None),
    ]
    for i, call in enumerate(calls):
        out16.c.current_window = box2i.of(*full)
        out32.c.current_window = box2i.of(*full)
        assert call() != 0, i
        assert "outside its buffer" in _lib.last_error(), (i, _lib.last_error())
        assert out16.current_window.is_empty() or out32.current_window.is_empty(), i
    _lib.check(cvs.cvs_stream_sync(None))


# ------------------------------------------------------------------ the other BASELINE configs at full size, every pixel

def test_config3_full_4k_every_pixel(cvs, orc):
    """3840x2160 f16 -> 9-tap Gaussian -> Lanczos3 -> 1920x1080 f16, whole frame against the oracle's nodes."""
    w, h = 3840, 2160
    taps = synth.gaussian_taps(9, 1.5)
    layer = synth.layer_frame(w, h, 1, 0)
    want = _oracle_config3(orc, layer, (w // 2, h // 2), taps, 0.5, 0.5)
    d_src = DeviceFrame.from_host(layer)
    d_out = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
    _lib.check(cvs.cvs_blur_lanczos_f16_dev(d_out.ref(), d_src.ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.array, want.array, "4K config 3, every pixel")


def test_config5_full_4k_every_pixel(cvs, orc):
    """One 3840x2160 frame of the 10-node graph (colour -> blur -> 4-step composite), whole frame against the oracle."""
    from canvas_amd.stream import GraphStream
    from tests.util import oracle_graph
    w, h = 3840, 2160
    g = GraphStream(w, h, ring=1)
    out = g.render(0)
    _lib.check(cvs.cvs_stream_sync(None))
    want = oracle_graph(orc, GraphStream.host_inputs(w, h, 0), g.matrix, orc.transfer_table(0), None, g.taps)
    got = out.download()
    assert same_window(got.current_window, want.current_window)
    assert_same_f16(got.array, want.array, "4K config 5, every pixel")


def test_config4_8k_rows_against_oracle(cvs, orc):
    """7680x4320 3-layer over stack: the first and the last 64 rows of the frame against the oracle (the kernel is a
    flat stream of pixel pairs: both ends of the buffer, 983 040 pixels, every layer translucent above the base)."""
    w, h, rows = 7680, 4320, 64
    full = (0, 0, w - 1, h - 1)
    arrays = [synth.layer_pixels(w, h, k, 0) for k in range(3)]
    dl = []
    for a in arrays:
        d = DeviceFrame(full, np.uint16)
        d.upload(a)
        dl.append(d)
    out = DeviceFrame(full, np.uint16)
    chain_color_over([(out, dl)], None)
    _lib.check(cvs.cvs_stream_sync(None))
    assert cvs.cvs_chain_last_was_fused() == 1
    got = out.download().array
    for sl in (slice(0, rows), slice(h - rows, h)):
        part = [HostFrame((0, 0, w - 1, rows - 1), np.uint16, a[sl]) for a in arrays]
        want = orc.chain_color_over(part, None)
        assert_same_f16(got[sl], want.array, "8K config 4 rows %r" % (sl,))


def test_concurrent_callers_each_get_their_own_results(cvs, orc):
    """SURVEY 8b threading: pulls arrive from any thread, concurrently.  Six threads (each on the library's per-thread
    stream) hammer different entry points with their own frames -- shared state under them: the scratch pool, the tap
    table cache, the transfer and byte tables -- and every result must still equal the oracle's."""
    import threading
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    taps = synth.gaussian_taps(9, 1.5)
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(9000 + seed)
            w, h = 64 + 8 * seed, 36 + 2 * seed
            full = (0, 0, w - 1, h - 1)
            for it in range(6):
                layers = [synth.layer_frame(w, h, k, seed * 10 + it) for k in range(3)]
                want_chain = orc.chain_color_over(layers, m, orc.transfer_table(0), None)
                dl = [DeviceFrame.from_host(l) for l in layers]
                out = DeviceFrame(full, np.uint16)
                chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
                got = out.download()
                assert np.array_equal(got.array, want_chain.array), "chain"
                # blur + halving resample (two launches, pooled f32 intermediate, cached tables)
                want3 = _oracle_config3(orc, layers[1], (w // 2, h // 2), taps, 0.5, 0.5)
                small = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
                _lib.check(cvs.cvs_blur_lanczos_f16_dev(small.ref(), dl[1].ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
                assert_same_f16(small.download().array, want3.array, "config 3")
                # triangle scaler on f32 frames
                src = rand_f32_frame(rng, full)
                t_full = (0, 0, 2 * w - 1, 2 * h - 1)
                want_s = HostFrame(t_full, np.float32)
                orc.lib().orc_scale_bilinear_f32(want_s.ref(), v2f(0, 0), src.ref(), v2f(0, 0), v2f(2.0, 2.0))
                d_src, d_up = DeviceFrame.from_host(src), DeviceFrame(t_full, np.float32)
                _lib.check(cvs.cvs_scale_bilinear_f32_dev(d_up.ref(), v2f(0, 0), d_src.ref(), v2f(0, 0), v2f(2.0, 2.0), None))
                got_s = d_up.download()
                assert same_window(got_s.current_window, want_s.current_window)
                assert_same_f32(got_s.window_view(), want_s.window_view(), "scale")
                # display bytes
                packed = np.zeros((h, w), np.uint32)
                want_b = np.zeros((h, w), np.uint32)
                _lib.check(cvs.video_frame_to_bytes(packed.ctypes.data, layers[2].ref(), _lib.LUT_LINEAR_TO_SRGB, _lib.DISPLAY_RGBA8))
                orc.lib().orc_frame_to_bytes(want_b.ctypes.data_as(C.POINTER(C.c_uint32)), layers[2].ref(), u16p(orc.transfer_table(3)), 0)
                assert np.array_equal(packed, want_b), "bytes"
        except Exception as e:          # noqa: BLE001 -- reported on the main thread
            errors.append("thread %d: %r" % (seed, e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors


def test_animated_zoom_from_several_threads_churns_the_table_cache(cvs, orc):
    """Four threads, each scaling with a factor that changes every call (an animated zoom): far more distinct tap tables
    than the cache holds, so entries are evicted all the time while other threads are between looking a table up and
    enqueueing the launch that reads it.  A table in that state is pinned; every result must equal the oracle's."""
    import threading
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(9500 + seed)
            full = (0, 0, 47, 26)
            src = rand_f32_frame(rng, full)
            d_src = DeviceFrame.from_host(src)
            for it in range(30):
                fx, fy = 0.6 + 0.013 * (4 * it + seed), 1.4 - 0.011 * (4 * it + seed)
                t_full = (0, 0, 63, 35)
                want = HostFrame(t_full, np.float32)
                orc.lib().orc_scale_bilinear_f32(want.ref(), v2f(0, 0), src.ref(), v2f(0, 0), v2f(fx, fy))
                d_out = DeviceFrame(t_full, np.float32)
                _lib.check(cvs.cvs_scale_bilinear_f32_dev(d_out.ref(), v2f(0, 0), d_src.ref(), v2f(0, 0), v2f(fx, fy), None))
                got = d_out.download()
                assert same_window(got.current_window, want.current_window), (fx, fy)
                assert_same_f32(got.window_view(), want.window_view(), "zoom %r" % ((fx, fy),))
                # and a resample whose tables come from the same cache
                small = DeviceFrame((0, 0, 23, 13), np.float32)
                want_l = HostFrame((0, 0, 23, 13), np.float32)
                lf = 0.45 + 0.002 * (4 * it + seed)
                orc.lib().orc_resample_lanczos_f32(want_l.ref(), src.ref(), C.c_float(lf), C.c_float(lf), 3)
                _lib.check(cvs.cvs_resample_lanczos_f32_dev(small.ref(), d_src.ref(), C.c_float(lf), C.c_float(lf), 3, None))
                assert_same_f32(small.download().array, want_l.array, "lanczos %r" % lf)
        except Exception as e:          # noqa: BLE001 -- reported on the main thread
            errors.append("thread %d: %r" % (seed, e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors


@pytest.mark.parametrize("tfull,sfull,scur,tp,sp,fac", SCALE_CASES)
def test_scale_bilinear_f16_twin(cvs, orc, tfull, sfull, scur, tp, sp, fac):
    """The scaler between two f16 frames = widen -> video_scale_bilinear_f32 -> truncate, without those two copies."""
    rng = np.random.default_rng(56)
    src = rand_f16_frame(rng, sfull, scur)
    src32 = HostFrame(sfull, np.float32, orc.half_to_float(src.array), scur)
    want32 = HostFrame(tfull, np.float32)
    orc.lib().orc_scale_bilinear_f32(want32.ref(), v2f(*tp), src32.ref(), v2f(*sp), v2f(*fac))
    want = HostFrame(tfull, np.uint16, orc.float_to_half(want32.array), want32.current_window)
    d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.uint16)
    _lib.check(cvs.cvs_scale_bilinear_f16_dev(d_out.ref(), v2f(*tp), d_src.ref(), v2f(*sp), v2f(*fac), None))
    got = d_out.download()
    assert same_window(got.current_window, want.current_window)
    if not want.current_window.is_empty():
        assert_same_f16(got.window_view(), want.window_view(), "f16 scale %r" % (fac,))


@pytest.mark.parametrize("form", ["tiles", "strips"])
@pytest.mark.parametrize("tw,fmt", [(1025, "f16"), (1026, "f16"), (1100, "f32"), (1027, "f32")])
@pytest.mark.parametrize("fac", [(2.0, 2.0), (1.5, 1.5), (1.25, 1.25)])
def test_scale_wide_targets_two_columns_per_lane(cvs, orc, force_fir, tw, fmt, fac, form):
    """Enlarging into a wide target, in the kernel's two forms.  Tiles (k_fir_tile_vh, the automatic choice): 128 columns x 16
    lines per workgroup, two columns per lane.  Strips (k_fir_vh): targets of 1024 columns and more take its
    two-columns-per-lane instances.  Either way the adjacent pair is one 16-byte store for halfs (columns l and l + 64 for
    floats), and an odd row pitch (1025, 1027) puts half of the rows' pairs off a 16-byte boundary: those f16 targets go to
    the strips with one column per lane.  Ragged last strip and ragged last tile row (37 lines) either way."""
    th = 37
    force_fir(form)
    sw, sh = int(tw / fac[0]) + 2, int(th / fac[1]) + 2
    sfull, tfull = (0, 0, sw - 1, sh - 1), (0, 0, tw - 1, th - 1)
    rng = np.random.default_rng(4100 + tw)
    src16 = rand_f16_frame(rng, sfull, sfull)
    src32 = HostFrame(sfull, np.float32, orc.half_to_float(src16.array), sfull)
    want32 = HostFrame(tfull, np.float32)
    orc.lib().orc_scale_bilinear_f32(want32.ref(), v2f(0, 0), src32.ref(), v2f(0, 0), v2f(*fac))
    if fmt == "f16":
        d_src, d_out = DeviceFrame.from_host(src16), DeviceFrame(tfull, np.uint16)
        _lib.check(cvs.cvs_scale_bilinear_f16_dev(d_out.ref(), v2f(0, 0), d_src.ref(), v2f(0, 0), v2f(*fac), None))
        got = d_out.download()
        assert same_window(got.current_window, want32.current_window)
        assert_same_f16(got.window_view(), orc.float_to_half(want32.window_view()), "wide f16 scale %r" % (fac,))
    else:
        d_src, d_out = DeviceFrame.from_host(src32), DeviceFrame(tfull, np.float32)
        _lib.check(cvs.cvs_scale_bilinear_f32_dev(d_out.ref(), v2f(0, 0), d_src.ref(), v2f(0, 0), v2f(*fac), None))
        got = d_out.download()
        assert same_window(got.current_window, want32.current_window)
        assert_same_f32(got.window_view(), want32.window_view(), "wide f32 scale %r" % (fac,))
    tiled = form == "tiles" and not (fmt == "f16" and tw % 2)
    assert cvs.cvs_scale_last_was_fused() == 1 and cvs.cvs_fir_last_kernel() == (_lib.FIR_KERNEL_TILE_VH if tiled else _lib.FIR_KERNEL_VH)


@pytest.mark.parametrize("fmt", ["f16", "f32"])
def test_scaler_form_by_format_and_size(cvs, force_fir, fmt):
    """Which form the library picks where both take the call (tile_vh_ops.hip cvk_fir_tvh_preferred, measured: profiles/r04/
    scaler_forms.txt): floats -- the tiles; halfs -- the tiles up to about a 4K target, the strips beyond.  A 4224 x 2600
    target (11 Mpx) from 2112 x 1300: halfs go to the strips, floats to the tiles, and pinned to the other form the pixels
    are the same."""
    w, h = 2112, 1300
    src16 = synth.layer_frame(w, h, 2, 0)
    outs = {}
    for pin in (None, "tiles", "strips"):
        force_fir(pin)
        if fmt == "f16":
            d_src, d_out = DeviceFrame.from_host(src16), DeviceFrame((0, 0, 2 * w - 1, 2 * h - 1), np.uint16)
            _lib.check(cvs.cvs_scale_bilinear_f16_dev(d_out.ref(), v2f(0, 0), d_src.ref(), v2f(0, 0), v2f(2.0, 2.0), None))
        else:
            from tests.models import h2f_ieee
            d_src = DeviceFrame.from_host(HostFrame(src16.full_window, np.float32, h2f_ieee(src16.array).astype(np.float32)))
            d_out = DeviceFrame((0, 0, 2 * w - 1, 2 * h - 1), np.float32)
            _lib.check(cvs.cvs_scale_bilinear_f32_dev(d_out.ref(), v2f(0, 0), d_src.ref(), v2f(0, 0), v2f(2.0, 2.0), None))
        outs[pin] = (_KERNEL_NAMES[cvs.cvs_fir_last_kernel()], d_out.download().array.copy())
        d_src.free(); d_out.free()
    assert outs["tiles"][0] == "tile-vh" and outs["strips"][0] == "vh"
    assert outs[None][0] == ("vh" if fmt == "f16" else "tile-vh")
    same = assert_same_f16 if fmt == "f16" else assert_same_f32
    same(outs["tiles"][1], outs["strips"][1], "tiles against strips, %s" % fmt)
    same(outs[None][1], outs["strips"][1], "automatic choice against strips, %s" % fmt)


@pytest.mark.parametrize("pin", [None, "tiles", "strips"])
@pytest.mark.parametrize("fmt,count,fac,tw,th", [("f16", 5, (2.0, 2.0), 384, 70), ("f16", 11, (1.5, 1.5), 300, 41), ("f32", 3, (2.0, 2.0), 260, 33),
                                                ("f32", 9, (3.0, 2.0), 512, 25), ("f16", 4, (0.5, 0.5), 300, 40), ("f16", 3, (1.5, 2.0), 300, 40)])
def test_scale_batch_is_the_single_calls(cvs, orc, force_fir, fmt, count, fac, tw, th, pin):
    """cvs_scale_bilinear_f16/f32_batch_dev: independent frames of one geometry, up to eight per launch (grid.z = frame) of
    one of the two vertical-first kernels -- left alone, halfs on the strips and floats on the tiles; pinned, either.  The
    pixels must be those of `count` single calls, and those are checked against the oracle.  11 and 9 frames: a launch of
    eight and one of three / of one (a single frame goes the ordinary way); factor 1/2: batched on the strips; a horizontal
    factor below the vertical one: no batched form, frame by frame -- same answer."""
    force_fir(pin)
    rng = np.random.default_rng(900 + count)
    sw, sh = int(tw / fac[0]) + 3, int(th / fac[1]) + 3
    sfull, tfull = (-2, 1, sw - 3, sh), (4, -3, tw + 3, th - 4)
    scur = (0, 2, sw - 5, sh - 1)
    tp, sp = (4.5, -3.0), (0.75, 2.0)
    srcs16 = [rand_f16_frame(rng, sfull, scur) for _ in range(count)]
    if fmt == "f16":
        d_src = [DeviceFrame.from_host(f) for f in srcs16]
        d_one = [DeviceFrame(tfull, np.uint16) for _ in range(count)]
        d_bat = [DeviceFrame(tfull, np.uint16) for _ in range(count)]
        tab = lambda fr: (C.POINTER(_lib.rgba_frame_f16_t) * len(fr))(*[C.pointer(f.c) for f in fr])
        single, batch = cvs.cvs_scale_bilinear_f16_dev, cvs.cvs_scale_bilinear_f16_batch_dev
    else:
        srcs32 = [HostFrame(sfull, np.float32, orc.half_to_float(f.array), scur) for f in srcs16]
        d_src = [DeviceFrame.from_host(f) for f in srcs32]
        d_one = [DeviceFrame(tfull, np.float32) for _ in range(count)]
        d_bat = [DeviceFrame(tfull, np.float32) for _ in range(count)]
        tab = lambda fr: (C.POINTER(_lib.rgba_frame_f32_t) * len(fr))(*[C.pointer(f.c) for f in fr])
        single, batch = cvs.cvs_scale_bilinear_f32_dev, cvs.cvs_scale_bilinear_f32_batch_dev
    for o, i in zip(d_one, d_src):
        _lib.check(single(o.ref(), v2f(*tp), i.ref(), v2f(*sp), v2f(*fac), None))
    _lib.check(batch(tab(d_bat), v2f(*tp), tab(d_src), v2f(*sp), v2f(*fac), count, None))
    if fac[0] >= fac[1] and fac[1] > 1.0 and count % 8 != 1:
        tiles = pin == "tiles" or (pin is None and fmt == "f32")
        assert cvs.cvs_fir_last_kernel() == (_lib.FIR_KERNEL_TILE_VH if tiles else _lib.FIR_KERNEL_VH)
    force_fir(None)
    src32_0 = HostFrame(sfull, np.float32, orc.half_to_float(srcs16[0].array), scur)
    want32 = HostFrame(tfull, np.float32)
    orc.lib().orc_scale_bilinear_f32(want32.ref(), v2f(*tp), src32_0.ref(), v2f(*sp), v2f(*fac))
    for k, (a, b) in enumerate(zip(d_one, d_bat)):
        ga, gb = a.download(), b.download()
        assert same_window(ga.current_window, gb.current_window), (k, ga.current_window.tuple(), gb.current_window.tuple())
        assert np.array_equal(ga.array.view(np.uint8), gb.array.view(np.uint8)), "frame %d of the batch differs from its single call" % k
        if k == 0:
            assert same_window(gb.current_window, want32.current_window)
            if fmt == "f16":
                assert_same_f16(gb.window_view(), orc.float_to_half(want32.window_view()), "batched scale f16")
            else:
                assert_same_f32(gb.window_view(), want32.window_view(), "batched scale f32")
    for d in d_src + d_one + d_bat:
        d.free()


def test_scale_batch_falls_back_on_mixed_geometry_and_overlap(cvs, orc):
    """Frames of different geometry, or a target that is another frame's source, are not batched: the call still gives every
    frame what its single call gives."""
    rng = np.random.default_rng(77)
    a = [rand_f16_frame(rng, (0, 0, 199, 29), (0, 0, 199, 29)) for _ in range(3)]
    b = rand_f16_frame(rng, (0, 0, 149, 29), (0, 0, 149, 29))                    # a narrower source
    srcs = [DeviceFrame.from_host(f) for f in (a[0], b, a[1], a[2])]
    outs = [DeviceFrame((0, 0, 399, 59), np.uint16) for _ in range(4)]
    ones = [DeviceFrame((0, 0, 399, 59), np.uint16) for _ in range(4)]
    tab = lambda fr: (C.POINTER(_lib.rgba_frame_f16_t) * len(fr))(*[C.pointer(f.c) for f in fr])
    _lib.check(cvs.cvs_scale_bilinear_f16_batch_dev(tab(outs), v2f(0, 0), tab(srcs), v2f(0, 0), v2f(2.0, 2.0), 4, None))
    for o, i in zip(ones, srcs):
        _lib.check(cvs.cvs_scale_bilinear_f16_dev(o.ref(), v2f(0, 0), i.ref(), v2f(0, 0), v2f(2.0, 2.0), None))
    for k in range(4):
        x, y = outs[k].download(), ones[k].download()
        assert same_window(x.current_window, y.current_window) and np.array_equal(x.array, y.array), k
    # the same target twice: the second write must see what a sequence of single calls leaves (the last frame's pixels)
    twice = [outs[0], outs[0]]
    _lib.check(cvs.cvs_scale_bilinear_f16_batch_dev(tab(twice), v2f(0, 0), tab([srcs[0], srcs[2]]), v2f(0, 0), v2f(2.0, 2.0), 2, None))
    assert np.array_equal(outs[0].download().array, ones[2].download().array)
    for d in srcs + outs + ones:
        d.free()


def test_tile_scaler_random_geometry(cvs, orc):
    """The tile form of the vertical-first scaler (k_fir_tile_vh) takes targets of 256 columns and more whose tables are short
    and narrow.  80 random set-ups wide enough for it -- enlarging factors with the vertical one never the larger, fractional
    points, source windows inside their buffers, buffers with negative origins, an Inf and a NaN in the source -- against
    the oracle, f32 and f16; most of them must indeed have run on the tiles."""
    rng = np.random.default_rng(4242)
    seen = {}
    for case in range(80):
        fy = float(rng.choice([1.125, 1.25, 1.5, 2.0, 2.5, 3.0]))
        fx = float(rng.choice([f for f in (1.25, 1.5, 2.0, 2.5, 3.0, 4.0) if f >= fy]))
        tw, th = int(rng.integers(256, 520)), int(rng.integers(5, 70))
        tx0, ty0 = int(rng.integers(-6, 4)) * 2, int(rng.integers(-5, 4))
        tfull = (tx0, ty0, tx0 + tw - 1, ty0 + th - 1)
        sw, sh = int(tw / fx) + int(rng.integers(-20, 6)), int(th / fy) + int(rng.integers(-3, 4))
        sx0, sy0 = int(rng.integers(-4, 4)), int(rng.integers(-4, 4))
        sfull = (sx0, sy0, sx0 + max(sw, 8) - 1, sy0 + max(sh, 3) - 1)
        scur = sfull if case % 3 else _random_window(rng, sfull, allow_empty=False)
        tp = (float(rng.choice([0.0, 0.5, 2.25])) + tx0, float(rng.choice([0.0, 1.0, 3.5])) + ty0)
        sp = (float(rng.choice([0.0, 0.75, 2.0])) + sx0, float(rng.choice([0.0, 0.5, 1.0])) + sy0)
        src16 = rand_f16_frame(rng, sfull, scur)
        v = src16.window_view()
        if v.shape[0] > 2 and v.shape[1] > 9:
            v[1, 7, 0] = 0x7C00
            v[v.shape[0] - 1, 3, 2] = 0x7E00
        src32 = HostFrame(sfull, np.float32, orc.half_to_float(src16.array), scur)
        want32 = HostFrame(tfull, np.float32)
        orc.lib().orc_scale_bilinear_f32(want32.ref(), v2f(*tp), src32.ref(), v2f(*sp), v2f(fx, fy))
        if case % 2:
            d_src, d_out = DeviceFrame.from_host(src16), DeviceFrame(tfull, np.uint16)
            _lib.check(cvs.cvs_scale_bilinear_f16_dev(d_out.ref(), v2f(*tp), d_src.ref(), v2f(*sp), v2f(fx, fy), None))
            got = d_out.download()
            assert same_window(got.current_window, want32.current_window), (case, got.current_window.tuple(), want32.current_window.tuple())
            if not want32.current_window.is_empty():
                assert_same_f16(got.window_view(), orc.float_to_half(want32.window_view()), "tile scale f16, case %d %r" % (case, (fx, fy)))
        else:
            d_src, d_out = DeviceFrame.from_host(src32), DeviceFrame(tfull, np.float32)
            _lib.check(cvs.cvs_scale_bilinear_f32_dev(d_out.ref(), v2f(*tp), d_src.ref(), v2f(*sp), v2f(fx, fy), None))
            got = d_out.download()
            assert same_window(got.current_window, want32.current_window), (case, got.current_window.tuple(), want32.current_window.tuple())
            if not want32.current_window.is_empty():
                assert_same_f32(got.window_view(), want32.window_view(), "tile scale f32, case %d %r" % (case, (fx, fy)))
        name = _KERNEL_NAMES[cvs.cvs_fir_last_kernel()]
        seen[name] = seen.get(name, 0) + 1
        d_src.free(); d_out.free()
    assert seen.get("tile-vh", 0) >= 50, seen
    assert cvs.cvs_fir_fell_through_count() == 0


# ------------------------------------------------------------------ randomized window sweeps (region walks of video_mix.c / copy / scale)

def _random_window(rng, full, allow_empty=True, inside=None):
    """A random window inside `full` (and inside `inside`, when given: the reference's mixers write a lone frame's whole
    row span, video_mix.c:146-149, so an input window reaching beyond the OUTPUT buffer makes the reference itself
    write out of bounds -- its callers always pull inputs into temps that share the output's full window)."""
    if inside is not None:
        full = (max(full[0], inside[0]), max(full[1], inside[1]), min(full[2], inside[2]), min(full[3], inside[3]))
        if full[2] < full[0] or full[3] < full[1]:
            return (0, 0, -1, -1)
    x0, y0, x1, y1 = full
    if allow_empty and rng.random() < 0.08:
        return (0, 0, -1, -1)
    ax, bx = sorted(int(v) for v in rng.integers(x0, x1 + 1, 2))
    ay, by = sorted(int(v) for v in rng.integers(y0, y1 + 1, 2))
    return (ax, ay, bx, by)


def test_mix_over_and_cross_random_windows(cvs, orc):
    """300 random (out, upper) window pairs per mixer on the device twins, frames with different buffers and origins,
    whole-buffer compare: the 9-region walk, the `left` selector quirk (video_mix.c:137,265) and every copy / zero /
    blend region land where the reference puts them."""
    rng = np.random.default_rng(20261003)
    for case in range(300):
        out_full = (int(rng.integers(-6, 3)), int(rng.integers(-4, 3)), int(rng.integers(14, 30)), int(rng.integers(8, 16)))
        b_full = (int(rng.integers(-6, 3)), int(rng.integers(-4, 3)), int(rng.integers(14, 30)), int(rng.integers(8, 16)))
        mix = float(rng.choice([0.0, 0.25, 0.5, 1.0, 1.5, -0.5]))
        # over, in place on `out`
        out = rand_f32_frame(rng, out_full, _random_window(rng, out_full), "mixed")
        upper = rand_f32_frame(rng, b_full, _random_window(rng, b_full, inside=out_full), "mixed")
        want = out.copy()
        d_out, d_up = DeviceFrame.from_host(out), DeviceFrame.from_host(upper)
        _lib.check(cvs.cvs_mix_over_f32_dev(d_out.ref(), d_up.ref(), C.c_float(mix), None))
        orc.lib().orc_mix_over_f32(want.ref(), upper.ref(), C.c_float(mix))
        got = d_out.download()
        assert same_window(got.current_window, want.current_window), ("over", case)
        assert_same_f32(got.array, want.array, "over, random case %d" % case)
        # cross into a third buffer
        a = rand_f32_frame(rng, b_full, _random_window(rng, b_full, inside=out_full), "mixed")
        target = rand_f32_frame(rng, out_full)
        want = target.copy()
        d_t, d_a = DeviceFrame.from_host(target), DeviceFrame.from_host(a)
        _lib.check(cvs.cvs_mix_cross_f32_dev(d_t.ref(), d_a.ref(), d_up.ref(), C.c_float(mix), None))
        orc.lib().orc_mix_cross_f32(want.ref(), a.ref(), upper.ref(), C.c_float(mix))
        got = d_t.download()
        assert same_window(got.current_window, want.current_window), ("cross", case)
        assert_same_f32(got.array, want.array, "cross, random case %d" % case)


def test_copy_and_convert_random_windows(cvs, orc):
    rng = np.random.default_rng(20261004)
    for case in range(200):
        out_full = (int(rng.integers(-6, 3)), int(rng.integers(-4, 3)), int(rng.integers(10, 30)), int(rng.integers(6, 16)))
        in_full = (int(rng.integers(-6, 3)), int(rng.integers(-4, 3)), int(rng.integers(10, 30)), int(rng.integers(6, 16)))
        src = rand_f16_frame(rng, in_full, _random_window(rng, in_full))
        out = rand_f16_frame(rng, out_full)
        want = out.copy()
        d_src, d_out = DeviceFrame.from_host(src), DeviceFrame.from_host(out)
        _lib.check(cvs.cvs_copy_frame_f16_dev(d_out.ref(), d_src.ref(), None))
        orc.lib().orc_copy_frame_f16(want.ref(), src.ref())
        got = d_out.download()
        assert same_window(got.current_window, want.current_window), case
        assert_same_f16(got.array, want.array, "copy, random case %d" % case)
        alpha = float(rng.choice([1.0, 0.5, 0.0, 2.0]))
        src32 = rand_f32_frame(rng, in_full, _random_window(rng, in_full), "mixed")
        out32 = rand_f32_frame(rng, out_full)
        want32 = out32.copy()
        d_s32, d_o32 = DeviceFrame.from_host(src32), DeviceFrame.from_host(out32)
        _lib.check(cvs.cvs_copy_frame_alpha_f32_dev(d_o32.ref(), d_s32.ref(), C.c_float(alpha), None))
        orc.lib().orc_copy_frame_alpha_f32(want32.ref(), src32.ref(), C.c_float(alpha))
        got32 = d_o32.download()
        assert same_window(got32.current_window, want32.current_window), case
        if not want32.current_window.is_empty():
            assert_same_f32(got32.window_view(), want32.window_view(), "copy alpha, random case %d" % case)


def test_scale_bilinear_random_geometry(cvs, orc):
    """120 random scaler set-ups (factors on both sides of 1, fractional points, source windows inside their buffers)
    against the oracle: window reported, and every pixel inside it."""
    rng = np.random.default_rng(20261005)
    for case in range(120):
        sfull = (int(rng.integers(-4, 3)), int(rng.integers(-3, 3)), int(rng.integers(10, 28)), int(rng.integers(6, 18)))
        tfull = (int(rng.integers(-4, 3)), int(rng.integers(-3, 3)), int(rng.integers(10, 40)), int(rng.integers(6, 30)))
        scur = _random_window(rng, sfull, allow_empty=False)
        fac = (float(rng.choice([0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0])), float(rng.choice([0.25, 0.5, 0.8, 1.0, 1.25, 2.0, 4.0])))
        tp = (float(rng.choice([0.0, 0.5, 2.25])), float(rng.choice([0.0, 1.0, 3.5])))
        sp = (float(rng.choice([0.0, 0.75, 2.0])), float(rng.choice([0.0, 0.5, 1.0])))
        src = rand_f32_frame(rng, sfull, scur)
        want = HostFrame(tfull, np.float32)
        orc.lib().orc_scale_bilinear_f32(want.ref(), v2f(*tp), src.ref(), v2f(*sp), v2f(*fac))
        d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.float32)
        _lib.check(cvs.cvs_scale_bilinear_f32_dev(d_out.ref(), v2f(*tp), d_src.ref(), v2f(*sp), v2f(*fac), None))
        got = d_out.download()
        assert same_window(got.current_window, want.current_window), (case, fac, tp, sp, got.current_window.tuple(), want.current_window.tuple())
        if not want.current_window.is_empty():
            assert_same_f32(got.window_view(), want.window_view(), "scale, random case %d %r" % (case, fac))


@pytest.mark.parametrize("fac,fmt", [((1.5, 1.5), "f32"), ((2.0, 2.0), "f16"), ((0.75, 0.75), "f16"), ((1.25, 1.125), "f32"), ((0.8, 0.6), "f32"), ((0.75, 1.5), "f16"), ((1.25, 2.0), "f32"), ((0.5, 0.5), "f16"), ((0.4, 0.35), "f32")])
def test_scaler_in_one_launch_agrees_with_the_two_passes(cvs, force_fir, fac, fmt):
    """1920x1080 through the triangle scaler: the automatic choice runs both passes in one launch (vertical pass first:
    sweep_vh_ops.hip; horizontal factor smaller, so horizontal first: sweep_hv_ops.hip); pinned to the older kernels it runs the reference's two passes through an f32 frame.
    Same tables, same order of roundings: the frames must be equal bit for bit (each form is checked against the oracle at
    small sizes by the tests above)."""
    w, h = 1920, 1080
    tw, th = int(w * fac[0]), int(h * fac[1])
    src16 = synth.layer_frame(w, h, 1, 0)
    src16.array[5, 7, 1] = 0x7C00                                    # an Inf and a NaN: they must spread the same way
    src16.array[400, 900, 2] = 0x7E00
    outs, fused = [], []
    # the tile form takes the enlarging calls whose rows of LDS fit (tile_vh_ops.hip cvk_fir_tvh_supported)
    tile_cases = {((1.5, 1.5), "f32"), ((2.0, 2.0), "f16"), ((1.25, 1.125), "f32")}
    for kernel in (None, "strips", "passes"):
        force_fir(kernel)
        if fmt == "f16":
            d_src, d_out = DeviceFrame.from_host(src16), DeviceFrame((0, 0, tw - 1, th - 1), np.uint16)
            _lib.check(cvs.cvs_scale_bilinear_f16_dev(d_out.ref(), v2f(0, 0), d_src.ref(), v2f(0, 0), v2f(*fac), None))
        else:
            from tests.models import h2f_ieee
            d_src = DeviceFrame.from_host(HostFrame(src16.full_window, np.float32, h2f_ieee(src16.array).astype(np.float32)))
            d_out = DeviceFrame((0, 0, tw - 1, th - 1), np.float32)
            _lib.check(cvs.cvs_scale_bilinear_f32_dev(d_out.ref(), v2f(0, 0), d_src.ref(), v2f(0, 0), v2f(*fac), None))
        fused.append(cvs.cvs_scale_last_was_fused())
        # video_scale.c:252: the smaller factor's pass first -- horizontal first is the channel-pair sweep, else k_fir_vh
        # ... and when the horizontal pass goes first: the per-line gather unless the horizontal axis reduces (hv_goes_first)
        # (the horizontal-first gather exists in the default arithmetic flavour only: the contracted one runs the two passes)
        contracted = cvs.cvs_get_arithmetic() == _lib.ARITH_CONTRACTED
        vh = "tile-vh" if kernel is None and (fac, fmt) in tile_cases else "vh"
        assert _KERNEL_NAMES[cvs.cvs_fir_last_kernel()] == ("pass" if kernel == "passes" or (contracted and fac[0] < fac[1]) else "hv" if fac[0] < fac[1] else vh)
        got = d_out.download()
        outs.append((got.current_window.tuple(), got.array.copy()))
        d_src.free(); d_out.free()
    contracted = cvs.cvs_get_arithmetic() == _lib.ARITH_CONTRACTED
    assert fused == [0 if contracted and fac[0] < fac[1] else 1] * 2 + [0]
    assert outs[0][0] == outs[1][0] == outs[2][0]
    for k, form in ((0, "automatic choice"), (1, "strips")):
        (assert_same_f16 if fmt == "f16" else assert_same_f32)(outs[k][1], outs[2][1], "scaler, one launch (%s) against two" % form)


def test_fir_blur_random_geometry(cvs, orc):
    """150 random blurs: tap counts 1..34 (odd ones from 3 to 31 take the register-window kernel, the rest the tiled
    gather kernel), source windows inside their buffers, targets with other origins, f32 and f16 entry points."""
    rng = np.random.default_rng(20261006)
    for case in range(150):
        sfull = (int(rng.integers(-5, 3)), int(rng.integers(-4, 3)), int(rng.integers(20, 300)), int(rng.integers(8, 50)))
        tfull = (int(rng.integers(-5, 3)), int(rng.integers(-4, 3)), int(rng.integers(20, 300)), int(rng.integers(8, 50)))
        scur = _random_window(rng, sfull)
        ntaps = int(rng.integers(1, 35))
        taps = rng.uniform(-0.2, 1.0, ntaps).astype(np.float32)
        taps /= np.float32(taps.sum())
        src = rand_f32_frame(rng, sfull, scur, lo=-0.5, hi=1.5)
        want = HostFrame(tfull, np.float32)
        orc.lib().orc_fir_blur_f32(want.ref(), src.ref(), f32p(taps), ntaps)
        d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.float32)
        _lib.check(cvs.cvs_fir_blur_f32_dev(d_out.ref(), d_src.ref(), f32p(taps), ntaps, None))
        got = d_out.download()
        assert same_window(got.current_window, want.current_window), (case, ntaps)
        if not want.current_window.is_empty():
            assert_same_f32(got.window_view(), want.window_view(), "blur f32, random case %d, %d taps" % (case, ntaps))
        # the f16 entry point on the truncated source
        from canvas_amd.synth import truncate_to_half
        src16 = HostFrame(sfull, np.uint16, truncate_to_half(src.array), scur)
        wide = HostFrame(sfull, np.float32, orc.half_to_float(src16.array), scur)
        want32 = HostFrame(tfull, np.float32)
        orc.lib().orc_fir_blur_f32(want32.ref(), wide.ref(), f32p(taps), ntaps)
        want16 = HostFrame(tfull, np.uint16, orc.float_to_half(want32.array), want32.current_window)
        d_src16, d_out16 = DeviceFrame.from_host(src16), DeviceFrame(tfull, np.uint16)
        _lib.check(cvs.cvs_fir_blur_f16_dev(d_out16.ref(), d_src16.ref(), f32p(taps), ntaps, None))
        got16 = d_out16.download()
        assert same_window(got16.current_window, want16.current_window), (case, ntaps)
        if not want16.current_window.is_empty():
            assert_same_f16(got16.window_view(), want16.window_view(), "blur f16, random case %d, %d taps" % (case, ntaps))


def test_lanczos_random_geometry(cvs, orc):
    """80 random resamples: factor 1/2 on both axes (uniform taps: register-window kernel) and other factors (per-line
    taps: tiled gather kernel, or two passes when a footprint does not fit), kernel sizes 1..4, windows with origins."""
    rng = np.random.default_rng(20261007)
    for case in range(80):
        sfull = (int(rng.integers(-5, 3)), int(rng.integers(-4, 3)), int(rng.integers(20, 200)), int(rng.integers(8, 60)))
        tfull = (int(rng.integers(-3, 3)), int(rng.integers(-3, 3)), int(rng.integers(8, 120)), int(rng.integers(6, 40)))
        scur = _random_window(rng, sfull, allow_empty=False)
        if rng.random() < 0.5:
            fx = fy = 0.5
        else:
            fx, fy = float(rng.choice([0.25, 0.4, 0.5, 1.0, 1.5, 2.0])), float(rng.choice([0.3, 0.5, 0.75, 1.0, 2.0]))
        ksize = int(rng.integers(1, 5))
        src = rand_f32_frame(rng, sfull, scur, lo=-0.5, hi=1.5)
        want = HostFrame(tfull, np.float32)
        orc.lib().orc_resample_lanczos_f32(want.ref(), src.ref(), C.c_float(fx), C.c_float(fy), ksize)
        d_src, d_out = DeviceFrame.from_host(src), DeviceFrame(tfull, np.float32)
        _lib.check(cvs.cvs_resample_lanczos_f32_dev(d_out.ref(), d_src.ref(), C.c_float(fx), C.c_float(fy), ksize, None))
        got = d_out.download()
        assert same_window(got.current_window, want.current_window), (case, fx, fy, ksize)
        if not want.current_window.is_empty():
            assert_same_f32(got.window_view(), want.window_view(), "lanczos, random case %d (%g, %g, k=%d)" % (case, fx, fy, ksize))


def test_lanczos_random_factors_every_sweep_against_the_oracle(cvs, orc, force_fir):
    """60 resamples at factors drawn from [0.28, 3.2] per axis (no two lines share a tap list), targets of several strips,
    source windows inside their buffers, a few Inf / NaN pixels: the automatic choice, then the per-line gather and the
    channel-pair sweep pinned in turn (each falls back to the next kernel in line where it has no instance) -- all three
    must give the oracle's frame.  f16 frames on every third case (the f16 entry: widen, resample, truncate)."""
    rng = np.random.default_rng(20261103)
    for case in range(60):
        fx, fy = float(np.float32(rng.uniform(0.28, 3.2))), float(np.float32(rng.uniform(0.28, 3.2)))
        sw, sh = int(rng.integers(40, 260)), int(rng.integers(24, 90))
        tw, th = max(8, min(420, int(sw * fx))), max(6, min(140, int(sh * fy)))
        sfull, tfull = (0, 0, sw - 1, sh - 1), (0, 0, tw - 1, th - 1)
        scur = _random_window(rng, sfull, allow_empty=False) if case % 4 == 0 else sfull
        half = case % 3 == 0
        if half:
            src = rand_f16_frame(rng, sfull, scur)
            src32 = HostFrame(sfull, np.float32, orc.half_to_float(src.array), scur)
        else:
            src = src32 = rand_f32_frame(rng, sfull, scur, lo=-0.5, hi=1.5)
            for k, v in enumerate([np.inf, np.nan, -np.inf]):
                src.array[(13 * k + case) % sh, (29 * k + 3 * case) % sw, k % 4] = v
        want = HostFrame(tfull, np.float32)
        orc.lib().orc_resample_lanczos_f32(want.ref(), src32.ref(), C.c_float(fx), C.c_float(fy), 3)
        d_src = DeviceFrame.from_host(src)
        for kernel in (None, "hv", "tiled"):
            force_fir(kernel)
            if half:
                d_out = DeviceFrame(tfull, np.uint16)
                _lib.check(cvs.cvs_resample_lanczos_f16_dev(d_out.ref(), d_src.ref(), C.c_float(fx), C.c_float(fy), 3, None))
                got = d_out.download()
                assert same_window(got.current_window, want.current_window), (case, fx, fy, kernel)
                if not want.current_window.is_empty():
                    assert_same_f16(got.window_view(), orc.float_to_half(want.window_view()), "case %d (%g, %g) %s f16" % (case, fx, fy, kernel))
            else:
                d_out = DeviceFrame(tfull, np.float32)
                _lib.check(cvs.cvs_resample_lanczos_f32_dev(d_out.ref(), d_src.ref(), C.c_float(fx), C.c_float(fy), 3, None))
                got = d_out.download()
                assert same_window(got.current_window, want.current_window), (case, fx, fy, kernel)
                if not want.current_window.is_empty():
                    assert_same_f32(got.window_view(), want.window_view(), "case %d (%g, %g) %s" % (case, fx, fy, kernel))
            d_out.free()
        d_src.free()


def test_chain_and_crossfade_on_arbitrary_half_codes(cvs, orc):
    """Layers made of uniformly random 16-bit patterns -- every exponent class, subnormals, Inf, NaN, both signs, in
    colour and in alpha: the band test of the shared-reciprocal divide, the Inf saturation of the truncation and the
    zero-alpha rule all get inputs no image would give them.  (NaN payloads and zero signs are folded, as everywhere.)"""
    rng = np.random.default_rng(20261008)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    w, h = 128, 33
    full = (0, 0, w - 1, h - 1)
    for case in range(12):
        nl = int(rng.integers(2, 5))
        layers = [HostFrame(full, np.uint16, rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)) for _ in range(nl)]
        if case % 3 == 0:
            layers[0].array[..., 3] = 0x3C00                     # an opaque base now and then
        dl = [DeviceFrame.from_host(l) for l in layers]
        out = DeviceFrame(full, np.uint16)
        for matrix, pre, table in [(m, _lib.LUT_REC709_TO_LINEAR_SCENE, orc.transfer_table(0)), (None, _lib.LUT_NONE, None)]:
            want = orc.chain_color_over(layers, matrix, table, None)
            chain_color_over([(out, dl)], matrix, pre, _lib.LUT_NONE)
            _lib.check(cvs.cvs_stream_sync(None))
            assert cvs.cvs_chain_last_was_fused() == 1
            assert_same_f16(out.download().array, want.array, "arbitrary codes, case %d, %d layers, matrix %s" % (case, nl, matrix is not None))
        mix = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
        want = _oracle_cross_f16(orc, full, layers[0], layers[1], mix)
        _lib.check(cvs.cvs_mix_cross_f16_dev(out.ref(), dl[0].ref(), dl[1].ref(), C.c_float(mix), None))
        assert_same_f16(out.download().array, want.array, "arbitrary codes, crossfade, case %d" % case)


def test_pointwise_ops_on_arbitrary_bit_patterns(cvs, orc):
    """mix_over / mix_cross / copy_alpha on uniformly random f32 bit patterns, gain/offset, colour matrix and the
    conversions on uniformly random half codes: NaNs, infinities, denormals and signed zeros everywhere."""
    rng = np.random.default_rng(20261009)
    full = (0, 0, 95, 40)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    for case in range(8):
        bits = lambda: rng.integers(0, 2 ** 32, (41, 96, 4), dtype=np.uint32).view(np.float32)      # noqa: E731
        a, b = HostFrame(full, np.float32, bits()), HostFrame(full, np.float32, bits())
        for mix in (1.0, 0.4):
            want = a.copy()
            orc.lib().orc_mix_over_f32(want.ref(), b.ref(), C.c_float(mix))
            d_a, d_b = DeviceFrame.from_host(a), DeviceFrame.from_host(b)
            _lib.check(cvs.cvs_mix_over_f32_dev(d_a.ref(), d_b.ref(), C.c_float(mix), None))
            assert_same_f32(d_a.download().array, want.array, "over on raw bits, case %d" % case)
            want = HostFrame(full, np.float32)
            orc.lib().orc_mix_cross_f32(want.ref(), a.ref(), b.ref(), C.c_float(mix))
            d_o, d_a2 = DeviceFrame(full, np.float32), DeviceFrame.from_host(a)
            _lib.check(cvs.cvs_mix_cross_f32_dev(d_o.ref(), d_a2.ref(), d_b.ref(), C.c_float(mix), None))
            assert_same_f32(d_o.download().array, want.array, "cross on raw bits, case %d" % case)
        codes = HostFrame(full, np.uint16, rng.integers(0, 65536, (41, 96, 4), dtype=np.uint16))
        d_c = DeviceFrame.from_host(codes)
        # gain / offset
        want16, out16 = HostFrame(full, np.uint16), DeviceFrame(full, np.uint16)
        orc.lib().orc_gain_offset_f16(want16.ref(), codes.ref(), C.c_float(1.75), C.c_float(-0.125))
        _lib.check(cvs.cvs_gain_offset_f16_dev(out16.ref(), d_c.ref(), C.c_float(1.75), C.c_float(-0.125), None))
        assert_same_f16(out16.download().array, want16.array, "gain/offset on raw codes")
        # colour matrix with both tables
        want16 = codes.copy()
        orc.lib().orc_color_matrix_f16(want16.ref(), f32p(m), u16p(orc.transfer_table(0)), u16p(orc.transfer_table(2)))
        _lib.check(cvs.cvs_color_matrix_f16_to_dev(out16.ref(), d_c.ref(), f32p(m), 0, 2, None))
        assert_same_f16(out16.download().array, want16.array, "colour matrix on raw codes")
        # widen, narrow of raw bits
        wide = DeviceFrame(full, np.float32)
        _lib.check(cvs.cvs_frame_f16_to_f32_dev(wide.ref(), d_c.ref(), None))
        assert_same_f32(wide.download().array, orc.half_to_float(codes.array), "widen raw codes")
        d_a3 = DeviceFrame.from_host(a)
        _lib.check(cvs.cvs_frame_f32_to_f16_dev(out16.ref(), d_a3.ref(), None))
        assert_same_f16(out16.download().array, orc.float_to_half(a.array), "narrow raw bits")


# ------------------------------------------------------------------ HIP graph capture of launch-bound sequences

def test_graph_capture_replays_a_node_by_node_stack(cvs, orc):
    """A ragged three-layer stack takes the node-by-node path: a dozen short kernels and three pooled scratch frames.
    Recorded once as a HIP graph, replayed on changing layer CONTENTS: every replay equals the oracle."""
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    rng = np.random.default_rng(77001)
    full = (0, 0, 95, 53)
    wins = [full, (5, 3, 60, 40), (40, 2, 95, 30)]            # see test_chain_ragged_windows for the choice
    layers = [rand_f16_frame(rng, full, w, alpha="one" if i == 0 else "rand") for i, w in enumerate(wins)]
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame(full, np.uint16)
    stream = cvs.cvs_stream_create()
    chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)       # warm: tables, pool blocks
    _lib.check(cvs.cvs_stream_sync(stream))
    assert cvs.cvs_chain_last_was_fused() == 0
    _lib.check(cvs.cvs_graph_begin(stream))
    chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
    graph = cvs.cvs_graph_end(stream)
    assert graph, _lib.last_error()
    try:
        for replay in range(3):
            fresh = [rand_f16_frame(rng, full, w, alpha="one" if i == 0 else "rand") for i, w in enumerate(wins)]
            for d, f in zip(dl, fresh):
                d.upload(f.array, stream)
            cvs.cvs_memset(out.ptr, 0x11, out.nbytes, stream)
            _lib.check(cvs.cvs_graph_launch(graph, stream))
            _lib.check(cvs.cvs_stream_sync(stream))
            want = orc.chain_color_over(fresh, m, orc.transfer_table(0), None)
            got = out.download()
            x0, y0, x1, y1 = want.current_window.tuple()
            assert_same_f16(got.array[y0:y1 + 1, x0:x1 + 1], want.array[y0:y1 + 1, x0:x1 + 1], "graph replay %d" % replay)
        # other work on the same stream between replays does not disturb the graph's own scratch
        other = DeviceFrame(full, np.uint16)
        chain_color_over([(other, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)
        _lib.check(cvs.cvs_graph_launch(graph, stream))
        _lib.check(cvs.cvs_stream_sync(stream))
        assert np.array_equal(other.download().array[y0:y1 + 1, x0:x1 + 1], out.download().array[y0:y1 + 1, x0:x1 + 1])
    finally:
        cvs.cvs_graph_destroy(graph)
        cvs.cvs_stream_destroy(stream)
    # capture state is per thread and exclusive
    assert cvs.cvs_graph_end(None) is None
    _lib.check(cvs.cvs_graph_begin(None))
    assert cvs.cvs_graph_begin(None) != 0
    empty = cvs.cvs_graph_end(None)
    cvs.cvs_graph_destroy(empty)


def test_graph_capture_of_the_config5_frame(cvs, orc):
    from canvas_amd.stream import GraphStream
    from tests.util import oracle_graph
    w, h = 160, 90
    g = GraphStream(w, h, ring=1)
    stream = cvs.cvs_stream_create()
    g.render(0, stream)
    _lib.check(cvs.cvs_stream_sync(stream))
    _lib.check(cvs.cvs_graph_begin(stream))
    out = g.render(0, stream)
    graph = cvs.cvs_graph_end(stream)
    assert graph, _lib.last_error()
    try:
        for frame in (3, 4):
            inputs = GraphStream.host_inputs(w, h, frame)
            g.slots[0]["src"].upload(inputs[0].array, stream)
            for o, f in zip(g.slots[0]["over"], inputs[1:]):
                o.upload(f.array, stream)
            _lib.check(cvs.cvs_graph_launch(graph, stream))
            _lib.check(cvs.cvs_stream_sync(stream))
            want = oracle_graph(orc, inputs, g.matrix, orc.transfer_table(0), None, g.taps)
            assert_same_f16(out.download().array, want.array, "config 5 frame %d through a graph" % frame)
    finally:
        cvs.cvs_graph_destroy(graph)
        cvs.cvs_stream_destroy(stream)


def test_scratch_memory_comes_back(cvs, orc):
    """Hundreds of calls with ever-changing frame sizes (so the pool keeps meeting new block sizes and the tap-table
    cache keeps evicting), then a trim: HBM in use returns to where it started -- nothing is leaked per call."""
    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        _lib.check(cvs.cvs_stream_sync(None))
        _lib.check(cvs.cvs_mem_info(C.byref(f), C.byref(t)))
        return f.value

    rng = np.random.default_rng(4242)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    taps = synth.gaussian_taps(9, 1.5)

    def burst(count):
        for _ in range(count):
            w, h = int(rng.integers(40, 400)), int(rng.integers(30, 200))
            full = (0, 0, w - 1, h - 1)
            layers = [DeviceFrame.from_host(rand_f16_frame(rng, full, (2, 1, w - 3, h - 2) if k else full)) for k in range(3)]
            out = DeviceFrame(full, np.uint16)
            chain_color_over([(out, layers)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)          # ragged: node by node, pooled f32 frames
            small = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
            _lib.check(cvs.cvs_blur_lanczos_f16_dev(small.ref(), layers[0].ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
            big = DeviceFrame((0, 0, 2 * w - 1, 2 * h - 1), np.uint16)
            _lib.check(cvs.cvs_scale_bilinear_f16_dev(big.ref(), v2f(0, 0), layers[0].ref(), v2f(0, 0), v2f(2.0, 2.0), None))
            del layers, out, small, big

    burst(20)                                    # first use of everything (tables, LUTs, code objects)
    cvs.cvs_pool_trim()
    before = free_bytes()
    burst(150)
    cvs.cvs_pool_trim()
    after = free_bytes()
    assert before - after < 64 << 20, (before, after)        # the tap-table cache holds at most 16 small tables


# ------------------------------------------------------------------ device contexts: several GPUs (or one, several times) in one process

def _every_cached_table_once(cvs, orc, seed):
    """One call of each kind that builds a table ON the device (transfer table, FIR tap tables, byte table) plus the chain and
    a pooled intermediate; returns what each produced, checked against the oracle by the caller."""
    rng = np.random.default_rng(seed)
    w, h = 96, 54
    full = (0, 0, w - 1, h - 1)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    layers = [rand_f16_frame(rng, full, full, alpha="one" if k == 0 else "rand") for k in range(2)]
    want_chain = orc.chain_color_over(layers, m, orc.transfer_table(0), None)
    dl = [DeviceFrame.from_host(l) for l in layers]
    out = DeviceFrame(full, np.uint16)
    chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
    assert_same_f16(out.download().array, want_chain.array, "chain")
    # the triangle scaler: two cached tap tables
    src32 = HostFrame(full, np.float32, orc.half_to_float(layers[1].array))
    want = HostFrame((0, 0, 2 * w - 1, 2 * h - 1), np.float32)
    orc.lib().orc_scale_bilinear_f32(want.ref(), v2f(0, 0), src32.ref(), v2f(0, 0), v2f(2.0, 2.0))
    big = DeviceFrame((0, 0, 2 * w - 1, 2 * h - 1), np.uint16)
    _lib.check(cvs.cvs_scale_bilinear_f16_dev(big.ref(), v2f(0, 0), dl[1].ref(), v2f(0, 0), v2f(2.0, 2.0), None))
    assert_same_f16(big.download().array, orc.float_to_half(want.array), "scaler")
    # config 3 (pooled scratch on the fallback paths, the Lanczos taps)
    taps = synth.gaussian_taps(9, 1.5)
    want3 = _oracle_config3(orc, layers[1], (w // 2, h // 2), taps, 0.5, 0.5)
    small = DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16)
    _lib.check(cvs.cvs_blur_lanczos_f16_dev(small.ref(), dl[1].ref(), f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, None))
    assert_same_f16(small.download().array, want3.array, "config 3")
    # the display edge: a byte table composed from a transfer table
    want_bytes = np.empty((h, w), np.uint32)
    orc.lib().orc_frame_to_bytes(want_bytes.ctypes.data_as(C.POINTER(C.c_uint32)), layers[1].ref(), u16p(orc.transfer_table(3)), 0)
    dev_bytes = cvs.cvs_pool_malloc(w * h * 4, None)
    _lib.check(cvs.cvs_frame_to_bytes_dev(dev_bytes, dl[1].ref(), _lib.LUT_LINEAR_TO_SRGB, _lib.DISPLAY_RGBA8, None))
    got_bytes = np.empty((h, w), np.uint32)
    _lib.check(cvs.cvs_memcpy_d2h(got_bytes.ctypes.data, dev_bytes, w * h * 4, None))
    cvs.cvs_pool_free(dev_bytes, None)
    assert np.array_equal(got_bytes, want_bytes), "frame to bytes"
    for d in dl + [out, big, small]:
        d.free()
    return cvs.cvs_lut_device(_lib.LUT_REC709_TO_LINEAR_SCENE)


def test_device_contexts_keep_their_own_tables_and_pools(cvs, orc):
    """library state keyed by device.  On a one-GPU box two EXTRA contexts on device 0 stand in for two more
    GPUs: each must build its own transfer / tap / byte tables and scratch pool on first use (different device addresses), give
    the oracle's pixels, and leave the default context as it was."""
    assert cvs.cvs_current_context() >= 0
    home = cvs.cvs_current_context()
    table_home = _every_cached_table_once(cvs, orc, 1)
    a, b = cvs.cvs_context_open(0), cvs.cvs_context_open(0)
    assert a >= 0 and b >= 0 and len({home, a, b}) == 3, _lib.last_error()
    assert cvs.cvs_context_device(a) == 0 and cvs.cvs_context_device(b) == 0 and cvs.cvs_context_count() >= 3
    seen = {home: table_home}
    try:
        for ctx, seed in ((a, 2), (b, 3), (a, 4)):
            assert cvs.cvs_set_context(ctx) in (home, a, b)
            assert cvs.cvs_current_context() == ctx and cvs.cvs_current_device() == 0
            table = _every_cached_table_once(cvs, orc, seed)
            assert seen.setdefault(ctx, table) == table          # the same context: the same table, built once
        assert len(set(seen.values())) == 3                      # three contexts: three copies in HBM
    finally:
        assert cvs.cvs_set_context(-1) in (a, b)
    assert cvs.cvs_current_context() == home
    assert _every_cached_table_once(cvs, orc, 5) == table_home


def test_one_thread_per_context_renders_its_own_frames(cvs, orc):
    """Frame g belongs to context cvs_frame_owner(g, n); one thread per context (what VideoPullQueue(devices=...) does): the
    threads run at the same time, each in its own pool, caches and stream, and every frame equals the oracle's."""
    import threading
    ctxs = [cvs.cvs_context_open(0) for _ in range(2)]
    assert min(ctxs) >= 0
    frames = list(range(12))
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    w, h = 160, 90
    want = {g: orc.chain_color_over([synth.layer_frame(w, h, k, g) for k in range(2)], m, orc.transfer_table(0), None).array for g in frames}
    arith = cvs.cvs_get_arithmetic()
    got, errors = {}, []

    def worker(slot):
        try:
            assert cvs.cvs_set_context(ctxs[slot]) >= -1
            for g in frames:
                if cvs.cvs_frame_owner(g, len(ctxs)) != slot:
                    continue
                assert cvs.cvs_current_context() == ctxs[slot] and cvs.cvs_get_arithmetic() == arith
                dl = [DeviceFrame.from_host(synth.layer_frame(w, h, k, g)) for k in range(2)]
                out = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
                chain_color_over([(out, dl)], m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE)
                got[g] = out.download().array
                for d in dl + [out]:
                    d.free()
        except Exception as e:                                   # noqa: BLE001
            errors.append((slot, repr(e)))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(len(ctxs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert sorted(got) == frames
    for g in frames:
        assert_same_f16(got[g], want[g], "frame %d on context %d" % (g, g % len(ctxs)))
