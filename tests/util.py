"""Shared helpers for the parity tests."""
import ctypes as C

import numpy as np

from canvas_amd.abi import HostFrame, box2i


def canon_f32(a):
    """Bit pattern with the two things the reference does not pin folded away: the sign of zero
    (its build uses -fno-signed-zeros, SConstruct:82-83) and NaN payload/sign (x86 and gfx950
    produce different default NaNs)."""
    a = np.ascontiguousarray(a, np.float32)
    bits = a.view(np.uint32).copy()
    bits[a == 0] = 0
    bits[np.isnan(a)] = 0x7FC00000
    return bits


def canon_f16(codes):
    codes = np.ascontiguousarray(codes, np.uint16).copy()
    mag = codes & 0x7FFF
    codes[mag == 0] = 0
    codes[mag > 0x7C00] = 0x7E00
    return codes


def assert_same_f32(got, want, what=""):
    g, w = canon_f32(got), canon_f32(want)
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        i = tuple(bad[0])
        raise AssertionError("%s: %d of %d values differ; first at %s: got %r want %r" % (
            what, len(bad), g.size, i, np.asarray(got)[i], np.asarray(want)[i]))


def assert_same_f16(got, want, what=""):
    g, w = canon_f16(got), canon_f16(want)
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        i = tuple(bad[0])
        raise AssertionError("%s: %d of %d codes differ; first at %s: got 0x%04x want 0x%04x" % (
            what, len(bad), g.size, i, np.asarray(got)[i], np.asarray(want)[i]))


def same_window(a, b):
    """Equal boxes; any two empty boxes count as equal (only emptiness is contractual)."""
    if a.is_empty() and b.is_empty():
        return True
    return a.tuple() == b.tuple()


def rand_f32_frame(rng, full, win=None, alpha="rand", lo=0.0, hi=1.0):
    fw = box2i.of(*full)
    arr = rng.uniform(lo, hi, (fw.height, fw.width, 4)).astype(np.float32)
    if alpha == "one":
        arr[..., 3] = 1
    elif alpha == "zero":
        arr[..., 3] = 0
    elif alpha == "mixed":
        a = rng.uniform(0, 1, arr.shape[:2]).astype(np.float32)
        a[rng.uniform(size=a.shape) < 0.2] = 0
        a[rng.uniform(size=a.shape) < 0.2] = 1
        arr[..., 3] = a
    return HostFrame(full, np.float32, arr, full if win is None else win)


def rand_f16_frame(rng, full, win=None, orc=None, alpha="rand"):
    f = rand_f32_frame(rng, full, win, alpha)
    from canvas_amd.synth import truncate_to_half
    return HostFrame(full, np.uint16, truncate_to_half(f.array), full if win is None else win)


def f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def u16p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def oracle_graph(orc, layers, matrix, pre_table, post_table, taps):
    """BASELINE config 5 with the CPU oracle's nodes: colour filter on an f16 copy of layer 0, blur node pulling it
    as f32, workspace stack (base fetched as f32, every further layer widened and blended over at mix 1.0),
    f16 pull of the result."""
    graded = layers[0].copy()
    m = np.ascontiguousarray(matrix, np.float32).reshape(9)
    taps = np.ascontiguousarray(taps, np.float32)
    orc.lib().orc_color_matrix_f16(graded.ref(), f32p(m), None if pre_table is None else u16p(pre_table),
                                   None if post_table is None else u16p(post_table))
    return oracle_blur_over(orc, graded, taps, layers[1:])


def oracle_blur_over(orc, source, taps, overlays, out_full=None):
    full = out_full or source.full_window
    wide = HostFrame(source.full_window, np.float32, orc.half_to_float(source.array), source.current_window)
    acc = HostFrame(full, np.float32)
    orc.lib().orc_fir_blur_f32(acc.ref(), wide.ref(), f32p(taps), len(taps))
    for ov in overlays:
        ov32 = HostFrame(ov.full_window, np.float32, orc.half_to_float(ov.array), ov.current_window)
        orc.lib().orc_mix_over_f32(acc.ref(), ov32.ref(), C.c_float(1.0))
    return HostFrame(full, np.uint16, orc.float_to_half(acc.array), acc.current_window)
