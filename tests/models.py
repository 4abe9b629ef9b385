"""Independent numpy models used to cross-check the C oracle (NOT the oracle itself).

They state the arithmetic from first principles (IEEE-754 definitions, the formulas quoted in
SURVEY.md section 8a) in a different form from oracle/*.c, so that a slip in either shows up.
"""
import numpy as np


def f2h_rz_model(values):
    """f32 -> f16 by truncation, straight from the IEEE field definitions.

    Behaviour being modelled: src/cprocess/half.c:47-51 with genhalf.py:25-55 --
    e < -24 -> +-0; -24 <= e < -14 -> truncated subnormal; e > 15 -> +-Inf;
    exponent field 255 keeps the top 10 payload bits (so a NaN whose payload sits only in the
    low 13 bits becomes Inf)."""
    bits = np.ascontiguousarray(values, np.float32).view(np.uint32).astype(np.int64)
    sign = ((bits >> 31) & 1) << 15
    ef = (bits >> 23) & 0xFF
    e = ef - 127
    m = bits & 0x7FFFFF
    sig = m | 0x800000
    out = np.zeros(bits.shape, np.int64)
    normal = (e >= -14) & (e <= 15)
    out = np.where(normal, ((e + 15) << 10) | (m >> 13), out)
    sub = (e >= -24) & (e < -14)
    shift = np.clip(-e - 1, 0, 40)
    out = np.where(sub, sig >> shift, out)
    out = np.where((e > 15) & (ef != 255), 0x7C00, out)
    out = np.where(ef == 255, 0x7C00 + (m >> 13), out)
    return (out | sign).astype(np.uint16)


def h2f_ieee(codes):
    return np.ascontiguousarray(codes, np.uint16).view(np.float16).astype(np.float32)


def over_model(lower, lower_win, upper, upper_win, full, mix):
    """Per-pixel statement of un-premultiplied alpha-over on full-size f32 arrays (H, W, 4).

    Only valid where the reference's region walk is well defined: identical full windows and
    windows for which the `left` selector quirk (video_mix.c:265) picks the geometrically left
    frame.  Returns (pixels, window); pixels outside the window are left as in `lower`."""
    f32 = np.float32
    mix = f32(min(max(mix, 0.0), 1.0))
    out = lower.copy()

    def empty(w):
        return w[2] < w[0] or w[3] < w[1]

    def mask(w, shape):
        m = np.zeros(shape[:2], bool)
        if not empty(w):
            m[w[1] - full[1]: w[3] - full[1] + 1, w[0] - full[0]: w[2] - full[0] + 1] = True
        return m

    if empty(lower_win):
        if mix == 0:
            return out, (0, 0, -1, -1)
        w = (max(upper_win[0], full[0]), max(upper_win[1], full[1]), min(upper_win[2], full[2]), min(upper_win[3], full[3]))
        mk = mask(w, out.shape)
        out[mk] = upper[mk]
        out[mk, 3] = upper[mk, 3] * mix if mix != 1 else upper[mk, 3]
        return out, w
    if empty(upper_win) or mix == 0:
        return out, tuple(lower_win)
    outer = (max(min(lower_win[0], upper_win[0]), full[0]), max(min(lower_win[1], upper_win[1]), full[1]),
             min(max(lower_win[2], upper_win[2]), full[2]), min(max(lower_win[3], upper_win[3]), full[3]))
    ml, mu, mo = mask(lower_win, out.shape), mask(upper_win, out.shape), mask(outer, out.shape)
    both = ml & mu & mo
    only_u = mu & ~ml & mo
    neither = mo & ~ml & ~mu
    ab = (upper[..., 3] * mix).astype(f32)
    aa = (lower[..., 3] * (f32(1.0) - (upper[..., 3] * mix).astype(f32)).astype(f32)).astype(f32)
    a = (aa + ab).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        rgb = (((lower[..., :3] * aa[..., None]).astype(f32) + (upper[..., :3] * ab[..., None]).astype(f32)).astype(f32)
               / a[..., None]).astype(f32)
    blended = np.concatenate([rgb, a[..., None]], -1)
    blended[a == 0] = 0
    out[both] = blended[both]
    out[only_u] = upper[only_u]
    out[only_u, 3] = (upper[only_u, 3] * mix).astype(f32)
    out[neither] = 0
    return out, outer
