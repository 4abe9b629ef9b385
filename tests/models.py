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


# ---------------------------------------------------------------- more independent statements (full windows unless said otherwise)

F32 = np.float32


def _div_or_zero(num, den):
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (num / den).astype(F32)
    return np.where(den != 0, q, F32(0))


def cross_model(a, b, mix_b):
    """video_mix.c:193-205 on whole (H, W, 4) f32 arrays: weights 1 - m and m, un-premultiplied normalisation."""
    mix_b = F32(min(max(mix_b, 0.0), 1.0))
    mix_a = F32(F32(1.0) - mix_b)
    aa, ab = (a[..., 3] * mix_a).astype(F32), (b[..., 3] * mix_b).astype(F32)
    alpha = (aa + ab).astype(F32)
    out = np.zeros_like(a)
    for c in range(3):
        num = ((a[..., c] * aa).astype(F32) + (b[..., c] * ab).astype(F32)).astype(F32)
        out[..., c] = _div_or_zero(num, alpha)
    out[..., 3] = alpha
    out[alpha == 0] = 0
    return out


def gain_offset_model(codes, gain, offset):
    """video_filter.c:34-39 as the library defines its rounding: widen, c * gain + offset in two f32 steps, truncate."""
    v = h2f_ieee(codes)
    out = v.copy()
    with np.errstate(invalid="ignore", over="ignore"):
        for c in range(3):
            out[..., c] = ((v[..., c] * F32(gain)).astype(F32) + F32(offset)).astype(F32)
    return f2h_rz_model(out)


def bytes_model(codes, ramp, pre, premultiplied_argb):
    """widget_gl.c:291-307 / writeVideo.c:328-340 (bytes r,g,b,a) and RgbaFrameF16.c:114-149 (premultiplied ARGB32)."""
    c = codes if pre is None else pre[codes]
    r, g, b, a = (ramp[c[..., k]].astype(np.uint32) for k in range(4))
    if not premultiplied_argb:
        return r | (g << 8) | (b << 16) | (a << 24)
    return (a << 24) | ((((r * a) >> 8) & 0xFF) << 16) | ((((g * a) >> 8) & 0xFF) << 8) | (((b * a) >> 8) & 0xFF)


def weave_model(frame, full, cur, other, ocur):
    """Pulldown23RemovalFilter.c:88-104 as flat-array arithmetic: `other` is a packed buffer for `cur`; every even row i of
    cur takes `width` pixels from flat index (i - cur.y0) * width - cur.x0 (the x = 0 addressing of :101); indices outside
    the buffer and pixels outside `ocur` read as zero."""
    out = frame.copy()
    x0, y0, x1, y1 = cur
    width, height = x1 - x0 + 1, y1 - y0 + 1
    if width <= 0 or height <= 0:
        return out
    flat = other.reshape(-1, 4).copy()
    ys, xs = np.divmod(np.arange(width * height), width)
    inside = (xs + x0 >= ocur[0]) & (xs + x0 <= ocur[2]) & (ys + y0 >= ocur[1]) & (ys + y0 <= ocur[3])
    flat[~inside] = 0
    for i in range((y0 + 1) & ~1, y1 + 1, 2):
        idx = (i - y0) * width - x0 + np.arange(width)
        ok = (idx >= 0) & (idx < width * height)
        row = np.zeros((width, 4), frame.dtype)
        row[ok] = flat[idx[ok]]
        out[i - full[1], x0 - full[0]:x1 - full[0] + 1] = row
    return out


def widget_ramp_model(intent):
    """widget_gl.c:955-968: ramp[i] = (uint8_t) lrint(clampf(powf(h2f(i), intent) * 255, 0, 255)); clampf sends NaN to 0."""
    with np.errstate(all="ignore"):
        x = np.arange(65536, dtype=np.uint16).view(np.float16).astype(F32)
        v = (np.power(x, F32(intent)).astype(F32) * F32(255.0)).astype(F32)
        v = np.where(v > F32(0.0), v, F32(0.0))                  # framework.h clampf: max first (NaN -> lo), then min
        v = np.where(v < F32(255.0), v, F32(255.0))
        return np.rint(v).astype(np.uint8)                       # lrint: to nearest, ties to even


def _fir_pass(src, valid, taps_for_line, axis):
    """One gather pass along `axis` (0: rows, 1: columns) over a whole array.  taps_for_line(t) -> [(source index, weight)]
    ascending; `valid` marks the source lines that exist (others are skipped taps).  Sums start at 0.0f."""
    n = src.shape[axis]
    out = np.zeros_like(src)
    for t in range(n):
        acc = np.zeros_like(np.take(src, 0, axis=axis))
        for s, w in taps_for_line(t):
            if 0 <= s < n and valid[s]:
                acc = (acc + (np.take(src, s, axis=axis) * F32(w)).astype(F32)).astype(F32)
        if axis == 0:
            out[t] = acc
        else:
            out[:, t] = acc
    return out


def blur_model(src, taps):
    """Separable FIR on a whole frame whose window is the whole frame: centre ntaps // 2, x pass then y pass."""
    c = len(taps) // 2
    lines = lambda t: [(t - c + k, taps[k]) for k in range(len(taps))]          # noqa: E731
    h = _fir_pass(src, np.ones(src.shape[1], bool), lines, 1)
    return _fir_pass(h, np.ones(src.shape[0], bool), lines, 0)


def lanczos_model(src, tsize, factor_x, factor_y, kernel_size, tap_generator):
    """Gather resample of a whole frame to tsize = (w, h), origin 0 on both sides: per target line the taps of
    tap_generator(factor, kernel_size, frac(t / factor)) -> (taps, centre), x pass then y pass."""
    def resample(a, n_out, factor, axis):
        n_in = a.shape[axis]
        shape = list(a.shape)
        shape[axis] = n_out
        out = np.zeros(shape, F32)
        for t in range(n_out):
            centre_f = F32(F32(t) / F32(factor))
            centre = int(np.floor(centre_f))
            taps, tc = tap_generator(float(factor), kernel_size, float(F32(centre_f - F32(centre))))
            acc = np.zeros_like(np.take(a, 0, axis=axis))
            for k, w in enumerate(taps):
                s = centre - tc + k
                if 0 <= s < n_in:
                    acc = (acc + (np.take(a, s, axis=axis) * F32(w)).astype(F32)).astype(F32)
            if axis == 0:
                out[t] = acc
            else:
                out[:, t] = acc
        return out
    return resample(resample(src, tsize[0], factor_x, 1), tsize[1], factor_y, 0)


def dv_reconstruct_model(y, cb, cr, lut, triangle):
    """video_reconstruct.c:50-137 on whole planes (480x720, 480x180, 480x180) -> half codes (480, 720, 4); the frame's
    row 0 is picture line 0 here (the caller places it at y = -1).  triangle = (taps, centre) of filter_createTriangle(4, 0)."""
    taps, centre = triangle
    cbf = ((cb.astype(F32) - F32(128)) / F32(224)).astype(F32)
    crf = ((cr.astype(F32) - F32(128)) / F32(224)).astype(F32)
    pb, pr = np.zeros((480, 720), F32), np.zeros((480, 720), F32)
    for xs in range(180):                                  # scatter in ascending sample order, like the reference
        for k, w in enumerate(taps):
            i = xs * 4 - centre + k
            if 0 <= i < 720:
                pb[:, i] = (pb[:, i] + (cbf[:, xs] * F32(w)).astype(F32)).astype(F32)
                pr[:, i] = (pr[:, i] + (crf[:, xs] * F32(w)).astype(F32)).astype(F32)
    yf = ((y.astype(F32) - F32(16)) / F32(219)).astype(F32)
    m = [[1.0, 0.0, 1.5748], [1.0, -0.187324, -0.468124], [1.0, 1.8556, 0.0]]
    out = np.zeros((480, 720, 4), F32)
    for c in range(3):
        out[..., c] = (((yf * F32(m[c][0])).astype(F32) + (pb * F32(m[c][1])).astype(F32)).astype(F32) + (pr * F32(m[c][2])).astype(F32)).astype(F32)
    out[..., 3] = 1
    return lut[f2h_rz_model(out)]


def dv_subsample_model(codes, lut, triangle):
    """video_subsample.c:99-187 on a whole (480, 720, 4) frame of half codes -> (Y, Cb, Cr) planes.
    triangle = (taps, centre) of filter_createTriangle(1/4, 0).  (uint8)float = truncate to int32, keep the low byte."""
    taps, centre = triangle
    v = h2f_ieee(lut[codes])
    def dot(row):
        return (((v[..., 0] * F32(row[0])).astype(F32) + (v[..., 1] * F32(row[1])).astype(F32)).astype(F32) + (v[..., 2] * F32(row[2])).astype(F32)).astype(F32)
    low = lambda f: (f.astype(np.int64) & 0xFF).astype(np.uint8)         # noqa: E731
    yy = low(((dot((0.2126, 0.7152, 0.0722)) * F32(219)).astype(F32) + F32(16)).astype(F32))
    pb, pr = dot((-0.114572, -0.385428, 0.5)), dot((0.5, -0.454153, -0.045847))
    cb, cr = np.zeros((480, 180), F32), np.zeros((480, 180), F32)
    for tx in range(180):
        for k, w in enumerate(taps):
            sx = tx * 4 - centre + k
            if 0 <= sx < 720:
                cb[:, tx] = (cb[:, tx] + (pb[:, sx] * F32(w)).astype(F32)).astype(F32)
                cr[:, tx] = (cr[:, tx] + (pr[:, sx] * F32(w)).astype(F32)).astype(F32)
    return yy, low(((cb * F32(224)).astype(F32) + F32(128)).astype(F32)), low(((cr * F32(224)).astype(F32) + F32(128)).astype(F32))


def scale_model(src, sfull, scur, tfull, tp, sp, fac, triangle):
    """video_scale.c:231-286 in whole-array form.  src: (H, W, 4) f32 over sfull; scur: the source's current window;
    returns (array over tfull, window).  triangle(sub, offset) -> (taps, centre) (filter_createTriangle).
    Statement: per axis, target lines start at zero; upscaling scatters every source line onto the target lines its
    triangle touches, downscaling gathers; the axis with the smaller factor goes first through an f32 frame whose
    window is computed with `* factor` (video_scale.c:256-262); the window reported is the span of lines touched."""
    def empty(w):
        return w[2] < w[0] or w[3] < w[1]

    def one_pass(a, afull, acur, bfull, tmin, smin, factor, axis):          # axis 0: along y, 1: along x
        bh, bw = bfull[3] - bfull[1] + 1, bfull[2] - bfull[0] + 1
        out = np.zeros((max(bh, 0), max(bw, 0), 4), F32)
        o_lo = max(acur[axis], bfull[axis])                                  # the other axis (x for a y pass, y for an x pass) is clipped
        o_hi = min(acur[2 + axis], bfull[2 + axis])
        s0, s1 = acur[0 if axis else 1], acur[2 if axis else 3]
        t0, t1 = bfull[0 if axis else 1], bfull[2 if axis else 3]
        used = []
        if not (factor == 1.0 and tmin == smin):
            pairs = []                                                        # (target line, source line, weight) in the reference's order
            if factor > 1.0:
                for s in range(s0, s1 + 1):
                    centre_f = F32(F32(F32(s) - F32(smin)) * F32(factor) + F32(tmin))
                    centre = int(np.floor(centre_f))
                    taps, tc = triangle(float(factor), float(F32(centre_f - F32(centre))))
                    pairs += [(centre - tc + k, s, w) for k, w in enumerate(taps)]
            else:
                for t in range(t0, t1 + 1):
                    centre_f = F32(F32(F32(t) - F32(tmin)) / F32(factor) + F32(smin))
                    centre = int(np.floor(centre_f))
                    taps, tc = triangle(float(factor), float(F32(centre_f - F32(centre))))
                    pairs += [(t, centre - tc + k, w) for k, w in enumerate(taps)]
            for t, s, w in pairs:
                if t < t0 or t > t1 or s < s0 or s > s1:
                    continue
                if axis == 0 or o_lo <= o_hi:
                    used.append(t)
                if o_lo > o_hi:
                    continue
                if axis:
                    tgt = out[o_lo - bfull[1]: o_hi - bfull[1] + 1, t - bfull[0]]
                    line = a[o_lo - afull[1]: o_hi - afull[1] + 1, s - afull[0]]
                else:
                    tgt = out[t - bfull[1], o_lo - bfull[0]: o_hi - bfull[0] + 1]
                    line = a[s - afull[1], o_lo - afull[0]: o_hi - afull[0] + 1]
                tgt[...] = (tgt + (line * F32(w)).astype(F32)).astype(F32)
            lo, hi = (min(used), max(used)) if used else (2 ** 31 - 1, -2 ** 31)
            win = (lo, o_lo, hi, o_hi) if axis else (o_lo, lo, o_hi, hi)
            return out, win
        # identity on this axis: a clipped copy (video_copy_frame_alpha_f32 with alpha 1)
        win = (max(acur[0], bfull[0]), max(acur[1], bfull[1]), min(acur[2], bfull[2]), min(acur[3], bfull[3]))
        if not empty(win):
            out[win[1] - bfull[1]: win[3] - bfull[1] + 1, win[0] - bfull[0]: win[2] - bfull[0] + 1] = \
                a[win[1] - afull[1]: win[3] - afull[1] + 1, win[0] - afull[0]: win[2] - afull[0] + 1]
        return out, win

    fx, fy = F32(fac[0]), F32(fac[1])
    if fx == 1.0 and tp[0] == sp[0]:
        if fy == 1.0 and tp[1] == sp[1]:
            return one_pass(src, sfull, scur, tfull, 0.0, 0.0, 1.0, 0)
        return one_pass(src, sfull, scur, tfull, tp[1], sp[1], fy, 0)
    if fy == 1.0 and tp[1] == sp[1]:
        return one_pass(src, sfull, scur, tfull, tp[0], sp[0], fx, 1)
    x_first = fx < fy
    if x_first:
        mid = (int(F32(F32(sp[0]) - F32(F32(tp[0]) - F32(tfull[0])) * fx)), scur[1], int(F32(F32(sp[0]) + F32(F32(tfull[2]) - F32(tp[0])) * fx)), scur[3])
    else:
        mid = (scur[0], int(F32(F32(sp[1]) - F32(F32(tp[1]) - F32(tfull[1])) * fy)), scur[2], int(F32(F32(sp[1]) + F32(F32(tfull[3]) - F32(tp[1])) * fy)))
    mid = (max(mid[0], tfull[0]), max(mid[1], tfull[1]), min(mid[2], tfull[2]), min(mid[3], tfull[3]))
    if x_first:
        m, mwin = one_pass(src, sfull, scur, mid, tp[0], sp[0], fx, 1)
        return one_pass(m, mid, mwin, tfull, tp[1], sp[1], fy, 0)
    m, mwin = one_pass(src, sfull, scur, mid, tp[1], sp[1], fy, 0)
    return one_pass(m, mid, mwin, tfull, tp[0], sp[0], fx, 1)
