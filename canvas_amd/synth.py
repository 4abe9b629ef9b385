"""Synthetic inputs as BASELINE.md section 4 / SURVEY.md section 8(d) define them.

Layer k of frame i: f16 RGBA from a counter-based RNG (Philox) seeded 0xC0FFEE + 1000*k + i;
r,g,b ~ U[0,1) truncated to half; alpha == 1.0 on layer 0 and ~ U[0,1) on layers >= 1 (so the
divide of the over operator is always live); full_window = current_window = (0,0)-(W-1,H-1).
"""
import os

import numpy as np

from .abi import HostFrame

SEED_BASE = 0xC0FFEE


def truncate_to_half(x):
    """f32 -> f16 codes, rounding toward zero (finite, in-range values only)."""
    x = np.ascontiguousarray(x, np.float32)
    h = x.astype(np.float16)
    over = np.abs(h.astype(np.float32)) > np.abs(x)
    codes = h.view(np.uint16).copy()
    codes[over] -= 1            # one code toward zero; sign bit is on top, magnitude below
    return codes


def layer_pixels(width, height, layer, frame, opaque_base=True):
    """opaque_base=False is NOT the BASELINE input: it gives layer 0 a random alpha too, so that the
    over operator's divides see denominators other than 1.0 (bench.py --translucent-base).
    CANVAS_SYNTH_CACHE=<dir>: keep generated frames as .npy there (measurement scripts that start the same
    workload in many processes; the cached file holds exactly what the generator returns)."""
    cache = os.environ.get("CANVAS_SYNTH_CACHE")
    path = None
    if cache:
        path = os.path.join(cache, "synth_%dx%d_l%d_f%d_%d.npy" % (width, height, layer, frame, int(bool(opaque_base))))
        if os.path.exists(path):
            return np.load(path)
    rng = np.random.Generator(np.random.Philox(SEED_BASE + 1000 * layer + frame))
    px = rng.random((height, width, 4), dtype=np.float32)
    codes = truncate_to_half(px)
    if layer == 0 and opaque_base:
        codes[..., 3] = 0x3C00
    if path:
        os.makedirs(cache, exist_ok=True)
        tmp = "%s.%d.tmp.npy" % (path, os.getpid())
        np.save(tmp, codes)
        os.replace(tmp, path)
    return codes


def layer_frame(width, height, layer, frame):
    return HostFrame((0, 0, width - 1, height - 1), np.uint16, layer_pixels(width, height, layer, frame))


def gaussian_taps(ntaps=9, sigma=1.5):
    """Config 3's blur: taps normalised in f32."""
    c = ntaps // 2
    x = np.arange(ntaps, dtype=np.float32) - np.float32(c)
    t = np.exp(-(x * x) / np.float32(2.0 * sigma * sigma)).astype(np.float32)
    return (t / t.sum(dtype=np.float32)).astype(np.float32)
