#!/usr/bin/env python3
"""Regenerates tests/golden/cprocess_small.npz from the CPU oracle (oracle/*.c).

What these vectors are -- and are not.  The reference cannot be built or imported in this image
(DESIGN.md section 2), so these are NOT outputs of the reference: they freeze the oracle's answers on small
seeded cases so that (a) a later edit of the oracle that changes any bit is caught on CPU
(tests/test_golden.py) and (b) the GPU parity tests have a second, file-based anchor that does not depend
on the oracle being rebuilt on the GPU box.  The reference's own known-answer tests are asserted
separately, from their published expected values (tests/test_oracle_pins.py).

  python tests/golden/make_golden.py          # rewrites the .npz next to this script
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from canvas_amd import REC709_RGB_TO_YPBPR, synth  # noqa: E402
from canvas_amd.abi import HostFrame, v2f  # noqa: E402


def f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def cases():
    lib = oracle.lib()
    out = {}
    rng = np.random.default_rng(2026)

    probe = np.concatenate([rng.uniform(-4, 4, 4096), rng.uniform(-70000, 70000, 1024),
                            rng.uniform(-1, 1, 1024) * 2.0 ** rng.integers(-30, -10, 1024)]).astype(np.float32)
    out["f2h_in"], out["f2h_out"] = probe, oracle.float_to_half(probe)
    for k in range(4):
        out["lut%d" % k] = oracle.transfer_table(k)
    out["gamma45"] = oracle.gamma45_ramp()
    for name, fn, args in [("tri_0.5_0", oracle.fir_triangle, (0.5, 0.0)), ("tri_0.25_0.5", oracle.fir_triangle, (0.25, 0.5)),
                           ("tri_2_0.25", oracle.fir_triangle, (2.0, 0.25)), ("lan3_0.5_0", oracle.fir_lanczos, (0.5, 3, 0.0)),
                           ("lan3_0.5_0.25", oracle.fir_lanczos, (0.5, 3, 0.25)), ("lan3_2_0.5", oracle.fir_lanczos, (2.0, 3, 0.5))]:
        taps, centre = fn(*args)
        out["fir_" + name] = taps
        out["fir_" + name + "_centre"] = np.array([centre])

    # config 2 at 64x36, 2..4 layers
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    for nl in (2, 3, 4):
        layers = [synth.layer_frame(64, 36, k, 0) for k in range(nl)]
        out["chain%d" % nl] = oracle.chain_color_over(layers, m, oracle.transfer_table(0), None).array

    # over / cross on 24x12 with nested and partial windows
    full = (0, 0, 23, 11)
    for tag, pw, qw in [("full", full, full), ("nested", full, (3, 2, 10, 6)), ("partial", (2, 1, 12, 7), (6, 4, 20, 10))]:
        a = HostFrame(full, np.float32, rng.uniform(0, 1, (12, 24, 4)).astype(np.float32), pw)
        b = HostFrame(full, np.float32, rng.uniform(0, 1, (12, 24, 4)).astype(np.float32), qw)
        out["mix_%s_a" % tag], out["mix_%s_b" % tag] = a.array.copy(), b.array.copy()
        o = a.copy()
        lib.orc_mix_over_f32(o.ref(), b.ref(), C.c_float(0.7))
        out["over_%s" % tag], out["over_%s_win" % tag] = o.array, np.array(o.current_window.tuple())
        x = HostFrame(full, np.float32)
        lib.orc_mix_cross_f32(x.ref(), a.ref(), b.ref(), C.c_float(0.3))
        out["cross_%s" % tag], out["cross_%s_win" % tag] = x.array, np.array(x.current_window.tuple())

    # triangle scaler 2x up and 0.5x down (the partial-coverage case), colour functions, config 3 pipeline
    src = HostFrame((0, 0, 15, 8), np.float32, rng.uniform(0, 1, (9, 16, 4)).astype(np.float32))
    out["scale_src"] = src.array.copy()
    up = HostFrame((0, 0, 31, 17), np.float32)
    lib.orc_scale_bilinear_f32(up.ref(), v2f(0, 0), src.ref(), v2f(0, 0), v2f(2.0, 2.0))
    out["scale_up"], out["scale_up_win"] = up.array, np.array(up.current_window.tuple())
    big = HostFrame((0, 0, 31, 17), np.float32, rng.uniform(0, 1, (18, 32, 4)).astype(np.float32))
    out["scale_big"] = big.array.copy()
    down = HostFrame((0, 0, 15, 8), np.float32)
    lib.orc_scale_bilinear_f32(down.ref(), v2f(0, 0), big.ref(), v2f(0, 0), v2f(0.5, 0.5))
    out["scale_down"], out["scale_down_win"] = down.array, np.array(down.current_window.tuple())

    h = synth.layer_frame(40, 20, 1, 3)
    out["color_in"] = h.array.copy()
    c1, c2 = h.copy(), h.copy()
    lib.orc_color_rgb_to_xyz_sdtv(c1.ref())
    lib.orc_color_xyz_to_srgb(c2.ref())
    out["color_rgb_to_xyz"], out["color_xyz_to_srgb"] = c1.array, c2.array

    layer = synth.layer_frame(96, 54, 1, 0)
    taps = synth.gaussian_taps(9, 1.5)
    s32 = HostFrame(layer.full_window, np.float32, oracle.half_to_float(layer.array))
    bl = HostFrame(layer.full_window, np.float32)
    lib.orc_fir_blur_f32(bl.ref(), s32.ref(), f32p(taps), 9)
    sm = HostFrame((0, 0, 47, 26), np.float32)
    lib.orc_resample_lanczos_f32(sm.ref(), bl.ref(), C.c_float(0.5), C.c_float(0.5), 3)
    out["config3_96x54"] = oracle.float_to_half(sm.array)
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cprocess_small.npz")
    np.savez_compressed(path, **cases())
    print("wrote", path, os.path.getsize(path), "bytes")
