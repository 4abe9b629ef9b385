"""ctypes front-end of the CPU oracle (oracle/*.c -> oracle/liboracle.so).

TEST INFRASTRUCTURE.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg.  The product package (canvas_amd/) never imports this.
See oracle/oracle.h for the pinning status of each function.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from canvas_amd.abi import (box2i, fir_filter, rgba_frame_f16, rgba_frame_f32, v2f, video_source,
                            HostFrame)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False, arch="", out=None):
    """Compile the C restatement.  `out` lets bench.py build a -march=native copy elsewhere."""
    out = out or _SO
    srcs = [os.path.join(_HERE, f) for f in ("tables.c", "mix.c", "scale.c", "color.c", "dv.c", "oracle.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    cmd = ["make", "-C", _HERE, "-B", "OUT=" + out] + (["ARCH=" + arch] if arch else [])
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return out


class ws_item(C.Structure):
    _fields_ = [("x", C.c_int64), ("length", C.c_int64), ("z", C.c_int64), ("offset", C.c_int64),
                ("source", C.POINTER(video_source))]


def _bind(lib):
    P = C.POINTER
    u16p, f32p = P(C.c_uint16), P(C.c_float)
    sig = {
        "orc_half_to_float": (None, [f32p, u16p, C.c_int]),
        "orc_float_to_half": (None, [u16p, f32p, C.c_int]),
        "orc_half_to_float_fast": (None, [f32p, u16p, C.c_int]),
        "orc_float_to_half_fast": (None, [u16p, f32p, C.c_int]),
        "orc_half_lookup": (None, [u16p, u16p, u16p, C.c_int]),
        "orc_transfer_table": (u16p, [C.c_int]),
        "orc_transfer": (None, [C.c_int, u16p, u16p, C.c_size_t]),
        "orc_gamma45_ramp": (P(C.c_uint8), []),
        "orc_fir_triangle": (None, [C.c_float, C.c_float, P(fir_filter)]),
        "orc_fir_lanczos": (None, [C.c_float, C.c_int, C.c_float, P(fir_filter)]),
        "orc_fir_free": (None, [P(fir_filter)]),
        "orc_get_frame_f16": (None, [P(video_source), C.c_int, P(rgba_frame_f16)]),
        "orc_get_frame_f32": (None, [P(video_source), C.c_int, P(rgba_frame_f32)]),
        "orc_copy_frame_f16": (None, [P(rgba_frame_f16), P(rgba_frame_f16)]),
        "orc_copy_frame_alpha_f32": (None, [P(rgba_frame_f32), P(rgba_frame_f32), C.c_float]),
        "orc_mix_cross_f32": (None, [P(rgba_frame_f32), P(rgba_frame_f32), P(rgba_frame_f32), C.c_float]),
        "orc_mix_cross_f32_pull": (None, [P(rgba_frame_f32), P(video_source), C.c_int, P(video_source), C.c_int, C.c_float]),
        "orc_mix_over_f32": (None, [P(rgba_frame_f32), P(rgba_frame_f32), C.c_float]),
        "orc_scale_bilinear_f32": (None, [P(rgba_frame_f32), v2f, P(rgba_frame_f32), v2f, v2f]),
        "orc_scale_bilinear_f32_pull": (None, [P(rgba_frame_f32), v2f, P(video_source), C.c_int, P(box2i), v2f, v2f]),
        "orc_color_matrix_f16": (None, [P(rgba_frame_f16), f32p, u16p, u16p]),
        "orc_color_rgb_to_xyz_sdtv": (None, [P(rgba_frame_f16)]),
        "orc_color_xyz_to_srgb": (None, [P(rgba_frame_f16)]),
        "orc_gain_offset_f16": (None, [P(rgba_frame_f16), P(rgba_frame_f16), C.c_float, C.c_float]),
        "orc_solid_f16": (None, [P(rgba_frame_f16), P(box2i), f32p]),
        "orc_solid_f32": (None, [P(rgba_frame_f32), P(box2i), f32p]),
        "orc_fir_blur_f32": (None, [P(rgba_frame_f32), P(rgba_frame_f32), f32p, C.c_int]),
        "orc_resample_lanczos_f32": (None, [P(rgba_frame_f32), P(rgba_frame_f32), C.c_float, C.c_float, C.c_int]),
        "orc_workspace_get_frame_f32": (None, [P(ws_item), C.c_int, C.c_int, P(rgba_frame_f32)]),
        "orc_reconstruct_dv": (None, [P(rgba_frame_f16), P(C.c_void_p), P(C.c_int)]),
        "orc_subsample_dv": (None, [P(C.c_void_p), P(C.c_int), P(rgba_frame_f16)]),
        "orc_frame_to_bytes": (None, [P(C.c_uint32), P(rgba_frame_f16), u16p, C.c_int]),
        "orc_widget_ramp": (None, [P(C.c_uint8), C.c_float]),
        "orc_pulldown23_frames": (C.c_int, [C.c_int, C.c_int, P(C.c_int), P(C.c_int)]),
        "orc_weave_fields_f16": (None, [P(rgba_frame_f16), P(rgba_frame_f16)]),
        "orc_frame_to_rgba8_intent": (None, [P(C.c_uint32), P(rgba_frame_f16), u16p, C.c_float]),
        "orc_chain_color_over_f16": (None, [P(rgba_frame_f16), P(P(rgba_frame_f16)), C.c_int, f32p, u16p, u16p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


_lib = None
_SO_FMA = os.path.join(_HERE, "liboracle_fma.so")
_lib_fma = None


def lib(path=None):
    global _lib
    if path is not None:
        return _bind(C.CDLL(path))
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_SO))
    return _lib


def build_fma(force=False):
    """The clang / contraction-on flavour (oracle/Makefile, target fma): what the reference's preferred build rounds like."""
    srcs = [os.path.join(_HERE, f) for f in ("tables.c", "mix.c", "scale.c", "color.c", "dv.c", "oracle.h")]
    if not force and os.path.exists(_SO_FMA) and all(os.path.getmtime(_SO_FMA) >= os.path.getmtime(s) for s in srcs):
        return _SO_FMA
    subprocess.run(["make", "-C", _HERE, "-B", "fma"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return _SO_FMA


class flavour:
    """`with oracle.flavour("fma"): ...` -- every convenience of this module runs on the contraction-on build inside the
    block (the gcc / no-contraction build is the default everywhere else)."""

    def __init__(self, name):
        assert name in ("gcc", "fma")
        self.name = name

    def __enter__(self):
        global _lib, _lib_fma
        self.saved = _lib
        if self.name == "fma":
            if _lib_fma is None:
                if not os.path.exists(_SO_FMA):
                    build_fma()
                _lib_fma = _bind(C.CDLL(_SO_FMA))
            _lib = _lib_fma
        else:
            _lib = None
            lib()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


# ---- numpy conveniences -------------------------------------------------------------------

def _u16(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def half_to_float(codes):
    codes = np.ascontiguousarray(codes, np.uint16)
    out = np.empty(codes.shape, np.float32)
    lib().orc_half_to_float(_f32(out), _u16(codes), codes.size)
    return out


def float_to_half(values):
    values = np.ascontiguousarray(values, np.float32)
    out = np.empty(values.shape, np.uint16)
    lib().orc_float_to_half(_u16(out), _f32(values), values.size)
    return out


def half_lookup(table, codes):
    table = np.ascontiguousarray(table, np.uint16)
    codes = np.ascontiguousarray(codes, np.uint16)
    out = np.empty(codes.shape, np.uint16)
    lib().orc_half_lookup(_u16(table), _u16(out), _u16(codes), codes.size)
    return out


def transfer_table(which):
    p = lib().orc_transfer_table(which)
    return np.ctypeslib.as_array(p, shape=(65536,)).copy()


def gamma45_ramp():
    return np.ctypeslib.as_array(lib().orc_gamma45_ramp(), shape=(65536,)).copy()


def _fir(fn, *args):
    f = fir_filter(None, 0, 0)
    fn(*args, C.byref(f))
    taps = np.ctypeslib.as_array(f.coeff, shape=(f.width,)).copy()
    res = (taps, f.center)
    lib().orc_fir_free(C.byref(f))
    return res


def fir_triangle(sub, offset):
    return _fir(lib().orc_fir_triangle, C.c_float(sub), C.c_float(offset))


def fir_lanczos(sub, kernel_size, offset):
    return _fir(lib().orc_fir_lanczos, C.c_float(sub), kernel_size, C.c_float(offset))


def chain_color_over(layers, matrix, pre_lut=None, post_lut=None, full_window=None):
    """BASELINE config 2 on host frames; returns a new f16 HostFrame."""
    fw = full_window or layers[0].full_window
    out = HostFrame(fw, np.uint16)
    arr = (C.POINTER(rgba_frame_f16) * len(layers))(*[C.pointer(l.c) for l in layers])
    m = None if matrix is None else np.ascontiguousarray(matrix, np.float32).reshape(9)    # None: plain stack
    pre_a = None if pre_lut is None else np.ascontiguousarray(pre_lut, np.uint16)
    post_a = None if post_lut is None else np.ascontiguousarray(post_lut, np.uint16)
    lib().orc_chain_color_over_f16(out.ref(), arr, len(layers), None if m is None else _f32(m),
                                   None if pre_a is None else _u16(pre_a),
                                   None if post_a is None else _u16(post_a))
    return out
