"""Measurement scripts that need the diagnostic build of the library (make -C canvas_amd/csrc diag ->
tools/bin/libcanvas_hip_diag.so: the CVS_* knobs and the timing-only kernel variants) call use_diag_library() before
the first canvas_amd._lib.load().  The package itself never loads that build."""
import os


def use_diag_library():
    from canvas_amd import _lib
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libcanvas_hip_diag.so")
    if not os.path.exists(path):
        raise SystemExit("%s missing: make -C canvas_amd/csrc diag" % path)
    _lib.LIB_PATH = path
    return path
