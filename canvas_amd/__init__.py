"""canvas_amd -- MI355X (gfx950) implementation of Canvas's per-pixel f16 RGBA video path.

Layout
  csrc/kernels/   hand-written HIP kernels (gfx950 only)
  csrc/host/      host C behind the C-ABI of include/canvas_hip.h
  libcanvas_hip.so  the built library (git-ignored; `__graft_entry__.build()` makes it)
  abi.py, _lib.py ctypes mirror of the C-ABI
  device.py       device-resident frames for Python callers
  synth.py        the synthetic-input generator BASELINE.md section 4 defines
  shard.py        frame -> GPU round-robin driver (one process per GPU)

The pixel path has no CPU implementation: without the shared library, or without a HIP device,
every entry point fails loudly.
"""
from .abi import HostFrame, box2i, v2f  # noqa: F401

REC709_RGB_TO_YPBPR = (  # src/cprocess/video_subsample.c:104-108, column-major as color.c passes matrices
    0.2126, -0.114572, 0.5,
    0.7152, -0.385428, -0.454153,
    0.0722, 0.5, -0.045847,
)
