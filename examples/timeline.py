#!/usr/bin/env python3
"""A small timeline built from the same `fluggo.media.process` objects the Canvas editor creates
(fluggo/editor/graph/video.py): two clips cut together with a crossfade, a picture-in-picture layer scaled down over
them, a title bar on top.  Every frame is rendered on the GPU; only the 8-bit preview crosses PCIe.

    python examples/timeline.py [out_dir]        # writes frame_000.png ... (and prints Mpx/s)
"""
import os
import struct
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluggo.media import process  # noqa: E402
from fluggo.media.basetypes import box2i  # noqa: E402

W, H, FRAMES = 1280, 720, 60


def write_png(path, rgba, w, h):
    """RGBA8 rows -> PNG (zlib only)."""
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    rows = b"".join(b"\0" + bytes(rgba[y * w * 4:(y + 1) * w * 4]) for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows, 6)) + chunk(b"IEND", b""))

out_dir = sys.argv[1] if len(sys.argv) > 1 else None

assert process.check_context_supported(), process.last_error()


def clip(start, end, gain):
    """A 'clip': a colour that drifts over time, graded (half-native: the gain filter works on f16 frames)."""
    return process.VideoGainOffsetFilter(process.SolidColorVideoSource(process.LerpFunc(start, end, FRAMES)), gain=gain, offset=0.0)


a = clip((0.8, 0.2, 0.1, 1.0), (0.9, 0.6, 0.1, 1.0), 1.0)
b = clip((0.1, 0.3, 0.8, 1.0), (0.1, 0.7, 0.6, 1.0), 1.1)

# cut + crossfade: a for 20 frames, a -> b over 20 frames (an AnimationFunc drives the mix), b for 20 frames
fade = process.AnimationFunc()
fade.add(process.POINT_LINEAR, 0.0, 0.0)
fade.add(process.POINT_HOLD, 20.0, 1.0)
sequence = process.VideoSequence()
sequence.append((a, 0, 20))
sequence.append((process.VideoMixFilter(src_a=process.VideoPassThroughFilter(a, offset=20), src_b=b, mix_b=fade), 0, 20))
sequence.append((process.VideoPassThroughFilter(b, offset=20), 0, 20))

# picture in picture: clip b at quarter size, placed at (W - 360, 40), half transparent through its gain node's alpha
pip_source = process.VideoGainOffsetFilter(process.SolidColorVideoSource((0.95, 0.95, 0.2, 0.6), box2i(0, 0, W - 1, H - 1)), gain=1.0)
pip = process.VideoScaler(pip_source, target_point=(W - 360, 40), source_point=(0, 0), scale_factors=(0.25, 0.25),
                          source_rect=box2i(0, 0, W - 1, H - 1))
title = process.SolidColorVideoSource((0.0, 0.0, 0.0, 0.7), box2i(0, H - 90, W - 1, H - 1))

timeline = process.VideoWorkspace()
timeline.add(source=sequence, x=0, length=FRAMES, z=0, offset=0)
timeline.add(source=pip, x=10, length=40, z=1, offset=0)
timeline.add(source=title, x=0, length=FRAMES, z=2, offset=0)

window = box2i(0, 0, W - 1, H - 1)
timeline.get_frame_rgba8(0, window)                       # first use: tables, code objects
t0 = time.perf_counter()
for i in range(FRAMES):
    rgba, cur = timeline.get_frame_rgba8(i, window)      # sRGB bytes made on the device, 4 B/px downloaded
dt = time.perf_counter() - t0
print("%d frames of %dx%d in %.1f ms: %.0f Mpx/s through the Python surface (preview bytes included)" % (FRAMES, W, H, dt * 1e3, FRAMES * W * H / dt / 1e6))
if out_dir:
    os.makedirs(out_dir, exist_ok=True)
    for i in range(0, FRAMES, 10):
        rgba, cur = timeline.get_frame_rgba8(i, window)
        if rgba is not None:
            write_png(os.path.join(out_dir, "frame_%03d.png" % i), rgba, cur.max.x - cur.min.x + 1, cur.max.y - cur.min.y + 1)
