#!/usr/bin/env python3
"""Leak check of the Python surface: many pulls through a graph of every node type; host RSS and free device memory
before and after must match (pool and caches warm after the first thousand)."""
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluggo.media import process  # noqa: E402
from fluggo.media.basetypes import box2i  # noqa: E402
from canvas_amd import _lib  # noqa: E402
import ctypes as C  # noqa: E402

lib = _lib.load()
W, H = 320, 180
window = box2i(0, 0, W - 1, H - 1)
a = process.VideoGainOffsetFilter(process.SolidColorVideoSource(process.LerpFunc((0.8, 0.2, 0.1, 1.0), (0.9, 0.6, 0.1, 1.0), 100.0)), gain=1.1)
b = process.SolidColorVideoSource((0.1, 0.3, 0.8, 0.6), box2i(20, 10, 300, 160))
fade = process.AnimationFunc()
fade.add(process.POINT_LINEAR, 0.0, 0.0)
fade.add(process.POINT_HOLD, 50.0, 1.0)
seq = process.VideoSequence()
seq.append((a, 0, 30))
seq.append((process.VideoMixFilter(src_a=process.VideoPassThroughFilter(a, offset=30), src_b=b, mix_b=fade), 0, 50))
seq.append((process.Pulldown23RemovalFilter(b, 1), 0, 40))
pip = process.VideoScaler(a, target_point=(200, 20), source_point=(0, 0), scale_factors=(0.3, 0.3), source_rect=window)
ws = process.VideoWorkspace()
ws.add(source=seq, x=0, length=120, z=0, offset=0)
ws.add(source=pip, x=10, length=100, z=1, offset=0)
ws.add(source=process.SolidColorVideoSource((0.0, 0.0, 0.0, 0.7), box2i(0, H - 30, W - 1, H - 1)), x=0, length=120, z=2, offset=0)


def free_device():
    free, total = C.c_size_t(), C.c_size_t()
    lib.cvs_mem_info(C.byref(free), C.byref(total))
    return free.value


def pulls(n):
    for i in range(n):
        f = i % 120
        ws.get_frame_f16(f, window)
        ws.get_frame_f32(f, window)
        ws.get_frame_rgba8(f, window)
        ws.get_frame_argb32(f, window).__len__()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pulls(1000)
rss0, dev0, t0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss, free_device(), time.perf_counter()
pulls(n)
rss1, dev1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss, free_device()
print("%d x 4 pulls in %.1f s; peak RSS %d -> %d KiB; free device memory %d -> %d bytes" % (n, time.perf_counter() - t0, rss0, rss1, dev0, dev1))
assert rss1 - rss0 < 8 * 1024, "host memory grew"
assert abs(dev0 - dev1) < (64 << 20), "device memory moved"
print("module soak ok")
