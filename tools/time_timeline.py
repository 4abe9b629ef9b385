#!/usr/bin/env python3
"""Where a preview pull through the Python surface spends its time: the pieces of examples/timeline.py pulled one by
one (1280x720, get_frame_rgba8 = render on the device + bytes on the device + 3.7 MB download)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluggo.media import process  # noqa: E402
from fluggo.media.basetypes import box2i  # noqa: E402

W, H, N = 1280, 720, 60
window = box2i(0, 0, W - 1, H - 1)
solid = process.SolidColorVideoSource((0.8, 0.2, 0.1, 1.0))
gain = process.VideoGainOffsetFilter(solid, gain=1.1, offset=0.0)
other = process.VideoGainOffsetFilter(process.SolidColorVideoSource((0.1, 0.3, 0.8, 1.0)), gain=1.0, offset=0.0)
mix = process.VideoMixFilter(src_a=gain, src_b=other, mix_b=process.LinearFrameFunc(1.0 / N, 0.0))
seq = process.VideoSequence()
seq.append((gain, 0, 20)); seq.append((mix, 0, 20)); seq.append((other, 0, 20))
pip = process.VideoScaler(process.SolidColorVideoSource((0.95, 0.95, 0.2, 0.6), window), target_point=(W - 360, 40), source_point=(0, 0),
                          scale_factors=(0.25, 0.25), source_rect=window)
ws2 = process.VideoWorkspace()
ws2.add(source=seq, x=0, length=N, z=0, offset=0)
ws2.add(source=process.SolidColorVideoSource((0.0, 0.0, 0.0, 0.7), box2i(0, H - 90, W - 1, H - 1)), x=0, length=N, z=1, offset=0)
ws3 = process.VideoWorkspace()
ws3.add(source=seq, x=0, length=N, z=0, offset=0)
ws3.add(source=pip, x=0, length=N, z=1, offset=0)
ws3.add(source=process.SolidColorVideoSource((0.0, 0.0, 0.0, 0.7), box2i(0, H - 90, W - 1, H - 1)), x=0, length=N, z=2, offset=0)

for name, node in [("solid colour", solid), ("gain(solid)", gain), ("crossfade", mix), ("sequence", seq), ("scaler (pip)", pip),
                   ("workspace: sequence + title", ws2), ("workspace: sequence + pip + title", ws3)]:
    for pull in ("get_frame_rgba8", "get_frame_f16"):
        fn = getattr(node, pull)
        fn(0, window)
        t0 = time.perf_counter()
        for i in range(N):
            fn(i, window)
        dt = (time.perf_counter() - t0) / N
        print("%-36s %-16s %.3f ms per frame" % (name, pull, dt * 1e3))
