#!/usr/bin/env python3
"""SURVEY 8f N4 as a number: VideoPullQueue(workers=N) throughput through the Python surface -- N frames in flight on
N HIP streams (one per worker thread) against ms per delivered frame.  The graph is the timeline of examples/timeline.py
(sequence with a crossfade + picture-in-picture scaler + title bar in a workspace) pulled as f16 host frames, the
reference's queue contract (VideoPullQueue.c:72-174: the callback gets an RgbaFrameF16), at preview (1280x720) and
full HD size; reference pool size is 2.

    python3 tools/time_pull_queue.py [--frames 240]
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluggo.media import process  # noqa: E402
from fluggo.media.basetypes import box2i  # noqa: E402


def timeline(w, h, n):
    window = box2i(0, 0, w - 1, h - 1)
    gain = process.VideoGainOffsetFilter(process.SolidColorVideoSource((0.8, 0.2, 0.1, 1.0)), gain=1.1, offset=0.0)
    other = process.VideoGainOffsetFilter(process.SolidColorVideoSource((0.1, 0.3, 0.8, 1.0)), gain=1.0, offset=0.0)
    mix = process.VideoMixFilter(src_a=gain, src_b=other, mix_b=process.LinearFrameFunc(1.0 / n, 0.0))
    seq = process.VideoSequence()
    seq.append((gain, 0, n // 3)); seq.append((mix, 0, n // 3)); seq.append((other, 0, n - 2 * (n // 3)))
    pip = process.VideoScaler(process.SolidColorVideoSource((0.95, 0.95, 0.2, 0.6), window), target_point=(w - w // 4 - 40, 40),
                              source_point=(0, 0), scale_factors=(0.25, 0.25), source_rect=window)
    ws = process.VideoWorkspace()
    ws.add(source=seq, x=0, length=n, z=0, offset=0)
    ws.add(source=pip, x=0, length=n, z=1, offset=0)
    ws.add(source=process.SolidColorVideoSource((0.0, 0.0, 0.0, 0.7), box2i(0, h - h // 8, w - 1, h - 1)), x=0, length=n, z=2, offset=0)
    return ws, window


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    a = ap.parse_args()
    for (w, h) in ((1280, 720), (1920, 1080)):
        ws, window = timeline(w, h, a.frames)
        for i in range(0, a.frames, 7):                       # warm: tables, scratch blocks, every branch of the sequence
            ws.get_frame_f16(i, window)
        t0 = time.perf_counter()
        for i in range(a.frames):
            ws.get_frame_f16(i, window)
        direct = (time.perf_counter() - t0) / a.frames
        print("%dx%d  direct get_frame_f16 on the calling thread        %.3f ms per frame" % (w, h, direct * 1e3))
        for workers in (1, 2, 4, 8):
            q = process.VideoPullQueue(workers=workers)
            done, count, lock = threading.Event(), [0], threading.Lock()

            def cb(i, frame, user):
                with lock:
                    count[0] += 1
                    if count[0] == a.frames:
                        done.set()
            warm = threading.Event()
            q.enqueue(ws, 0, window, lambda i, f, u: warm.set(), None)
            warm.wait(60)
            # at most 2 x workers requests outstanding, like a loader that keeps a short look-ahead: every request owns a
            # freshly allocated host frame, and hundreds of them at once would measure the host's page faults instead
            slots = threading.Semaphore(2 * workers)
            t0 = time.perf_counter()
            for i in range(a.frames):
                slots.acquire()
                q.enqueue(ws, i, window, lambda i, f, u: (cb(i, f, u), slots.release()), None)
            ok = done.wait(120)
            dt = (time.perf_counter() - t0) / a.frames
            print("%dx%d  VideoPullQueue(workers=%d)%s                          %.3f ms per frame  (%.2fx direct)" % (
                w, h, workers, "" if ok else " TIMED OUT", dt * 1e3, direct / dt))
            del q


if __name__ == "__main__":
    main()
