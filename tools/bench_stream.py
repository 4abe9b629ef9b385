#!/usr/bin/env python3
"""BASELINE config 5: 3840x2160 10-node graph (4 sources, colour -> blur -> 4-step composite) over a 600-frame synthetic stream.

    python tools/bench_stream.py [--frames 600] [--ring 4] [--width 3840 --height 2160] [--gpus N]

--gpus N > 1 without a launcher: the script starts its N ranks itself (canvas_amd/launch.py), one per GPU; under
torch.distributed.run it takes the launcher's ranks.  The stream has --frames frames IN TOTAL; global frame g is rendered
by rank g % N (rank r: frames r, r + N, ...).  Every rank proves one whole output frame (its first: stream frame r)
against tests/golden/stream_frames_sha256.json.

Prints one JSON line: whole-job Mpixels/s (output pixels), per-rank frame counts, ms per frame and the
split of a frame's time over its two launches (HIP events on the launch stream).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--ring", type=int, default=4)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the frames alternate over (tails of one frame overlap the next)")
    a = ap.parse_args()
    from canvas_amd import launch
    launch.ensure_ranks(a.gpus)

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from canvas_amd import _lib, shard
    from canvas_amd.stream import BYTES_PER_PIXEL, NODE_BYTES_PER_PIXEL, GraphStream
    if os.environ.get("CANVAS_LIB"):                      # A/B runs of tools/: another build of the library (never set by the package)
        _lib.LIB_PATH = os.environ["CANVAS_LIB"]
    lib = _lib.load()
    if lib.cvs_init(local) != 0:
        raise SystemExit("no HIP device: " + _lib.last_error())
    assert a.ring % a.streams == 0, "slot i always runs on stream i % streams: the ring must be a multiple of the stream count"
    g = GraphStream(a.width, a.height, ring=a.ring, first_frame=rank, frame_step=world)      # slot s: stream frame rank + s * world
    if dist is not None:
        g.matrix[:] = shard.broadcast_parameters(lib, dist, rank, g.matrix, [_lib.LUT_REC709_TO_LINEAR_SCENE])
    stream = lib.cvs_stream_create()
    for i in range(a.warmup):
        g.render(i, stream)
    _lib.check(lib.cvs_stream_sync(stream))

    # split of one frame over its three launches
    ev = [lib.cvs_event_create() for _ in range(3)]
    s = g.slots[0]
    lib.cvs_event_record(ev[0], stream)
    lib.cvs_color_matrix_f16_to_dev(s["graded"].ref(), s["src"].ref(), g._m, g.pre_lut, g.post_lut, stream)
    lib.cvs_event_record(ev[1], stream)
    lib.cvs_blur_over_f16_dev(s["out"].ref(), s["graded"].ref(), g._t, len(g.taps), s["over_refs"], g.overlays, stream)
    lib.cvs_event_record(ev[2], stream)
    lib.cvs_event_sync(ev[2])
    split = [lib.cvs_event_elapsed_ms(ev[i], ev[i + 1]) for i in range(2)]

    if dist is not None:
        dist.barrier()
    _lib.check(lib.cvs_stream_sync(stream))
    streams = [stream] + [lib.cvs_stream_create() for _ in range(a.streams - 1)]
    t0 = time.perf_counter()
    frames = list(range(rank, a.frames, world))         # this rank's share of the stream's frames [0, frames)
    for i, _g in enumerate(frames):                     # slot i % ring on stream i % streams (ring % streams == 0: a slot never changes stream)
        g.render(i, streams[i % a.streams])
    mine = len(frames)
    t_enqueued = time.perf_counter() - t0          # host time to enqueue everything (launch-bound if close to the total)
    for st in streams:
        _lib.check(lib.cvs_stream_sync(st))
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    from canvas_amd import verify
    g.render(0, stream)                                  # slot 0 = stream frame `rank`, rendered once more for the proof
    digest = verify.canon_sha256(g.slots[0]["out"].download(stream).array)
    want = verify.stream_fixture("config5_3840x2160", rank) if (a.width, a.height) == (3840, 2160) else None
    stats = shard.gather_stats(dist, mine, shard.checksum52(digest), dt, extra=(-1 if want is None else int(digest == want),))
    counts = [int(x[0]) for x in stats]
    seconds = max(float(x[2]) for x in stats)
    if rank == 0:
        px = a.width * a.height
        print(json.dumps({
            "metric": "Mpixels/s", "value": round(sum(counts) * px / seconds / 1e6, 1), "n_gpus": world,
            "config": {"workload": "config5: %dx%d 10-node graph (4 sources, colour->blur->4-step composite), %d-frame stream" % (a.width, a.height, a.frames)},
            "frames_per_rank": counts, "ranks_verified": sum(1 for x in stats if int(x[3]) == 1), "streams": a.streams, "host_enqueue_ms_per_frame": round(t_enqueued / max(mine, 1) * 1e3, 4), "ms_per_frame": round(seconds / max(counts) * 1e3, 4),
            "launch_ms": {"colour": round(split[0], 4), "blur+over": round(split[1], 4)},
            "node_bytes_per_pixel": NODE_BYTES_PER_PIXEL, "moved_bytes_per_pixel": BYTES_PER_PIXEL,
            "moved_GBps_per_gpu": round(max(counts) * px * BYTES_PER_PIXEL / seconds / 1e9, 1),
        }))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
