#!/usr/bin/env python3
"""bench.py -- BASELINE config 2: 3840x2160 f16 RGBA, Rec.709 transfer LUT + RGB->Y'PbPr matrix on
each of 2 layers, 2-layer alpha-over, f16 out; frames resident in HBM; Mpixels/s.

  python bench.py --gpus N --steps K --warmup W

One process per GPU.  Under `torch.distributed.run` the ranks are the launcher's (RANK / LOCAL_RANK / WORLD_SIZE);
started plainly with --gpus N > 1, this script starts its N ranks itself (canvas_amd/launch.py) before anything touches
the GPU.  Frames shard round-robin: global frame g belongs to rank g % N; no collective on the frame path.  RCCL
carries the parameter block (matrix + 128 KiB LUT) once at start and one all-gather of per-rank
{frames, checksum, seconds, verified, launch ms} at the end.

A "step" = one batch of --batch frames per GPU through the fused chain kernel (ONE call of the C-ABI entry
cvs_chain_color_over_f16_dev; the library cuts it into launches of about eight frames).
The input ring (--ring frame sets, default 8 x 199 MB = 1.6 GB) is several times the 256 MiB Infinity Cache, so every
launch streams from HBM (a 64-frame step walks the ring eight times).

Every rank proves its pixels: the output of its first frame (global frame = rank) is downloaded whole and its SHA-256
compared with the committed fixture tests/golden/stream_frames_sha256.json (made by the oracle in the build container);
the JSON line carries `ranks_verified`.  After the timed config-2 region the other BASELINE configs (3, 4, 5) run briefly
on every rank, each over more than 1.5 GB of rotating frames, and appear as `extra` sub-records.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 measured copy)
BYTES_PER_PIXEL_PER_LAYER = 8  # one rgba_f16 read per layer pixel + one written per output pixel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU (one C-ABI call)")
    ap.add_argument("--ring", type=int, default=8, help="distinct frame sets resident per GPU")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--translucent-base", action="store_true",
                    help="NOT the BASELINE input: random alpha on layer 0 as well, so every divide of the over operator is live")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-extra", action="store_true", help="skip the config 3 / 4 / 5 sub-records")
    ap.add_argument("--extra-seconds", type=float, default=1.5, help="timed region of each extra config, per rank (long enough for a 5 s SMI sampler to see the GPU busy)")
    return ap.parse_args()


def build_cpu_baseline():
    """The checker rebuilt for this host's cores (gcc fork + exec): called first thing in main(), BEFORE the process
    touches the GPU (a process that has initialised the GPU must not fork-and-exec on this pool)."""
    import oracle
    try:
        return oracle.build(force=True, arch="-march=native -mtune=native", out="/tmp/canvas_oracle_native_%d.so" % os.getpid())
    except Exception:
        return None


def cpu_baseline(args, seconds, so):
    """The oracle (a scalar C port of the reference path) on the host cores, same workload, bounded."""
    import oracle
    from canvas_amd import REC709_RGB_TO_YPBPR, synth
    try:
        if so is None:
            raise RuntimeError("no native build")
        olib = oracle.lib(so)
        flags = "gcc -std=c99 -O3 -march=native -fno-math-errno -ffp-contract=off"
    except Exception:
        olib = oracle.lib()
        flags = "gcc -std=c99 -O3 -fno-math-errno -ffp-contract=off (generic x86-64)"
    w, h, nl = args.width, args.height, args.layers
    layers = [synth.layer_frame(w, h, k, 0) for k in range(nl)]
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    lut = oracle.transfer_table(0)
    from canvas_amd.abi import HostFrame, rgba_frame_f16
    arr = (C.POINTER(rgba_frame_f16) * nl)(*[C.pointer(l.c) for l in layers])

    def one(out):
        olib.orc_chain_color_over_f16(out.ref(), arr, nl, m.ctypes.data_as(C.POINTER(C.c_float)),
                                      lut.ctypes.data_as(C.POINTER(C.c_uint16)), None)

    out = HostFrame((0, 0, w - 1, h - 1), np.uint16)
    one(out)                                            # warm: builds the half tables
    t0, n = time.perf_counter(), 0
    while n < 3 or time.perf_counter() - t0 < seconds:
        one(out)
        n += 1
    dt1 = time.perf_counter() - t0
    single = n * w * h / dt1 / 1e6

    cores = os.cpu_count() or 1
    cores = min(cores, 64)
    from concurrent.futures import ThreadPoolExecutor
    outs = [HostFrame((0, 0, w - 1, h - 1), np.uint16) for _ in range(cores)]

    def worker(o):
        c, t = 0, time.perf_counter()
        while c < 2 or time.perf_counter() - t < seconds:
            one(o)                                      # ctypes drops the GIL for the call
            c += 1
        return c

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(worker, outs))
    dtn = time.perf_counter() - t0
    if so and os.path.exists(so):
        os.unlink(so)
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(single, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": "%d x (%dx%d, %d layers) in %.1f s, 1 thread, %s" % (n, w, h, nl, dt1, flags),
        "all_cores": {"value": round(total * w * h / dtn / 1e6, 2), "cores": cores,
                      "sample": "%d frames in %.1f s, %d threads on independent frames" % (total, dtn, cores),
                      "note": "scales poorly by construction: like the reference (main.c:43-71, color.c:122) the port allocates and "
                              "frees full-frame f32 temporaries on every pull, so the threads contend in the allocator and on page faults"},
        "cpu": cpu,
    }


class Timed:
    """HIP events around every step of a region, on the stream the work is launched on."""

    def __init__(self, lib, stream):
        self.lib, self.stream, self.ev = lib, stream, []

    def step(self, fn):
        a, b = self.lib.cvs_event_create(), self.lib.cvs_event_create()
        self.lib.cvs_event_record(a, self.stream)
        fn()
        self.lib.cvs_event_record(b, self.stream)
        self.ev.append((a, b))

    def ms(self):
        out = [self.lib.cvs_event_elapsed_ms(a, b) for a, b in self.ev]
        for a, b in self.ev:
            self.lib.cvs_event_destroy(a), self.lib.cvs_event_destroy(b)
        self.ev = []
        return out


def spread(ms):
    return {"mean": round(float(np.mean(ms)), 4), "median": round(float(statistics.median(ms)), 4),
            "min": round(float(min(ms)), 4), "max": round(float(max(ms)), 4), "n": len(ms)}


def main():
    args = parse()
    from canvas_amd import launch
    launch.ensure_ranks(args.gpus)          # --gpus N without a launcher: become N ranks (never returns in the parent)
    native_oracle = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        native_oracle = build_cpu_baseline()                           # (the checker, built before the first GPU call)

    # stdout carries exactly ONE line, the JSON result: libraries that write banners there (RCCL prints its version,
    # host name and library path on communicator creation) are sent to stderr until the result is ready
    result = launch.ResultOnly().__enter__()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if world > 1:
        launch.tag_own_stderr(rank)         # "[rank r] " in front of every stderr line (no-op under this script's own launcher, which tags)
    # host placement, BEFORE the first HIP call: each rank on the CPUs local to its GPU (canvas_amd/launch.py place_rank)
    placement = launch.place_rank(local_rank, world)
    if world > 1:
        sys.stderr.write("rank %d of %d: GPU %d, host placement: %s\n" % (rank, world, local_rank, json.dumps(placement)))

    dist = None
    # CANVAS_FORCE_DIST=1 takes the multi-process path at world size 1 too (rehearsal of the RCCL code on one GPU)
    if world > 1 or (os.environ.get("CANVAS_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth, verify
    from canvas_amd.device import DeviceFrame, chain_color_over
    from canvas_amd.shard import broadcast_parameters, checksum52, frames_of_rank, gather_stats

    lib = _lib.load()
    _lib.check(lib.cvs_init(local_rank), "cvs_init(%d)" % local_rank)
    lib.init_half()
    stream = lib.cvs_stream_create()

    # parameter block: rank 0 owns it, everyone else receives it over RCCL
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    m = broadcast_parameters(lib, dist, rank, m, [_lib.LUT_REC709_TO_LINEAR_SCENE])

    w, h, nl = args.width, args.height, args.layers
    full = (0, 0, w - 1, h - 1)
    ring = []
    # weak scaling: every rank owns --ring frame sets whatever the world size (frames_of_rank counts per rank);
    # global frame g lives on rank g % world
    my_frames = frames_of_rank(rank, world, args.ring)
    # one arena for the whole ring: consecutive 64 MiB-aligned slots
    frame_bytes = w * h * 8
    slot = (frame_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    arena = lib.cvs_malloc(slot * (nl + 1) * len(my_frames))
    if not arena:
        raise MemoryError("ring arena: " + _lib.last_error())
    at = [arena]

    def place():
        d = DeviceFrame(full, np.uint16, ptr=at[0])
        at[0] += slot
        return d

    for g in my_frames:
        layers = []
        for k in range(nl):
            d = place()
            d.upload(synth.layer_pixels(w, h, k, g, opaque_base=not args.translucent_base))
            layers.append(d)
        ring.append((place(), layers))

    def step(i):
        jobs = [ring[(i * args.batch + b) % len(ring)] for b in range(args.batch)]
        chain_color_over(jobs, m, _lib.LUT_REC709_TO_LINEAR_SCENE, _lib.LUT_NONE, stream)

    def barrier():
        _lib.check(lib.cvs_stream_sync(stream), "sync")
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    barrier()
    assert lib.cvs_chain_last_was_fused() == 1, "the fused kernel did not run"
    launches_per_step = max(1, lib.cvs_chain_last_launch_count())      # the library cuts a step's batch into launches of ~8 frames

    timed = Timed(lib, stream)
    t0 = time.perf_counter()
    for i in range(args.steps):
        timed.step(lambda: step(args.warmup + i))
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = timed.ms()

    # ---- every rank proves its pixels: whole first frame, SHA-256 against the committed fixture
    standard = (w, h, nl) == (3840, 2160, 2) and not args.translucent_base
    digest = verify.canon_sha256(ring[0][0].download(stream).array)
    want = verify.stream_fixture("config2_3840x2160", my_frames[0]) if standard else None
    verified = -1 if want is None else int(digest == want)            # -1: no fixture for this shape / frame

    px_per_step = args.batch * w * h
    algo_bytes = px_per_step * BYTES_PER_PIXEL_PER_LAYER * (nl + 1)
    my_frac = algo_bytes / (float(np.mean(step_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS
    stats = gather_stats(dist, args.steps * args.batch, checksum52(digest), elapsed, extra=(verified, float(np.mean(step_ms)), my_frac))
    elapsed = max(s[2] for s in stats)                                  # the slowest rank's clock

    # same-run yardstick for this box: plain device-to-device copies of the same frames
    copy_gbs = None
    if rank == 0:
        try:
            e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
            pairs = [(ring[i % len(ring)][0], ring[(i + 1) % len(ring)][1][0]) for i in range(32)]
            for rep in range(2):
                lib.cvs_event_record(e0, stream)
                for dst, src in pairs:
                    lib.cvs_memcpy_d2d(dst.ptr, src.ptr, frame_bytes, stream)
                lib.cvs_event_record(e1, stream)
                _lib.check(lib.cvs_stream_sync(stream), "sync")
            copy_gbs = round(2 * frame_bytes * len(pairs) / (lib.cvs_event_elapsed_ms(e0, e1) * 1e-3) / 1e9, 1)
            lib.cvs_event_destroy(e0), lib.cvs_event_destroy(e1)
        except Exception:                                       # a yardstick, not a result
            copy_gbs = None

    extra = []
    if not args.no_extra and standard:
        from bench_extra import run_extras
        extra = run_extras(lib, dist, rank, world, stream, ring, my_frames, m, args.extra_seconds)

    if rank == 0:
        total_px = px_per_step * args.steps * world
        avg_ms = float(np.mean(step_ms))
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = round(tj["k_chain_bytes_per_output_pixel"] * px_per_step)
                traffic_src = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on a builder-side run, scaled to this run's pixels per step); NOT measured in this run" % tj.get("source", "profiles/hbm_traffic.json")
            except Exception:
                traffic = None
        res = {
            "metric": "Mpixels/s through 4K f16 RGBA colour-matrix+alpha-over chain; % HBM roofline",
            "value": round(total_px / elapsed / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",      # the arithmetic type; pixels are stored as f16 (see config.storage)
            "data": "synthetic (Philox, seed 0xC0FFEE+1000*layer+frame), resident in HBM" + (
                "; NON-BASELINE variant: random alpha on layer 0 too" if args.translucent_base else ""),
            "config": {"workload": "%dx%d f16 RGBA, Rec.709->linear LUT + RGB->Y'PbPr 3x3 on %d layers + %d-layer alpha-over, f16 out" % (w, h, nl, nl),
                       "storage": "rgba_f16 (8 B per pixel) in and out, f32 arithmetic in registers",
                       "frames_per_step_per_gpu": args.batch, "ring_frames_per_gpu": len(ring),
                       "sharding": "frame g -> gpu g %% %d, no data-path collective" % world,
                       "host_placement_rank0": placement,
                       "collectives": "none" if dist is None else "%s: one broadcast of the parameter block, one all-gather of per-rank results per record" % dist.get_backend()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_chain<%d layers, grade, pre-LUT>" % nl,
                         "launches_per_step": launches_per_step,
                         "avg_launch_ms": round(avg_ms / launches_per_step, 4),      # compare with rocprofv3's average k_chain duration
                         "step_ms": spread(step_ms),
                         "same_run_dtod_copy_GBps": copy_gbs,
                         "algorithmic_bytes_per_launch": algo_bytes // launches_per_step,
                         "algorithmic_bytes_per_step": algo_bytes,
                         "note": "one step = one C-ABI call over %d frames, which the library issues as %d back-to-back launches; "
                                 "achieved = %d B/px x %d px per step / HIP-event step time on the launch stream (rank 0), i.e. launch "
                                 "durations plus the gaps between them" % (args.batch, launches_per_step, BYTES_PER_PIXEL_PER_LAYER * (nl + 1), px_per_step)},
            "frames_per_rank": [s[0] for s in stats],
            "ranks_verified": sum(1 for s in stats if int(s[3]) == 1),
            "per_rank": [{"rank": r, "first_frame": r, "sha256_52bit": "%013x" % s[1],
                          "verified": {1: True, 0: False}.get(int(s[3]), "no fixture"),
                          "seconds": round(s[2], 4), "step_ms_mean": round(s[4], 4), "frac": round(s[5], 4)} for r, s in enumerate(stats)],
            "verified_against_fixture": all(int(s[3]) == 1 for s in stats) if standard else "no fixture for this shape",
            "device": lib.cvs_device_name().decode(),
        }
        if extra:
            res["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds, native_oracle)
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}
        result.emit(json.dumps(res))

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    bad = [r for r, s in enumerate(stats) if int(s[3]) == 0]
    if bad:
        sys.stderr.write("ranks %s rendered pixels that do not match the fixture\n" % bad)
        sys.exit(3)


if __name__ == "__main__":
    main()
