#!/usr/bin/env python3
"""bench.py -- BASELINE config 2: 3840x2160 f16 RGBA, Rec.709 transfer LUT + RGB->Y'PbPr matrix on
each of 2 layers, 2-layer alpha-over, f16 out; frames resident in HBM; Mpixels/s.

  python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE).  Frames shard
round-robin: global frame g belongs to rank g % N; no collective on the frame path.  RCCL carries
only the parameter block (matrix + 128 KiB LUT) once at start and the timings at the end.

A "step" = one batch of --batch frames per GPU through the fused chain kernel (one launch).
The input ring (--ring frame sets, default 8 x 199 MB = 1.6 GB) is several times the 256 MiB Infinity
Cache, so every launch streams from HBM (a 64-frame step walks the ring eight times).
"""
import argparse
import ctypes as C
import json
import os
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
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU (one kernel launch; 64 = the most job records one launch carries)")
    ap.add_argument("--ring", type=int, default=8, help="distinct frame sets resident per GPU")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--translucent-base", action="store_true",
                    help="NOT the BASELINE input: random alpha on layer 0 as well, so every divide of the over operator is live")
    ap.add_argument("--no-arena", action="store_true", help="one hipMalloc per frame instead of one arena for the ring")
    ap.add_argument("--slot-pad", type=int, default=0, help="extra bytes between consecutive frames of the arena (placement experiment; multiple of 256)")
    ap.add_argument("--pre-alloc-mb", type=int, default=0, help="allocate (and keep) this much device memory before the ring (placement experiment)")
    ap.add_argument("--arena-align-mb", type=int, default=0, help="round the ring's base address up to this many MiB (placement experiment)")
    ap.add_argument("--report-base", action="store_true", help="print the ring's base address on stderr (placement experiment)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def cpu_baseline(args, seconds):
    """The oracle (a scalar C port of the reference path) on the host cores, same workload, bounded."""
    import oracle
    from canvas_amd import REC709_RGB_TO_YPBPR, synth
    so = None
    try:
        so = oracle.build(force=True, arch="-march=native -mtune=native", out="/tmp/canvas_oracle_native_%d.so" % os.getpid())
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
                      "sample": "%d frames in %.1f s, %d threads on independent frames" % (total, dtn, cores)},
        "cpu": cpu,
    }


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON result: libraries that write banners there (RCCL prints its version,
    # host name and library path on communicator creation) are sent to stderr until the result is ready
    sys.stdout.flush()
    try:
        real_stdout = os.dup(1)
        os.dup2(2, 1)
    except OSError:                     # no usable stderr: leave stdout alone
        real_stdout = None
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    dist = None
    # CANVAS_FORCE_DIST=1 takes the multi-process path at world size 1 too (rehearsal of the RCCL code on one GPU)
    if world > 1 or (os.environ.get("CANVAS_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth
    from canvas_amd.device import DeviceFrame, chain_color_over
    from canvas_amd.shard import broadcast_parameters, frames_of_rank

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
    # one arena for the whole ring: a single large allocation maps with far fewer page-table entries than
    # dozens of 66 MB ones, and the TLB reach of the chip is what a multi-GB streaming working set leans on
    frame_bytes = w * h * 8
    slot = (frame_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20) + args.slot_pad
    dummy = lib.cvs_malloc(args.pre_alloc_mb << 20) if args.pre_alloc_mb else None      # noqa: F841 -- kept alive on purpose
    align = args.arena_align_mb << 20
    arena = None if args.no_arena else lib.cvs_malloc(slot * (nl + 1) * len(my_frames) + align)
    at = [arena if not align or arena is None else (arena + align - 1) // align * align]
    if rank == 0 and (args.pre_alloc_mb or align or args.report_base):
        print("ring base %#x (arena %#x)" % (at[0], arena), file=sys.stderr)

    def place():
        if arena is None:
            return DeviceFrame(full, np.uint16)
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

    ev = [(lib.cvs_event_create(), lib.cvs_event_create()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        lib.cvs_event_record(ev[i][0], stream)
        step(args.warmup + i)
        lib.cvs_event_record(ev[i][1], stream)
    barrier()
    elapsed = time.perf_counter() - t0

    launch_ms = [lib.cvs_event_elapsed_ms(a, b) for a, b in ev]
    for a, b in ev:
        lib.cvs_event_destroy(a), lib.cvs_event_destroy(b)

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # guard: the head of one output frame against the oracle
    verified = None
    if rank == 0:
        try:
            import oracle
            rows = 4
            from canvas_amd.abi import HostFrame
            heads = [HostFrame((0, 0, w - 1, rows - 1), np.uint16, synth.layer_pixels(w, h, k, my_frames[0], opaque_base=not args.translucent_base)[:rows]) for k in range(nl)]
            want = oracle.chain_color_over(heads, m, oracle.transfer_table(0), None)
            got = ring[0][0].download(stream).array[:rows]
            verified = bool(np.array_equal(got, want.array))
        except Exception as e:                                  # the oracle is only a checker here
            verified = "unchecked: %s" % e

    # same-run yardstick for this box: plain device-to-device copies of the same frames (SURVEY 8d asks for one beside
    # the roofline figure; boxes of the pool differ by several per cent)
    copy_gbs = None
    if rank == 0:
        try:
            e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
            nbytes = w * h * 8
            pairs = [(ring[i % len(ring)][0], ring[(i + 1) % len(ring)][1][0]) for i in range(32)]
            for rep in range(2):
                lib.cvs_event_record(e0, stream)
                for dst, src in pairs:
                    lib.cvs_memcpy_d2d(dst.ptr, src.ptr, nbytes, stream)
                lib.cvs_event_record(e1, stream)
                _lib.check(lib.cvs_stream_sync(stream), "sync")
            copy_gbs = round(2 * nbytes * len(pairs) / (lib.cvs_event_elapsed_ms(e0, e1) * 1e-3) / 1e9, 1)
            lib.cvs_event_destroy(e0), lib.cvs_event_destroy(e1)
        except Exception:                                       # a yardstick, not a result
            copy_gbs = None

    if rank == 0:
        px_per_step = args.batch * w * h
        total_px = px_per_step * args.steps * world
        avg_ms = float(np.mean(launch_ms))
        algo_bytes = px_per_step * BYTES_PER_PIXEL_PER_LAYER * (nl + 1)
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                # PMC-measured HBM bytes per output pixel of k_chain (FETCH_SIZE x2-corrected + WRITE_SIZE, separate
                # rocprofv3 --pmc passes, tools/profile_bench.sh), scaled to this run's pixels per launch
                traffic = round(json.load(open(tpath))["k_chain_bytes_per_output_pixel"] * px_per_step)
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
                       "sharding": "frame g -> gpu g %% %d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "k_chain<%d,pre-LUT>" % nl, "avg_launch_ms": round(avg_ms, 4),
                         "same_run_dtod_copy_GBps": copy_gbs,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "achieved = %d B/px x %d px per launch / HIP-event launch time" % (BYTES_PER_PIXEL_PER_LAYER * (nl + 1), px_per_step)},
            "verified_against_oracle": verified,
            "device": lib.cvs_device_name().decode(),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}
        sys.stdout.flush()
        if real_stdout is not None:
            os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        if real_stdout is not None:
            os.dup2(2, 1)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
