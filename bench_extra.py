"""The other BASELINE configs as sub-records of bench.py's JSON line (`extra`), measured after the timed config-2 region
on every rank of the run (frames shard the same way: global frame g -> rank g % N, nothing exchanged on the data path).

Every workload rotates over more than 1.5 GB of resident frames, so nothing it reads is served from the 256 MiB Infinity
Cache, and proves its pixels like the main run does: one whole output frame per rank (stream frame = rank) against the
committed SHA-256 fixture.  Only slot 0 of a workload holds the generator's frames; the other slots are filled on the device
with copies of config 2's ring (same value distribution, opaque bottom layer), which keeps the set-up to seconds.

  config 3   3840x2160 -> 9-tap separable Gaussian (sigma 1.5) -> Lanczos3 to 1920x1080      cvs_blur_lanczos_f16_dev
  config 4   7680x4320 3-layer alpha-over stack                                             cvs_chain_color_over_f16_dev (m = NULL)
  config 5   3840x2160 10-node graph (4 sources, colour -> blur -> 4-step composite)        canvas_amd.stream.GraphStream
  lanczos3_x0.40 / x0.75 / x1.50   3840x2160 Lanczos3 at factors other than 1/2: the GENERAL FIR path (per-line tap
             tables, sweep_hv_ops.hip), not a BASELINE config               cvs_resample_lanczos_f16_dev
  scaler_x2.00   1920x1080 -> 3840x2160 through the reference's own scaler        cvs_scale_bilinear_f16_dev
  config3_contracted, config5_contracted   the same two workloads in the library's OTHER arithmetic flavour
             (cvs_set_arithmetic(CVS_ARITH_CONTRACTED): a * b + c inside an expression fused, as the reference's preferred clang
             build computes it), each proven against the fixture made by the checker's build of that flavour; the headline and every
             other record run in the default flavour (the reference's gcc build)
"""
import contextlib
import ctypes as C
import time

import numpy as np

HBM_PEAK_BPS = 8.0e12


def _timed_passes(lib, _lib, streams, one_pass, seconds):
    """one_pass() enqueues one pass over the workload's slots; returns (passes, wall seconds) for about `seconds`."""
    def sync():
        for s in streams:
            _lib.check(lib.cvs_stream_sync(s), "sync")
    one_pass()
    sync()
    t0 = time.perf_counter()
    one_pass()
    sync()
    t1 = time.perf_counter() - t0
    n = max(3, int(seconds / max(t1, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(n):
        one_pass()
    sync()
    return n, time.perf_counter() - t0


def _record(dist, gather_stats, checksum52, name, workload, frames, px_per_frame, seconds, digest, want, bytes_per_px, note, rank, extra_fields=None,
            moved_bytes_per_px=None, kernels=None):
    """bytes_per_px: the per-node denominator; moved_bytes_per_px: what the launches actually read and write
    per pixel by construction (inputs once + outputs once of every LAUNCH; default: the same) -- both fractions are reported."""
    verified = -1 if want is None else int(digest == want)
    stats = gather_stats(dist, frames, checksum52(digest), seconds, extra=(verified,))
    if rank != 0:
        return None
    worst = max(s[2] for s in stats)
    total_frames = sum(s[0] for s in stats)
    per_gpu_frame_ms = worst / max(max(s[0] for s in stats), 1) * 1e3
    rec = {"config": name, "workload": workload, "n_gpus": len(stats),
           "value": round(total_frames * px_per_frame / worst / 1e6, 1), "unit": "Mpixels/s",
           "ms_per_frame_per_gpu": round(per_gpu_frame_ms, 4), "frames_per_rank": [s[0] for s in stats],
           "bytes_per_px": bytes_per_px,
           "frac_of_8TBps_per_gpu": round(px_per_frame * bytes_per_px / (per_gpu_frame_ms * 1e-3) / HBM_PEAK_BPS, 4),
           "ranks_verified": sum(1 for s in stats if int(s[3]) == 1), "timing": "wall clock around the enqueue + stream syncs", "note": note}
    moved = bytes_per_px if moved_bytes_per_px is None else moved_bytes_per_px
    gbs = lambda b: px_per_frame * b / (per_gpu_frame_ms * 1e-3) / 1e9          # noqa: E731
    rec["roofline"] = {"bound": "hbm", "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                       "achieved": round(gbs(bytes_per_px), 1), "frac": round(gbs(bytes_per_px) / (HBM_PEAK_BPS / 1e9), 4),
                       "bytes_per_px": bytes_per_px,
                       "achieved_moved": round(gbs(moved), 1), "frac_moved": round(gbs(moved) / (HBM_PEAK_BPS / 1e9), 4),
                       "moved_bytes_per_px": moved, "kernels": kernels,
                       "note": "frac: on SURVEY 8(d)'s per-node denominator; frac_moved: on the bytes the launches of this implementation "
                               "read and write by construction (each launch's inputs once + outputs once); wall clock per frame on this rank"}
    if extra_fields:
        rec.update(extra_fields)
    return rec


@contextlib.contextmanager
def arithmetic(lib, _lib, flavour):
    """The library in the named arithmetic flavour for the duration of the block (process-wide: nothing else may be in flight)."""
    before = lib.cvs_set_arithmetic(_lib.ARITH_CONTRACTED if flavour == "contracted" else _lib.ARITH_SEPARATE)
    try:
        yield
    finally:
        lib.cvs_set_arithmetic(before if before >= 0 else _lib.ARITH_SEPARATE)


class GraphStreamView:
    """A fixed group of a GraphStream's slots rendered with one batch call (its pointer tables are built once)."""

    def __init__(self, graph, slots):
        import copy
        self.g = copy.copy(graph)           # shares the frames; own cache of pointer tables
        self.g._batch_key = None
        self.slots = slots

    def render(self, stream):
        return self.g.render_batch(self.slots, stream)


def run_extras(lib, dist, rank, world, stream, ring, my_frames, matrix, seconds):
    from canvas_amd import _lib, synth, verify
    from canvas_amd.device import DeviceFrame, chain_color_over
    from canvas_amd.shard import checksum52, gather_stats
    from canvas_amd.stream import BYTES_PER_PIXEL, NODE_BYTES_PER_PIXEL, GraphStream

    out = []
    g0 = my_frames[0]
    f32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))          # noqa: E731

    # ---------------------------------------------------------------- config 3
    w, h = 3840, 2160
    taps = synth.gaussian_taps(9, 1.5)
    sources = [ring[0][1][1]] + [l for i, (_o, ls) in enumerate(ring) for k, l in enumerate(ls) if not (i == 0 and k == 1)]
    smalls = [DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16) for _ in sources]      # 16 x (66 + 17) MB = 1.3 GB + the f32 scratch frame

    # the frames of a pass are independent: they go to the library four at a time (cvs_blur_lanczos_f16_batch_dev: one
    # launch per batch, row segments sized for all four frames), the batches alternating over two HIP streams -- frames in
    # flight, as a pull queue with two workers keeps them (profiles/r03/config3_batches.txt: 1 frame per call on 1 stream
    # 0.084 ms, 4 per call 0.071, 4 per call on 2 streams 0.063)
    fp16 = C.POINTER(_lib.rgba_frame_f16_t)
    per = 4
    streams3 = [stream, lib.cvs_stream_create()]
    b3 = [((fp16 * per)(*[C.pointer(d.c) for d in smalls[a:a + per]]), (fp16 * per)(*[C.pointer(d.c) for d in sources[a:a + per]]))
          for a in range(0, len(sources) - per + 1, per)]
    assert len(b3) * per == len(sources)

    def pass3():
        for k, (dst, src) in enumerate(b3):
            _lib.check(lib.cvs_blur_lanczos_f16_batch_dev(dst, src, per, f32p(taps), 9, C.c_float(0.5), C.c_float(0.5), 3, streams3[k % 2]), "config 3")

    for flavour, sfx in (("separate", ""), ("contracted", "_contracted")):
        with arithmetic(lib, _lib, flavour):
            n, dt = _timed_passes(lib, _lib, streams3, pass3, seconds)
        digest = verify.canon_sha256(smalls[0].download(stream).array)
        rec = _record(dist, gather_stats, checksum52, "config3" + sfx, "3840x2160 f16 -> 9-tap Gaussian -> Lanczos3 -> 1920x1080 f16",
                      n * len(sources), w * h, dt, digest, verify.stream_fixture("config3_3840x2160_to_1920x1080" + ("@contracted" if sfx else ""), g0), 26,
                      "Mpixels/s and bytes are per INPUT pixel; 26 B/px is BASELINE's per-node denominator (blur 8 r + 8 w, scale 8 r + 2 w); "
                      "a fused form needs 10 B/px (8 r + 2 w); the 16 independent frames of a pass go to the library four at a time "
                      "(cvs_blur_lanczos_f16_batch_dev: one launch per batch), the batches alternating over two HIP streams", rank,
                      {"fused_lower_bound_bytes_per_px": 10, "frames_per_launch": 4, "streams": 2, "arithmetic": flavour},
                      moved_bytes_per_px=10, kernels=["k_blur_halve_pair<9, 11 taps, 128 lanes> (two source columns per lane, four frames per launch)"
                                                      + (", contracted build: one v_pk_fma_f32 per tap and channel pair where the default flavour issues a multiply and an add" if sfx else "")])
        if rec:
            out.append(rec)
    lib.cvs_stream_destroy(streams3[1])
    for d in smalls:
        d.free()

    # ---------------------------------------------------------------- the general FIR path (no BASELINE config: VERDICT r01 item 6)
    for factor, tag in ((0.4, "0.40"), (0.75, "0.75"), (1.5, "1.50")):
        tw, th = int(w * factor), int(h * factor)
        nout = 16 if factor < 1.0 else 6                                        # 16 x 66 MB of sources rotate either way
        outs = [DeviceFrame((0, 0, tw - 1, th - 1), np.uint16) for _ in range(nout)]

        def pass_l():
            for i, src in enumerate(sources):
                _lib.check(lib.cvs_resample_lanczos_f16_dev(outs[i % nout].ref(), src.ref(), C.c_float(factor), C.c_float(factor), 3, stream), "lanczos")

        n, dt = _timed_passes(lib, _lib, [stream], pass_l, seconds / 3)
        if nout < len(sources):                     # slot 0 was last written from another source: the generator's frame once more
            _lib.check(lib.cvs_resample_lanczos_f16_dev(outs[0].ref(), sources[0].ref(), C.c_float(factor), C.c_float(factor), 3, stream), "lanczos")
        digest = verify.canon_sha256(outs[0].download(stream).array)
        in_bytes, out_bytes = 8, 8 * (tw * th) / (w * h)
        rec = _record(dist, gather_stats, checksum52, "lanczos3_x" + tag, "3840x2160 f16 -> Lanczos3 -> %dx%d f16 (per-line tap tables)" % (tw, th),
                      n * len(sources), w * h, dt, digest, verify.stream_fixture("lanczos3_3840x2160_x" + tag, g0), round(in_bytes + out_bytes, 2),
                      "Mpixels/s and bytes are per INPUT pixel: source read once + target written once", rank, kernels=["k_fir_hv (per-line gather): one launch"])
        if rec:
            out.append(rec)
        for d in outs:
            d.free()

    # ---------------------------------------------------------------- the reference's own scaler, 1080p -> 4K (A9; no BASELINE config)
    from canvas_amd.abi import v2f
    small_w, small_h = w // 2, h // 2
    # 96 sources (1.6 GB: the SOURCES alone must not fit the 256 MiB Infinity Cache -- the kernels' stores are non-temporal and
    # leave it to what is read) into 16 targets (1.06 GB); every source that lands in target 0 is the generator's frame
    srcs = []
    nsrc = 6 * len(sources)
    for i in range(nsrc):
        d = DeviceFrame((0, 0, small_w - 1, small_h - 1), np.uint16)
        if i == 0:
            d.upload(synth.layer_pixels(small_w, small_h, 1, g0))
        elif i % len(sources) == 0:
            _lib.check(lib.cvs_memcpy_d2d(d.ptr, srcs[0].ptr, d.nbytes, stream), "d2d")
        else:                                                                      # a quarter of a 4K frame's pixels: same distribution
            donor = sources[i % len(sources)]
            off = (i // len(sources)) % 4 * d.nbytes
            _lib.check(lib.cvs_memcpy_d2d(d.ptr, donor.ptr + off, d.nbytes, stream), "d2d")
        srcs.append(d)
    bigs = [DeviceFrame((0, 0, w - 1, h - 1), np.uint16) for _ in sources]           # 96 x 17 MB read, 16 x 66 MB written, rotating

    streams_s = [stream, lib.cvs_stream_create()]               # two frames in flight (the reference's pull queue has two workers)

    def pass_s():
        for i, src in enumerate(srcs):
            _lib.check(lib.cvs_scale_bilinear_f16_dev(bigs[i % len(bigs)].ref(), v2f(0, 0), src.ref(), v2f(0, 0), v2f(2.0, 2.0), streams_s[i % 2]), "scaler")

    n, dt = _timed_passes(lib, _lib, streams_s, pass_s, seconds / 2)
    lib.cvs_stream_sync(streams_s[1])
    lib.cvs_stream_destroy(streams_s[1])
    digest = verify.canon_sha256(bigs[0].download(stream).array)
    rec = _record(dist, gather_stats, checksum52, "scaler_x2.00", "1920x1080 f16 -> video_scale_bilinear (triangle) x2 -> 3840x2160 f16",
                  n * len(srcs), w * h, dt, digest, verify.stream_fixture("scaler_1920x1080_x2.00", g0), 10,
                  "Mpixels/s and bytes are per OUTPUT pixel: source read once (2 B per output px) + target written once; both passes in one launch; "
                  "frames alternate over two HIP streams", rank,
                  kernels=["k_fir_tile_vh<2, 2, f16> (one launch: a workgroup per 128 columns x 64 lines, source rows in LDS)"])
    if rec:
        out.append(rec)
    for d in srcs + bigs:
        d.free()

    # ---------------------------------------------------------------- config 4
    w8, h8, nl = 7680, 4320, 3
    full8 = (0, 0, w8 - 1, h8 - 1)
    slots = []
    donors0 = [ls[0] for _o, ls in ring]            # opaque layer-0 frames of config 2's ring
    donors1 = [ls[1] for _o, ls in ring]
    for s in range(2):
        layers = []
        for k in range(nl):
            d = DeviceFrame(full8, np.uint16)
            if s == 0:
                d.upload(synth.layer_pixels(w8, h8, k, g0))
            else:                                    # four 4K frames make one 8K frame's worth of pixels
                pool = donors0 if k == 0 else donors1
                for q in range(4):
                    src = pool[(4 * k + q) % len(pool)]
                    _lib.check(lib.cvs_memcpy_d2d(d.ptr + q * src.nbytes, src.ptr, src.nbytes, stream), "d2d")
            layers.append(d)
        slots.append((DeviceFrame(full8, np.uint16), layers))

    def pass4():
        chain_color_over(slots, None, _lib.LUT_NONE, _lib.LUT_NONE, stream)

    n, dt = _timed_passes(lib, _lib, [stream], pass4, seconds)
    digest = verify.canon_sha256(slots[0][0].download(stream).array)
    rec = _record(dist, gather_stats, checksum52, "config4", "7680x4320 f16 RGBA 3-layer alpha-over stack, f16 out",
                  n * len(slots), w8 * h8, dt, digest, verify.stream_fixture("config4_7680x4320", g0), 8 * (nl + 1),
                  "fused chain kernel, plain stack (no colour stage); 2 frame sets of 1.06 GB rotate", rank, kernels=["k_chain<3 layers, plain> (one launch per frame)"])
    if rec:
        out.append(rec)
    for o, ls in slots:
        o.free()
        for l in ls:
            l.free()

    # ---------------------------------------------------------------- config 5
    donors = [d for pair in zip(donors0, donors1) for d in pair]
    g = GraphStream(w, h, ring=8, matrix=matrix, first_frame=g0, exact_slots=1, donors=donors)      # 8 x 6 x 66 MB = 3.2 GB
    streams = [stream, lib.cvs_stream_create()]
    assert g.ring % len(streams) == 0          # slot i always on stream i % 2: no two streams ever share a slot's buffers

    # the ring's frames are independent: their blur + over launches go as ONE batch (cvs_blur_over_f16_batch_dev); two
    # batches alternate over two streams
    half = g.ring // 2
    groups = [list(range(0, half)), list(range(half, g.ring))]
    gs = [GraphStreamView(g, grp) for grp in groups]

    def pass5():
        for k, view in enumerate(gs):
            view.render(streams[k % len(streams)])

    for flavour, sfx in (("separate", ""), ("contracted", "_contracted")):
        with arithmetic(lib, _lib, flavour):
            n, dt = _timed_passes(lib, _lib, streams, pass5, seconds)
        digest = verify.canon_sha256(g.slots[0]["out"].download(streams[0]).array)
        rec = _record(dist, gather_stats, checksum52, "config5" + sfx, "3840x2160 10-node graph (4 sources, colour -> blur -> 4-step composite), frame stream",
                      n * g.ring, w * h, dt, digest, verify.stream_fixture("config5_3840x2160" + ("@contracted" if sfx else ""), g0), NODE_BYTES_PER_PIXEL,
                      "72 B/px is BASELINE's per-node denominator; the launches move %d B/px; eight frames in two groups of four on two HIP streams: "
                      "a colour launch per frame, one blur + over launch per group (cvs_blur_over_f16_batch_dev)" % BYTES_PER_PIXEL,
                      rank, {"moved_bytes_per_px": BYTES_PER_PIXEL, "arithmetic": flavour}, moved_bytes_per_px=BYTES_PER_PIXEL,
                      kernels=["k_color_flat (8 r + 8 w)", "k_blur_pair<9 taps, 3 layers> (two columns per lane; 8 r + 3 x 8 r + 8 w; four frames per launch)"])
        if rec:
            out.append(rec)
    lib.cvs_stream_destroy(streams[1])
    return out
