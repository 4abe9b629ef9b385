"""The reference's own scaler (video_scale_bilinear_f32 and its f16 twin) at the factors the editor uses, per call:
ms, TB/s of the bytes under the window the call reports (a reduction produces part of the target only, as the reference
does), and which kernel took it (cvs_fir_last_kernel).  Sources rotate over 1.6 GB (the scaler kernels'
stores are non-temporal: the targets do not sweep the Infinity Cache, so the sources must not fit it on their own).   usage: python3 tools/time_scaler.py [--reps 40] [--only NAME] [--strips]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import _lib, synth                      # noqa: E402
from canvas_amd.abi import v2f                          # noqa: E402
from canvas_amd.device import DeviceFrame               # noqa: E402

NAMES = ["none", "window", "halve", "lanes", "vh", "tiled", "stream", "two-pass", "pass", "hv", "window-pair", "halve-pair", "tile-vh"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--only", default="")
    ap.add_argument("--streams", type=int, default=1, help="frames alternate over this many HIP streams (the bench's scaler record uses 2)")
    ap.add_argument("--diag-mode", type=int, default=0, help="diagnostic build only: k_fir_tile_vh's timing-only variants (1: stores only)")
    ap.add_argument("--batch", type=int, default=1, help="frames per call through cvs_scale_bilinear_*_batch_dev (1: the single-frame entry)")
    ap.add_argument("--tiles", action="store_true", help="pin it to the tile form wherever that takes the call (CVS_FIR_PATH_TILES)")
    ap.add_argument("--strips", action="store_true", help="pin the vertical-first scaler to k_fir_vh's strips (CVS_FIR_PATH_STRIPS)")
    args = ap.parse_args()
    if args.diag_mode:
        from tools._diag import use_diag_library
        use_diag_library()
    if os.environ.get("CANVAS_LIB"):                      # A/B runs: another build of the library (never set by the package)
        _lib.LIB_PATH = os.environ["CANVAS_LIB"]
    lib = _lib.load()
    assert lib.cvs_init(0) == 0
    lib.init_half()
    if args.strips:
        lib.cvs_fir_path_override(_lib.FIR_PATH_STRIPS)
    if args.tiles:
        lib.cvs_fir_path_override(_lib.FIR_PATH_TILES)
    if args.diag_mode:
        lib.cvk_fir_tvh_diag_mode.restype, lib.cvk_fir_tvh_diag_mode.argtypes = C.c_int, [C.c_int]
        assert lib.cvk_fir_tvh_diag_mode(args.diag_mode) == 0
    streams = [None] if args.streams <= 1 else [lib.cvs_stream_create() for _ in range(args.streams)]
    nout = max(2, 2 * len(streams)) * max(args.batch, 1)

    def sync():
        for st in streams:
            lib.cvs_stream_sync(st)
    cases = [("1080p->4K", (1920, 1080), (2.0, 2.0)), ("4K->1080p", (3840, 2160), (0.5, 0.5)), ("4K->0.75", (3840, 2160), (0.75, 0.75)),
             ("4K->1.5", (3840, 2160), (1.5, 1.5)), ("4K->2x", (3840, 2160), (2.0, 2.0)), ("4K->0.4", (3840, 2160), (0.4, 0.4)),
             ("4K->0.3", (3840, 2160), (0.3, 0.3)), ("4K->0.75x1.5", (3840, 2160), (0.75, 1.5)), ("4K->1.25x1.125", (3840, 2160), (1.25, 1.125))]
    for name, (w, h), fac in cases:
        if args.only and args.only not in name:
            continue
        tw, th = int(w * fac[0]), int(h * fac[1])
        for fmt in ("f16", "f32"):
            bpp = 8 if fmt == "f16" else 16
            nsrc = min(max(2, int(1.6e9 // (w * h * bpp)) + 1), 100)     # the sources alone exceed the 256 MiB Infinity Cache several times over
            host = synth.layer_frame(w, h, 1, 0)
            srcs = []
            for k in range(nsrc):
                d16 = DeviceFrame.from_host(host) if k == 0 else None
                if fmt == "f16":
                    if k == 0:
                        srcs.append(d16)
                    else:
                        d = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
                        _lib.check(lib.cvs_memcpy_d2d(d.ptr, srcs[0].ptr, d.nbytes, None))
                        srcs.append(d)
                else:
                    d = DeviceFrame((0, 0, w - 1, h - 1), np.float32)
                    if k == 0:
                        _lib.check(lib.cvs_frame_f16_to_f32_dev(d.ref(), d16.ref(), None))
                        lib.cvs_stream_sync(None)
                        d16.free()
                    else:
                        _lib.check(lib.cvs_memcpy_d2d(d.ptr, srcs[0].ptr, d.nbytes, None))
                    srcs.append(d)
            outs = [DeviceFrame((0, 0, tw - 1, th - 1), np.uint16 if fmt == "f16" else np.float32) for _ in range(nout)]
            lib.cvs_stream_sync(None)
            call = lib.cvs_scale_bilinear_f16_dev if fmt == "f16" else lib.cvs_scale_bilinear_f32_dev

            B = max(args.batch, 1)
            if B > 1:
                bcall = lib.cvs_scale_bilinear_f16_batch_dev if fmt == "f16" else lib.cvs_scale_bilinear_f32_batch_dev
                ft = _lib.rgba_frame_f16_t if fmt == "f16" else _lib.rgba_frame_f32_t
                ngroups = nout // B
                otabs = [(C.POINTER(ft) * B)(*[C.pointer(outs[g * B + k].c) for k in range(B)]) for g in range(ngroups)]
                stabs = [(C.POINTER(ft) * B)(*[C.pointer(srcs[(g * B + k) % nsrc].c) for k in range(B)]) for g in range(max(1, nsrc // B) if nsrc >= B else 1)]

            def run(i):
                if B > 1:
                    _lib.check(bcall(otabs[i % ngroups], v2f(0, 0), stabs[i % len(stabs)], v2f(0, 0), v2f(*fac), B, streams[i % len(streams)]))
                else:
                    _lib.check(call(outs[i % nout].ref(), v2f(0, 0), srcs[i % nsrc].ref(), v2f(0, 0), v2f(*fac), streams[i % len(streams)]))
            for i in range(3):
                run(i)
            sync()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for i in range(args.reps):
                    run(i)
                sync()
                best = min(best, (time.perf_counter() - t0) / (args.reps * B))
            # bytes: the window the call reports (video_scale.c:254-277 sizes the frame between the passes with `* factor`: a
            # reduction produces only part of the target) and the source pixels under it
            cw = outs[0].c.current_window
            out_px = max(cw.max.x - cw.min.x + 1, 0) * max(cw.max.y - cw.min.y + 1, 0)
            src_px = min(w * h, int(out_px / (fac[0] * fac[1])))
            nbytes = (src_px + out_px) * bpp
            print("%-16s %s  %.4f ms  %.2f TB/s (%.3f of 8)  window %dx%d of %dx%d  kernel=%s fused=%d" % (
                name, fmt, best * 1e3, nbytes / best / 1e12, nbytes / best / 8e12, cw.max.x - cw.min.x + 1, cw.max.y - cw.min.y + 1, tw, th,
                NAMES[lib.cvs_fir_last_kernel()], lib.cvs_scale_last_was_fused()), flush=True)
            for d in srcs + outs:
                d.free()


if __name__ == "__main__":
    main()
