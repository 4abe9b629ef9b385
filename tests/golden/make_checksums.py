#!/usr/bin/env python3
"""SHA-256 of the oracle's outputs for the BASELINE configs at their FULL sizes, frame 0 of the
synthetic stream: tests/golden/full_size_sha256.json.  The inputs are the counter-based synthetic frames of
canvas_amd.synth, the outputs are hashed after canonicalisation (both zeros -> +0, every NaN -> 0x7E00: neither is
pinned by the reference build).  Run here, with this container's libm; the GPU box compares the library's outputs
with the committed hashes, so a libm or generator drift shows up as a mismatch rather than silently moving both sides.

    python tests/golden/make_checksums.py            # takes about a minute
    python tests/golden/make_checksums.py --stream 8 # frames 0..7 of every config -> stream_frames_sha256.json (minutes)
    python tests/golden/make_checksums.py --stream 8 lanczos3_3840x2160_x0.40 ...   # only the named cases, merged into the file
    python tests/golden/make_checksums.py --flavour contracted [--stream 8 [names]]  # the same through the checker's OTHER build
        (clang, a * b + c fused: the reference's preferred build, SConstruct:46-48); keys get the suffix "@contracted" and are
        merged into the same files -- what the library's CVS_ARITH_CONTRACTED flavour is held to
"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def canon_sha256(codes):
    from tests.util import canon_f16
    return hashlib.sha256(canon_f16(codes).tobytes()).hexdigest()


def config2(oracle, g=0):
    from canvas_amd import REC709_RGB_TO_YPBPR, synth
    layers = [synth.layer_frame(3840, 2160, k, g) for k in range(2)]
    out = oracle.chain_color_over(layers, np.array(REC709_RGB_TO_YPBPR, np.float32), oracle.transfer_table(0), None)
    return out.array


def config3(oracle, g=0):
    from canvas_amd import synth
    from canvas_amd.abi import HostFrame
    f32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    taps = synth.gaussian_taps(9, 1.5)
    src16 = synth.layer_frame(3840, 2160, 1, g)
    src32 = HostFrame(src16.full_window, np.float32, oracle.half_to_float(src16.array))
    blurred = HostFrame(src16.full_window, np.float32)
    oracle.lib().orc_fir_blur_f32(blurred.ref(), src32.ref(), f32p(taps), 9)
    small = HostFrame((0, 0, 1919, 1079), np.float32)
    oracle.lib().orc_resample_lanczos_f32(small.ref(), blurred.ref(), C.c_float(0.5), C.c_float(0.5), 3)
    return oracle.float_to_half(small.array)


def config4(oracle, g=0):
    from canvas_amd import synth
    layers = [synth.layer_frame(7680, 4320, k, g) for k in range(3)]
    return oracle.chain_color_over(layers, None).array


def config5(oracle, g=0):
    from canvas_amd import REC709_RGB_TO_YPBPR, synth
    from tests.util import oracle_graph
    layers = [synth.layer_frame(3840, 2160, k, g) for k in range(4)]
    return oracle_graph(oracle, layers, np.array(REC709_RGB_TO_YPBPR, np.float32), oracle.transfer_table(0), None, synth.gaussian_taps(9, 1.5)).array


def lanczos3(factor):
    """stream-only cases (bench.py's `extra` records of the general FIR path): Lanczos3 of layer 1 at a factor that is not 1/2"""
    def run(oracle, g=0):
        from canvas_amd import synth
        from canvas_amd.abi import HostFrame
        src16 = synth.layer_frame(3840, 2160, 1, g)
        src32 = HostFrame(src16.full_window, np.float32, oracle.half_to_float(src16.array))
        out = HostFrame((0, 0, int(3840 * factor) - 1, int(2160 * factor) - 1), np.float32)
        oracle.lib().orc_resample_lanczos_f32(out.ref(), src32.ref(), C.c_float(factor), C.c_float(factor), 3)
        return oracle.float_to_half(out.array)
    return run


def scaler(w, h, factor):
    """stream-only cases: the reference's own scaler (video_scale_bilinear_f32) between f16 frames: widen, both passes, truncate"""
    def run(oracle, g=0):
        from canvas_amd import synth
        from canvas_amd.abi import HostFrame, v2f
        src16 = synth.layer_frame(w, h, 1, g)
        src32 = HostFrame(src16.full_window, np.float32, oracle.half_to_float(src16.array))
        out = HostFrame((0, 0, int(w * factor) - 1, int(h * factor) - 1), np.float32)
        oracle.lib().orc_scale_bilinear_f32(out.ref(), v2f(0, 0), src32.ref(), v2f(0, 0), v2f(factor, factor))
        assert out.current_window.tuple() == out.full_window.tuple(), out.current_window.tuple()
        return oracle.float_to_half(out.array)
    return run


CASES = {"config2_3840x2160": config2, "config3_3840x2160_to_1920x1080": config3, "config4_7680x4320": config4, "config5_3840x2160": config5}
STREAM_ONLY = {"lanczos3_3840x2160_x0.40": lanczos3(0.4), "lanczos3_3840x2160_x0.75": lanczos3(0.75), "lanczos3_3840x2160_x1.50": lanczos3(1.5),
               "scaler_1920x1080_x2.00": scaler(1920, 1080, 2.0)}       # (reducing, the reference covers only part of the target: video_scale.c:256-262)


CONTRACTED = "@contracted"          # key suffix of the digests made by the checker's clang / contraction-on build


def _in_flavour(flavour):
    import oracle
    oracle.lib()
    return oracle.flavour("fma" if flavour == "contracted" else "gcc")


def checksums(only=None, flavour="separate"):
    import oracle
    out = {}
    sfx = CONTRACTED if flavour == "contracted" else ""
    with _in_flavour(flavour):
        for name, fn in CASES.items():
            if only and name not in only:
                continue
            t0 = time.perf_counter()
            arr = fn(oracle)
            out[name + sfx] = {"shape": list(arr.shape), "sha256": canon_sha256(arr)}
            print("%-34s %s  (%.1f s)" % (name + sfx, out[name + sfx]["sha256"][:16], time.perf_counter() - t0), file=sys.stderr)
    return out


def stream_checksums(nframes=8, only=None, flavour="separate"):
    """Digests of stream frames 0..nframes-1 of every config: what rank r of an N-GPU bench run proves its first frame
    (global frame r) against -> tests/golden/stream_frames_sha256.json."""
    import oracle
    out = {}
    sfx = CONTRACTED if flavour == "contracted" else ""
    with _in_flavour(flavour):
        for name, fn in list(CASES.items()) + list(STREAM_ONLY.items()):
            if only and name not in only:
                continue
            out[name + sfx] = {}
            for g in range(nframes):
                t0 = time.perf_counter()
                out[name + sfx][str(g)] = canon_sha256(fn(oracle, g))
                print("%-34s frame %d %s  (%.1f s)" % (name + sfx, g, out[name + sfx][str(g)][:16], time.perf_counter() - t0), file=sys.stderr)
    return out


def _merge_into(path, new):
    merged = json.load(open(path)) if os.path.exists(path) else {}
    merged.update(new)
    with open(path, "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    flavour = "separate"
    if "--flavour" in sys.argv:
        k = sys.argv.index("--flavour")
        flavour = sys.argv[k + 1]
        assert flavour in ("separate", "contracted")
        del sys.argv[k:k + 2]
        if len(sys.argv) > 1 and sys.argv[1] == "--stream":
            _merge_into(os.path.join(HERE, "stream_frames_sha256.json"), stream_checksums(int(sys.argv[2]) if len(sys.argv) > 2 else 8, sys.argv[3:] or None, flavour))
        else:
            _merge_into(os.path.join(HERE, "full_size_sha256.json"), checksums(sys.argv[1:] or None, flavour))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--stream":
        path = os.path.join(HERE, "stream_frames_sha256.json")
        only = sys.argv[3:] or None                # names after the count: only those cases, merged into the file
        new = stream_checksums(int(sys.argv[2]) if len(sys.argv) > 2 else 8, only)
        if only and os.path.exists(path):
            merged = json.load(open(path))
            merged.update(new)
            new = merged
        with open(path, "w") as f:
            json.dump(new, f, indent=1, sort_keys=True)
            f.write("\n")
        print("wrote", path)
        sys.exit(0)
    path = os.path.join(HERE, "full_size_sha256.json")
    with open(path, "w") as f:
        json.dump(checksums(), f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)
