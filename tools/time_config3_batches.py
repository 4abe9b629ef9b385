"""BASELINE config 3 (4K f16 -> 9-tap blur -> Lanczos3 halving, one sweep) per frame, against the number of independent
frames handed to cvs_blur_lanczos_f16_batch_dev per call and the number of HIP streams the calls alternate over
(profiles/r03/config3_batches.txt).  CANVAS_DIAG=1 + CVS_BLUR_HALVE_ROWS pins the segment height on the diagnostic build."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("CANVAS_DIAG") == "1":
    from tools._diag import use_diag_library
    use_diag_library()
from canvas_amd import _lib, synth
from canvas_amd.device import DeviceFrame
if os.environ.get("CANVAS_LIB"):                      # A/B runs: another build of the library (never set by the package)
    _lib.LIB_PATH = os.environ["CANVAS_LIB"]
lib = _lib.load(); _lib.check(lib.cvs_init(0)); lib.init_half()
print("library %s, arithmetic %s" % (os.path.basename(_lib.LIB_PATH), "contracted" if lib.cvs_get_arithmetic() else "separate"), flush=True)
w, h = 3840, 2160
N = 24
srcs = [DeviceFrame.from_host(synth.layer_frame(w, h, 1, g % 2)) for g in range(N)]
outs = [DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16) for _ in range(N)]
taps = synth.gaussian_taps(9, 1.5)
tp = taps.ctypes.data_as(C.POINTER(C.c_float))
fp = C.POINTER(_lib.rgba_frame_f16_t)
streams = [lib.cvs_stream_create() for _ in range(4)]
combos = ((1, 1), (2, 1), (4, 1), (8, 1), (1, 2), (4, 2), (8, 2), (8, 3))
if len(sys.argv) > 2:                                   # time_config3_batches.py <frames per call> <streams>: that one only (profiling)
    combos = ((int(sys.argv[1]), int(sys.argv[2])),)
for per, ns in combos:
    tabs = []
    for a in range(0, N, per):
        tabs.append(((fp * per)(*[C.pointer(o.c) for o in outs[a:a + per]]), (fp * per)(*[C.pointer(o.c) for o in srcs[a:a + per]])))
    def run(reps):
        for _ in range(reps):
            for k, (d, s_) in enumerate(tabs):
                _lib.check(lib.cvs_blur_lanczos_f16_batch_dev(d, s_, per, tp, 9, C.c_float(0.5), C.c_float(0.5), 3, streams[k % ns]))
        for st in streams: lib.cvs_stream_sync(st)
    run(2)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); run(10); best = min(best, (time.perf_counter() - t0) / (10 * N))
    print("config 3, %d frame(s) per call, %d stream(s): %.4f ms per frame" % (per, ns, best * 1e3), flush=True)
