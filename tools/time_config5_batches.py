"""BASELINE config 5 (colour launch per frame + blur-with-three-overlays) per frame: single calls on 1-4 streams against
one blur + over launch per group of frames (cvs_blur_over_f16_batch_dev) on 1-4 streams (profiles/r03/config5_batches.txt)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("CANVAS_DIAG") == "1":
    from tools._diag import use_diag_library
    use_diag_library()
from canvas_amd import _lib, synth, REC709_RGB_TO_YPBPR
from canvas_amd.stream import GraphStream
import bench_extra
if os.environ.get("CANVAS_LIB"):                      # A/B runs: another build of the library (never set by the package)
    _lib.LIB_PATH = os.environ["CANVAS_LIB"]
lib = _lib.load(); _lib.check(lib.cvs_init(0)); lib.init_half()
print("library %s, arithmetic %s" % (os.path.basename(_lib.LIB_PATH), "contracted" if lib.cvs_get_arithmetic() else "separate"), flush=True)
QUICK = "quick" in sys.argv[1:]                        # only the bench's shape: batches of 4 on 2 streams
w, h = 3840, 2160
RING = 24
g = GraphStream(w, h, ring=RING, exact_slots=2, donors=None) if False else GraphStream(w, h, ring=RING, exact_slots=RING)
streams = [lib.cvs_stream_create() for _ in range(4)]
def sync():
    for s in streams: lib.cvs_stream_sync(s)
def timed(fn, frames_per_pass):
    fn(); sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(10): fn()
        sync(); best = min(best, (time.perf_counter() - t0) / (10 * frames_per_pass))
    return best * 1e3
for ns in (() if QUICK else (1, 2, 3, 4)):
    def per_frame():
        for i in range(RING): g.render(i, streams[i % ns])
    print("per frame, %d stream(s): %.4f ms per frame" % (ns, timed(per_frame, RING)), flush=True)
for per, ns in (((4, 2),) if QUICK else ((2, 2), (2, 3), (4, 1), (4, 2), (4, 3), (8, 2))):
    views = [bench_extra.GraphStreamView(g, list(range(a, a + per))) for a in range(0, RING, per)]
    def batched():
        for k, v in enumerate(views): v.render(streams[k % ns])
    print("batches of %d, %d stream(s): %.4f ms per frame" % (per, ns, timed(batched, RING)), flush=True)
