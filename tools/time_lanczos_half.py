"""Lanczos3 at 1/2 alone, 3840x2160 f16 -> 1920x1080 f16 (cvs_resample_lanczos_f16_dev): the two-column sweep behind an identity
blur (k_blur_halve_pair<1, 11>) against the decimating register-window kernel (k_blur<11, step 2>), pinned with
cvs_fir_path_override; 24 rotating sources, wall clock over 200 calls."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canvas_amd import _lib, synth
from canvas_amd.device import DeviceFrame
lib = _lib.load(); _lib.check(lib.cvs_init(0)); lib.init_half()
w, h, N = 3840, 2160, 24
srcs = [DeviceFrame.from_host(synth.layer_frame(w, h, 1, g % 2)) for g in range(N)]
outs = [DeviceFrame((0, 0, w // 2 - 1, h // 2 - 1), np.uint16) for _ in range(4)]
NAMES = {_lib.FIR_KERNEL_WINDOW: "k_blur<11, step 2>", _lib.FIR_KERNEL_HALVE_PAIR: "k_blur_halve_pair<1, 11>"}
for pin, label in ((_lib.FIR_PATH_ONE_COLUMN, "one column per lane"), (_lib.FIR_PATH_AUTO, "automatic")):
    lib.cvs_fir_path_override(pin)
    def run(n):
        for i in range(n):
            _lib.check(lib.cvs_resample_lanczos_f16_dev(outs[i & 3].ref(), srcs[i % N].ref(), C.c_float(0.5), C.c_float(0.5), 3, None))
        lib.cvs_stream_sync(None)
    run(20)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); run(200); best = min(best, (time.perf_counter() - t0) / 200)
    nbytes = (w * h + w * h // 4) * 8
    print("%-22s %.4f ms per frame  %.2f TB/s (%.3f of 8)  %s" % (label, best * 1e3, nbytes / best / 1e12, nbytes / best / 8e12, NAMES.get(lib.cvs_fir_last_kernel(), "?")), flush=True)
