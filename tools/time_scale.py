import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from canvas_amd import _lib, synth
from canvas_amd.abi import v2f
from canvas_amd.device import DeviceFrame
lib = _lib.load(); lib.cvs_init(0); lib.init_half()
w, h = 3840, 2160
src16 = DeviceFrame.from_host(synth.layer_frame(w, h, 0, 0))
src = DeviceFrame((0, 0, w - 1, h - 1), np.float32)
lib.cvs_frame_f16_to_f32_dev(src.ref(), src16.ref(), None)
for fac, tsz in [((0.5, 0.5), (1920, 1080)), ((2.0, 2.0), (3840, 2160)), ((0.25, 0.5), (960, 1080))]:
    s = src if fac[0] < 1 else DeviceFrame((0, 0, 1919, 1079), np.float32)
    out = DeviceFrame((0, 0, tsz[0] - 1, tsz[1] - 1), np.float32)
    for rep in range(3):
        lib.cvs_stream_sync(None); t0 = time.perf_counter()
        for _ in range(5):
            _lib.check(lib.cvs_scale_bilinear_f32_dev(out.ref(), v2f(0, 0), s.ref(), v2f(0, 0), v2f(*fac), None))
        lib.cvs_stream_sync(None); dt = (time.perf_counter() - t0) / 5
    print(fac, tsz, "ms per call", round(dt * 1e3, 3), out.current_window.tuple())
