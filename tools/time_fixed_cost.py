import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth
from canvas_amd.device import DeviceFrame, chain_color_over
lib = _lib.load(); lib.cvs_init(0); lib.init_half()
st = lib.cvs_stream_create()
m = np.array(REC709_RGB_TO_YPBPR, np.float32)
e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
for (w, h) in [(64, 36), (3840, 2160)]:
    layers = [DeviceFrame.from_host(synth.layer_frame(w, h, k, 0)) for k in range(2)]
    out = DeviceFrame((0, 0, w - 1, h - 1), np.uint16)
    for name, mat, pre in [("lut+matrix", m, 0), ("matrix only", m, -1), ("plain", None, -1)]:
        ts = []
        for _ in range(12):
            lib.cvs_event_record(e0, st)
            chain_color_over([(out, layers)], mat, pre, -1, st)
            lib.cvs_event_record(e1, st)
            lib.cvs_stream_sync(st)
            ts.append(lib.cvs_event_elapsed_ms(e0, e1))
        print(w, h, name, "median ms", round(float(np.median(ts[2:])), 4))
