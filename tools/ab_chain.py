#!/usr/bin/env python3
"""Interleaved A/B of the chain kernel variants in ONE process (CVS_CHAIN_VARIANT is read per call).
usage: python tools/ab_chain.py [--variants 0,1,2] [--rounds 6] [--steps 10] [--layers 2]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from canvas_amd import REC709_RGB_TO_YPBPR, _lib, synth  # noqa: E402
from canvas_amd.device import DeviceFrame, chain_color_over  # noqa: E402
from tools._diag import use_diag_library  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,1,2")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lut", default="rec709", help="rec709 | none")
    args = ap.parse_args()
    variants = [v for v in args.variants.split(",")]
    use_diag_library()
    lib = _lib.load()
    _lib.check(lib.cvs_init(0))
    lib.init_half()
    stream = lib.cvs_stream_create()
    w, h, nl = args.width, args.height, args.layers
    full = (0, 0, w - 1, h - 1)
    m = np.array(REC709_RGB_TO_YPBPR, np.float32)
    ring = []
    for g in range(args.batch):
        layers = []
        for k in range(nl):
            d = DeviceFrame(full, np.uint16)
            d.upload(synth.layer_pixels(w, h, k, g))
            layers.append(d)
        ring.append((DeviceFrame(full, np.uint16), layers))
    ref = None
    times = {v: [] for v in variants}
    e0, e1 = lib.cvs_event_create(), lib.cvs_event_create()
    for r in range(args.rounds + 1):
        for v in variants:
            os.environ["CVS_CHAIN_VARIANT"] = v.split(":")[0]
            parts = v.split(":")                      # variant[:gridmul[:block]]
            os.environ["CVS_DIAG_GRIDMUL"] = parts[1] if len(parts) > 1 and parts[1] else "1"
            os.environ["CVS_CHAIN_BLOCK"] = parts[2] if len(parts) > 2 else os.environ.get("AB_DEFAULT_BLOCK", "512")
            lib.cvs_event_record(e0, stream)
            for _ in range(args.steps):
                chain_color_over(ring, m, _lib.LUT_REC709_TO_LINEAR_SCENE if args.lut == 'rec709' else _lib.LUT_NONE, _lib.LUT_NONE, stream)
            lib.cvs_event_record(e1, stream)
            _lib.check(lib.cvs_stream_sync(stream))
            ms = lib.cvs_event_elapsed_ms(e0, e1) / args.steps
            if r > 0:
                times[v].append(ms)
            out = ring[0][0].download(stream).array
            if ref is None:
                ref = out.copy()
            elif int(v.split(":")[0]) < 10 and not np.array_equal(ref, out):
                print("variant %s: OUTPUT DIFFERS from variant %s" % (v, variants[0]))
    px = args.batch * w * h
    for v in variants:
        t = np.array(times[v])
        gbs = px * 8 * (nl + 1) / (np.median(t) * 1e-3) / 1e9
        print("variant %s: median %.4f ms  min %.4f ms  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s), %.0f Mpx/s" % (
            v, np.median(t), t.min(), gbs, gbs / 80.0, px / (np.median(t) * 1e-3) / 1e6))


if __name__ == "__main__":
    main()
