"""BASELINE config 5: a 10-node graph over a synthetic frame stream, on device frames.

    S0 ─ colour(Rec.709 LUT + RGB→Y'PbPr) ─ blur(9-tap separable Gaussian) ─┐
    S1, S2, S3 ─────────────────────────────────────────────────────────────┴ workspace stack → output

Ten nodes = four sources + colour(S0) + blur(colour) + four composite steps (the base fetch and three
overs, as workspace.c:530-544 counts them).  In the reference this is a colour filter on an f16 source
(color.c structure), a blur node that pulls it as f32 (main.c:105-144 widening) and is itself the lowest
workspace item (fetched as f32, no rounding), three f16 sources blended over it with
video_mix_over_f32(mix 1.0), and a consumer that pulls f16 (main.c:43-71 truncation).

Node-by-node compulsory traffic: colour 8+8, blur 8+8, composite 4x8+8 = 72 B per output pixel.
Here it is two launches per frame:

    cvs_color_matrix_f16_to_dev   8 B read + 8 B written per pixel
    cvs_blur_over_f16_dev         8 B (graded) + 3 x 8 B (upper layers) read + 8 B written   => 56 B per pixel

Frames are independent, so frame g belongs to rank g % world (shard.frames_of_rank); nothing is
exchanged on the data path.
"""
import ctypes as C

import numpy as np

from . import _lib, synth
from .device import DeviceFrame
from .shard import frames_of_rank

NODE_BYTES_PER_PIXEL = 16 + 16 + 40      # per-node compulsory traffic (the survey's denominator)
BYTES_PER_PIXEL = 16 + 40                 # what the two launches move
OVERLAYS = 3


class GraphStream:
    """Device-resident ring of input sets plus the intermediates of one in-flight frame per slot."""

    def __init__(self, width, height, ring=4, matrix=None, pre_lut=_lib.LUT_REC709_TO_LINEAR_SCENE,
                 post_lut=_lib.LUT_NONE, taps=None, overlays=OVERLAYS, first_frame=0, frame_step=1, exact_slots=None, donors=None):
        """Slot s holds the inputs of stream frame first_frame + s * frame_step.  exact_slots / donors: only the first
        `exact_slots` slots get the generator's frames; the others are filled on the device with copies of `donors`
        (resident f16 frames of the same size) -- same value distribution, no host generation, for throughput runs that
        check slot 0 only."""
        from . import REC709_RGB_TO_YPBPR
        self.lib = _lib.load()
        self.w, self.h, self.ring, self.overlays = width, height, ring, overlays
        self.full = (0, 0, width - 1, height - 1)
        self.matrix = np.ascontiguousarray(REC709_RGB_TO_YPBPR if matrix is None else matrix, np.float32).reshape(9)
        self.pre_lut, self.post_lut = pre_lut, post_lut
        self.taps = np.ascontiguousarray(synth.gaussian_taps(9, 1.5) if taps is None else taps, np.float32)
        self.slots = []
        nd = 0
        for slot in range(ring):
            exact = exact_slots is None or slot < exact_slots or not donors
            frames = []
            for k in range(overlays + 1):
                d = DeviceFrame(self.full, np.uint16)
                if exact:
                    d.upload(synth.layer_pixels(width, height, k, first_frame + slot * frame_step))
                else:
                    # the bottom layer must stay opaque (BASELINE input): donors[0::2] are layer-0 frames by convention
                    src_frame = donors[(2 * nd) % len(donors)] if k == 0 else donors[(2 * nd + 1) % len(donors)]
                    nd += 1
                    _lib.check(self.lib.cvs_memcpy_d2d(d.ptr, src_frame.ptr, d.nbytes, None), "d2d")
                frames.append(d)
            src, over = frames[0], frames[1:]
            refs = (C.POINTER(_lib.rgba_frame_f16_t) * overlays)(*[C.pointer(o.c) for o in over])
            self.slots.append({"src": src, "over": over, "over_refs": refs, "graded": DeviceFrame(self.full, np.uint16),
                               "out": DeviceFrame(self.full, np.uint16)})
        # the donor copies ran on this thread's own stream; the slots are rendered on whatever streams the caller
        # passes to render(): finish the copies before anyone reads them
        _lib.check(self.lib.cvs_stream_sync(None), "donor copies")
        self._m = self.matrix.ctypes.data_as(C.POINTER(C.c_float))
        self._t = self.taps.ctypes.data_as(C.POINTER(C.c_float))

    @staticmethod
    def host_inputs(width, height, frame, overlays=OVERLAYS):
        """The synthetic inputs of stream frame `frame` as host frames (for checking a rendered frame)."""
        return [synth.layer_frame(width, height, k, frame) for k in range(overlays + 1)]

    def render(self, slot, stream=None):
        """Enqueue the two launches of one frame on `stream`; returns the output DeviceFrame."""
        s, lib = self.slots[slot % self.ring], self.lib
        _lib.check(lib.cvs_color_matrix_f16_to_dev(s["graded"].ref(), s["src"].ref(), self._m, self.pre_lut, self.post_lut, stream), "colour")
        _lib.check(lib.cvs_blur_over_f16_dev(s["out"].ref(), s["graded"].ref(), self._t, len(self.taps), s["over_refs"],
                                             self.overlays, stream), "blur+over")
        return s["out"]

    def render_batch(self, slots, stream=None):
        """The same frames as render() of each slot, with the blur + over launches of all of them as ONE call of the batch
        entry (up to eight frames of one geometry per launch: taller row segments, fewer halo rows re-filtered).  The
        colour launches stay one per frame (they are HBM-bound and have no halo).  Returns the output DeviceFrames."""
        lib, n = self.lib, len(slots)
        ss = [self.slots[i % self.ring] for i in slots]
        for s in ss:
            _lib.check(lib.cvs_color_matrix_f16_to_dev(s["graded"].ref(), s["src"].ref(), self._m, self.pre_lut, self.post_lut, stream), "colour")
        key = tuple(i % self.ring for i in slots)
        if getattr(self, "_batch_key", None) != key:                      # pointer tables for this set of slots, built once
            fp = C.POINTER(_lib.rgba_frame_f16_t)
            self._b_out = (fp * n)(*[C.pointer(s["out"].c) for s in ss])
            self._b_src = (fp * n)(*[C.pointer(s["graded"].c) for s in ss])
            self._b_ov = (fp * (n * self.overlays))(*[C.pointer(o.c) for s in ss for o in s["over"]])
            self._batch_key = key
        _lib.check(lib.cvs_blur_over_f16_batch_dev(self._b_out, self._b_src, self._t, len(self.taps), self._b_ov, self.overlays, n, stream), "blur+over batch")
        return [s["out"] for s in ss]

    def run(self, frames_per_rank, rank=0, world=1, stream=None):
        """Render `frames_per_rank` frames on this rank (weak scaling: the stream has frames_per_rank * world frames,
        global frame g belongs to rank g % world); returns how many it rendered."""
        mine = frames_of_rank(rank, world, frames_per_rank)
        for i, _g in enumerate(mine):
            self.render(i, stream)
        return len(mine)
