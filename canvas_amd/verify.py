"""Checking a rendered frame against a committed fixture, for callers outside tests/ (bench.py's per-rank proof).

The fixtures (tests/golden/*.json) hold SHA-256 digests of the expected outputs for the synthetic stream's frames at the
BASELINE configs' full sizes, made in the build container by tests/golden/make_checksums.py (the CPU checker of tests/).  Hashing happens after
canonicalisation of the two things the reference build does not pin: the sign of zero (-fno-signed-zeros, SConstruct:82-83)
and NaN payload / sign (x86 and gfx950 default NaNs differ).  Nothing here computes pixels: it hashes what the library rendered and compares."""
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def canon_f16(codes):
    codes = np.ascontiguousarray(codes, np.uint16).copy()
    mag = codes & 0x7FFF
    codes[mag == 0] = 0
    codes[mag > 0x7C00] = 0x7E00
    return codes


def canon_sha256(codes):
    return hashlib.sha256(canon_f16(codes).tobytes()).hexdigest()


_cache = {}


def stream_fixture(config, frame):
    """Digest of stream frame `frame` of `config` ("config2_3840x2160", ...), or None when no fixture holds it."""
    if "stream" not in _cache:
        path = os.path.join(GOLDEN_DIR, "stream_frames_sha256.json")
        _cache["stream"] = json.load(open(path)) if os.path.exists(path) else {}
    return _cache["stream"].get(config, {}).get(str(frame))
