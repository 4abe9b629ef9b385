#!/usr/bin/env python3
"""bench.py on the DIAGNOSTIC build of the library (tools/bin/libcanvas_hip_diag.so), so that the CVS_CHAIN_* knobs
apply: launch-size sweeps, block sizes, timing-only kernel variants.  Same arguments as bench.py.
    CVS_CHAIN_LAUNCH_MB=800 python3 tools/bench_diag.py --no-cpu-baseline --steps 20"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._diag import use_diag_library  # noqa: E402

use_diag_library()
import bench  # noqa: E402

bench.main()
