#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of the kernels in a built object: kernel_regs.py blur_ops.hip.o [name substring]"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(ROOT, "canvas_amd", "csrc", "build", sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    shutil.copy(obj, tmp)
    subprocess.run([LLVM + "/llvm-objdump", "-d", "--offloading", os.path.basename(obj)], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for co in glob.glob(os.path.join(tmp, "*gfx950")):
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
            if pat in f.get("name", ""):
                print("%-70s vgpr %3s sgpr %3s scratch %s lds %s spills %s" % (f["name"][:70], f.get("vgpr_count"), f.get("sgpr_count"),
                      f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size"), f.get("vgpr_spill_count")))
