#!/usr/bin/env python3
"""Instruction histogram of the kernels of a built object whose mangled name contains every given substring:
     kernel_isa.py blur_pair_ops.hip.o Li9ELi64ELb1E [--dump FILE]
Static counts (the whole body, rare paths included); --dump writes the disassembly of the first match."""
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
args = sys.argv[1:]
dump = None
if "--dump" in args:
    k = args.index("--dump")
    dump = args[k + 1]
    del args[k:k + 2]
obj = os.path.join(ROOT, "canvas_amd", "csrc", "build", args[0])
pats = args[1:]
with tempfile.TemporaryDirectory() as tmp:
    shutil.copy(obj, tmp)
    subprocess.run([LLVM + "/llvm-objdump", "-d", "--offloading", os.path.basename(obj)], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for co in glob.glob(os.path.join(tmp, "*gfx950")):
        text = subprocess.run([LLVM + "/llvm-objdump", "-d", co], stdout=subprocess.PIPE, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n", text, re.M):
            name = m.group(1)
            if not all(p in name for p in pats):
                continue
            body = text[m.end():]
            end = body.find("\n\n")
            body = body if end < 0 else body[:end]
            lines = [l for l in body.splitlines() if re.match(r"\s+[a-z]", l)]
            c = collections.Counter(l.split()[0] for l in lines)
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            salu = sum(v for k, v in c.items() if k.startswith("s_"))
            print("%s\n  %d instructions: VALU %d, SALU %d, s_barrier %d, s_nop %d, s_waitcnt %d, LDS %d, VMEM %d" % (
                name, len(lines), valu, salu, c["s_barrier"], c["s_nop"], c["s_waitcnt"],
                sum(v for k, v in c.items() if k.startswith("ds_")), sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))))
            print("  " + ", ".join("%d×%s" % (v, k) for k, v in c.most_common(28)))
            if dump:
                open(dump, "w").write(body)
                dump = None
