#!/usr/bin/env python3
"""Averages of the rocprofv3 counters tools/profile_kernel.sh collected, for the launches of one kernel.
usage: summarize_kernel_pmc.py <tag> <kernel name substring (demangled, e.g. 'k_fir_lanes<12, 16, 1, true')>"""
import collections
import csv
import glob
import sys

tag, pat = sys.argv[1], sys.argv[2]
tot = {}
for d in ("p1", "p2", "p3"):
    fs = glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (tag, d), recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    for n, v in agg.items():
        tot[n] = sum(v) / len(v)
    kt = glob.glob("gpurun_out/%s_%s/**/*kernel_trace.csv" % (tag, d), recursive=True)[0]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if pat in r["Kernel_Name"]]
    if durs:
        tot["duration_us_" + d] = sum(durs) / len(durs)
        tot["launches_" + d] = len(durs)
    if meta and d == "p1":
        print("kernel %s\n  grid %s, workgroup %s, LDS %s, VGPRs %s (+%s), SGPRs %s" % (meta["Kernel_Name"][:90], meta["Grid_Size"], meta["Workgroup_Size"], meta["LDS_Block_Size"], meta["VGPR_Count"], meta["Accum_VGPR_Count"], meta["SGPR_Count"]))
for k in sorted(tot):
    print("%-28s %16.1f" % (k, tot[k]))
w = tot.get("SQ_WAVE_CYCLES", 0)
if w:
    print("VALU-active share of wave cycles  %.3f" % (tot["SQ_ACTIVE_INST_VALU"] / w))
    print("waiting share (s_waitcnt/barrier) %.3f" % (tot["SQ_WAIT_ANY"] / w))
    print("issue-stall share                 %.3f" % (tot["SQ_WAIT_INST_ANY"] / w))
if "SQ_INSTS_VALU" in tot and "SQ_WAVES" in tot:
    n = tot["SQ_WAVES"]
    print("per wave: VALU %.0f  SALU %.0f  SMEM %.0f  LDS %.0f  VMEM rd %.0f wr %.0f  branch %.0f" % tuple(tot.get(k, 0) / n for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")))
    if w:
        print("wave cycles per wave %.0f; per instruction %.1f" % (w / n, w / (tot["SQ_INSTS_VALU"] + tot.get("SQ_INSTS_SALU", 0) + tot.get("SQ_INSTS_LDS", 0) + tot.get("SQ_INSTS_SMEM", 0) + tot.get("SQ_INSTS_VMEM_RD", 0) + tot.get("SQ_INSTS_VMEM_WR", 0) + tot.get("SQ_INSTS_BRANCH", 0))))
