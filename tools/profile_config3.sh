#!/bin/bash
# PMC evidence for "config 3 is bound by VALU issue": the fused blur + halving kernel under two counter groups.
#   usage (GPU box, repo root): bash tools/profile_config3.sh > gpurun_out/config3_pmc.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CANVAS_SYNTH_CACHE=/tmp/cs
rm -rf gpurun_out/c3p1 gpurun_out/c3p2 gpurun_out/c3p3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/c3p1 -- python3 tools/bench_configs.py --which 3 > gpurun_out/c3p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/c3p2 -- python3 tools/bench_configs.py --which 3 > gpurun_out/c3p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/c3p3 -- python3 tools/bench_configs.py --which 3 > gpurun_out/c3p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
tot = {}
for d in ("c3p1", "c3p2", "c3p3"):
    f = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_blur_halve" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, v in agg.items():
        tot[n] = sum(v) / len(v)
    kt = glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % d, recursive=True)[0]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if "k_blur_halve" in r["Kernel_Name"]]
    tot["duration_us_" + d] = sum(durs) / len(durs)
for k in sorted(tot):
    print("%-28s %16.1f" % (k, tot[k]))
w = tot.get("SQ_WAVE_CYCLES", 0)
if w:
    print("VALU-active share of wave cycles  %.3f" % (tot["SQ_ACTIVE_INST_VALU"] / w))
    print("waiting share (s_waitcnt/barrier) %.3f" % (tot["SQ_WAIT_ANY"] / w))
    print("issue-stall share                 %.3f" % (tot["SQ_WAIT_INST_ANY"] / w))
if "SQ_INSTS_VALU" in tot and "SQ_WAVES" in tot:
    print("VALU instructions per wave        %.0f" % (tot["SQ_INSTS_VALU"] / tot["SQ_WAVES"]))
if "FETCH_SIZE" in tot:
    print("fetched MB per launch (x2-corrected for 16-byte lanes would overstate: 8-byte f16 loads) raw KiB %.0f" % tot["FETCH_SIZE"])
PY
