#!/bin/bash
# PMC picture of one kernel: instruction mix, waiting, occupancy, LDS conflicts, fetched bytes.
#   usage (GPU box, repo root): bash tools/profile_kernel.sh <kernel-name substring> <tag> -- python3 tools/<script>.py [args]
# Two counter groups in their own runs (no trace domains beside --kernel-trace), averages over the launches whose name matches.
pat="$1"; tag="$2"; shift 3
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CANVAS_SYNTH_CACHE=/tmp/cs
rm -rf gpurun_out/${tag}_p1 gpurun_out/${tag}_p2 gpurun_out/${tag}_p3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/${tag}_p1 -- "$@" > gpurun_out/${tag}_p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/${tag}_p2 -- "$@" > gpurun_out/${tag}_p2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/${tag}_p3 -- "$@" > gpurun_out/${tag}_p3.log 2>&1 || exit 1
python3 - "$pat" "$tag" <<'PY'
import csv, glob, collections, sys
pat, tag = sys.argv[1], sys.argv[2]
tot = {}
for d in ("p1", "p2", "p3"):
    fs = glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (tag, d), recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, v in agg.items():
        tot[n] = sum(v) / len(v)
    kt = glob.glob("gpurun_out/%s_%s/**/*kernel_trace.csv" % (tag, d), recursive=True)[0]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if pat in r["Kernel_Name"]]
    if durs:
        tot["duration_us_" + d] = sum(durs) / len(durs)
        tot["launches_" + d] = len(durs)
for k in sorted(tot):
    print("%-28s %16.1f" % (k, tot[k]))
w = tot.get("SQ_WAVE_CYCLES", 0)
if w:
    print("VALU-active share of wave cycles  %.3f" % (tot["SQ_ACTIVE_INST_VALU"] / w))
    print("waiting share (s_waitcnt/barrier) %.3f" % (tot["SQ_WAIT_ANY"] / w))
    print("issue-stall share                 %.3f" % (tot["SQ_WAIT_INST_ANY"] / w))
    if tot.get("SQ_BUSY_CYCLES"):
        print("waves resident per shader engine (wave quad-cycles * 4 / busy cycles) %.2f" % (4 * w / tot["SQ_BUSY_CYCLES"]))
if "SQ_INSTS_VALU" in tot and "SQ_WAVES" in tot:
    n = tot["SQ_WAVES"]
    print("per wave: VALU %.0f  SALU %.0f  SMEM %.0f  LDS %.0f  VMEM rd %.0f wr %.0f  branch %.0f" % tuple(tot.get(k, 0) / n for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")))
PY
