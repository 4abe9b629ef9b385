#!/bin/bash
# Instruction-cache picture of one kernel: requests, hits, misses per launch, next to the waves' issue-stall cycles.
#   usage (GPU box, repo root): bash tools/profile_icache.sh <kernel-name substring> <tag> -- python3 tools/<script>.py [args]
pat="$1"; tag="$2"; shift 3
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/${tag}_i1 gpurun_out/${tag}_i2
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d gpurun_out/${tag}_i1 -- "$@" > gpurun_out/${tag}_i1.log 2>&1 || { tail -5 gpurun_out/${tag}_i1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${tag}_i2 -- "$@" > gpurun_out/${tag}_i2.log 2>&1 || { tail -5 gpurun_out/${tag}_i2.log; exit 1; }
python3 - "$pat" "$tag" <<'PY'
import csv, glob, collections, sys
pat, tag = sys.argv[1], sys.argv[2]
for d in ("i1", "i2"):
    fs = glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (tag, d), recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, v in sorted(agg.items()):
        print("%-30s %14.1f  (%d launches)" % (n, sum(v) / len(v), len(v)))
PY
