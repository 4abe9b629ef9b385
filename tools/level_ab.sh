#!/bin/bash
# What sets the chain kernel's per-process throughput level?  The same bench command in many processes,
# each under ONE rocprofv3 PMC group (groups interleaved round-robin so that every group sees both levels), plus
# un-profiled processes with the clocks sampled while they run.  Summary: tools/summarize_level.py -> profiles/r02/level_ab.json
#   usage (GPU box, repo root):  bash tools/level_ab.sh [passes]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CANVAS_SYNTH_CACHE=/tmp/canvas_synth
PASSES=${1:-4}
OUT=gpurun_out/level
rm -rf $OUT && mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-extra --steps 8 --warmup 3"
PMCG=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE GRBM_EA_BUSY"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_UTCL2_BUSY GRBM_TC_BUSY"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
 "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_BUSY_sum"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
)
n=0
for pass in $(seq 1 $PASSES); do
  # un-profiled process, clocks sampled while it runs
  n=$((n+1)); d=$OUT/run_$(printf %02d $n)_clocks; mkdir -p $d
  ( for k in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "fclk|mclk|sclk|socclk|Power|junction|memory" | tr '\n' ' '; echo; sleep 1.5; done ) > $d/clocks.txt 2>&1 &
  smi=$!
  timeout -k 10 150 $B --steps 3000 > $d/bench.json 2> $d/bench.err; rc=$?
  kill $smi 2>/dev/null; wait $smi 2>/dev/null
  if [ $rc -ge 124 ]; then echo "bench timed out (rc $rc)"; exit 1; fi
  echo "run $n clocks $(python3 -c "import json,sys; print(json.load(open('$d/bench.json'))['roofline']['frac'])" 2>/dev/null)"
  for g in "${!PMCG[@]}"; do
    n=$((n+1)); d=$OUT/run_$(printf %02d $n)_g$g; mkdir -p $d
    echo "${PMCG[$g]}" > $d/group.txt
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc ${PMCG[$g]} --output-format csv -d $d/prof -- $B > $d/bench.json 2> $d/bench.err; rc=$?
    if [ $rc -ge 124 ]; then echo "profiled bench timed out (rc $rc)"; exit 1; fi
    echo "run $n g$g rc $rc $(python3 -c "import json,sys; print(json.load(open('$d/bench.json'))['roofline']['frac'])" 2>/dev/null)"
    # keep only the two csv files the summary reads
    find $d/prof -type f ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
  done
done
python3 tools/summarize_level.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
