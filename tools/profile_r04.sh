#!/bin/bash
# Round-4 evidence (GPU box, repo root):  bash tools/profile_r04.sh ; then python3 tools/summarize_r04.py on the build side.
# Every figure a claim rests on comes from bench.py's OWN launch shape (config 3: four frames per launch on two streams; config
# 5: a colour launch per frame + one blur-and-over launch per four frames on two streams; scaler: two calls in flight):
#   t_*   --kernel-trace --stats of the default headline run (config 2) and its FETCH_SIZE / WRITE_SIZE passes;
#   x_*   the extras of one bench run: --kernel-trace (per-frame time = span of a kernel's steady launches / frames), FETCH_SIZE,
#         WRITE_SIZE, and two SQ groups -- ONE set, the shipped workgroup shapes, both arithmetic flavours (the bench runs
#         config 3 and config 5 in the default flavour first, then contracted).
# One counter group per run, nothing but --kernel-trace beside --pmc.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CANVAS_SYNTH_CACHE=/tmp/cs
O=gpurun_out/r4p
rm -rf $O && mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_trace -- $B > $O/t_trace_bench.json 2> $O/t_trace.log || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/t_fetch -- $B --steps 3 --warmup 1 > $O/t_fetch_bench.json 2> $O/t_fetch.log || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/t_write -- $B --steps 3 --warmup 1 > $O/t_write_bench.json 2> $O/t_write.log || exit 1
echo "chain done"
E="python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --extra-seconds 0.5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_trace -- $E > $O/x_trace_bench.json 2> $O/x_trace.log || exit 1
echo "extras trace done"
E="python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --extra-seconds 0.1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/x_fetch -- $E > $O/x_fetch_bench.json 2> $O/x_fetch.log || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/x_write -- $E > $O/x_write_bench.json 2> $O/x_write.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/x_sq1 -- $E > $O/x_sq1_bench.json 2> $O/x_sq1.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $O/x_sq2 -- $E > $O/x_sq2_bench.json 2> $O/x_sq2.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $O/x_sq3 -- $E > $O/x_sq3_bench.json 2> $O/x_sq3.log || exit 1
echo "extras counters done"
find $O -name "*.csv" | wc -l
du -sh $O
