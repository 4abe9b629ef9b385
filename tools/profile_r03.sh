#!/bin/bash
# Round-3 evidence (GPU box, repo root):  bash tools/profile_r03.sh ; then python3 tools/summarize_r03.py on the build side.
#   1. per-launch kernel trace of the default bench region (not only --stats): the steady-state launches of the SAME run
#      whose JSON line is kept beside it, so that mean x launches_per_step can be held against that run's ms_per_step;
#   2. FETCH_SIZE / WRITE_SIZE of the chain kernel (one counter group per run, nothing but --kernel-trace beside --pmc);
#   3. the extras' kernels (config 3 / 5, scaler, Lanczos): per-launch trace, FETCH_SIZE, WRITE_SIZE, two SQ groups;
#   4. calibration of FETCH_SIZE / WRITE_SIZE for 8-byte-per-lane accesses on launches of known size (tools/calibrate_fetch.py).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CANVAS_SYNTH_CACHE=/tmp/cs
O=gpurun_out/r3p
rm -rf $O && mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace_bench.json 2> $O/trace.log || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 3 --warmup 1 > $O/fetch_bench.json 2> $O/fetch.log || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 3 --warmup 1 > $O/write_bench.json 2> $O/write.log || exit 1
echo "chain done"
E="python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --extra-seconds 0.1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_trace -- $E > $O/x_trace_bench.json 2> $O/x_trace.log || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/x_fetch -- $E > $O/x_fetch_bench.json 2> $O/x_fetch.log || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/x_write -- $E > $O/x_write_bench.json 2> $O/x_write.log || exit 1
echo "extras traffic done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/x_sq1 -- $E > $O/x_sq1_bench.json 2> $O/x_sq1.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH --output-format csv -d $O/x_sq2 -- $E > $O/x_sq2_bench.json 2> $O/x_sq2.log || exit 1
echo "extras sq done"
# config 5's two kernels one at a time (the bench alternates frames over two streams, so its kernels overlap in a trace)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_trace -- python3 tools/bench_stream.py --frames 200 --streams 1 > $O/c5_trace.json 2> $O/c5_trace.log || exit 1
# config 3's sweep four frames per launch on ONE stream (in the bench two such launches overlap on two streams)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_trace -- python3 tools/time_config3_batches.py 4 1 > $O/c3_trace.txt 2> $O/c3_trace.log || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- python3 tools/calibrate_fetch.py > $O/cal_fetch.txt 2> $O/cal_fetch.log || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- python3 tools/calibrate_fetch.py > $O/cal_write.txt 2> $O/cal_write.log || exit 1
find $O -name "*.csv" | wc -l
du -sh $O
