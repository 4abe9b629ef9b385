#!/bin/bash
# HBM traffic of the FIR kernels (configs 3 and 5): one rocprofv3 run per counter (FETCH_SIZE and WRITE_SIZE cannot
# share a pass).  Run on the GPU box from the repo root; summary -> tools/summarize_fir.py
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/fir_$c gpurun_out/stream_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/fir_$c -- python tools/bench_configs.py --which 3 --reps 3 > gpurun_out/fir_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/stream_$c -- python tools/bench_stream.py --frames 20 --streams 1 > gpurun_out/stream_$c.log 2>&1
done
find gpurun_out/fir_* gpurun_out/stream_* -name "*counter_collection.csv" | head
