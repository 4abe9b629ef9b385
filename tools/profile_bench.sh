#!/bin/bash
# rocprofv3 evidence for the bench kernel: one kernel-trace/stats run, then one run per PMC group
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; no --pmc together with other trace domains).
# Run on the GPU box from the repo root:  bash tools/profile_bench.sh ; summaries -> tools/summarize_profile.py
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CANVAS_SYNTH_CACHE=/tmp/cs
B="python3 bench.py --no-cpu-baseline --no-extra"          # the default bench invocation (64 frames per step, launches of 8), minus the CPU leg and the other configs
rm -rf gpurun_out/prof_trace gpurun_out/prof_pmc1 gpurun_out/prof_pmc2 gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- $B --steps 12 --warmup 3 > gpurun_out/prof_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_pmc1 -- $B --steps 3 --warmup 1 > gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_pmc2 -- $B --steps 3 --warmup 1 > gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $B --steps 3 --warmup 1 > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $B --steps 3 --warmup 1 > gpurun_out/prof_write.log 2>&1
find gpurun_out -name "*.csv" | head -40
# the other configs' kernels, kernel-trace only (per-kernel average durations for DESIGN.md section 4.2)
rm -rf gpurun_out/prof_stream gpurun_out/prof_configs
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stream -- python3 tools/bench_stream.py --frames 300 > gpurun_out/prof_stream.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_configs -- python3 tools/bench_configs.py --which 3,4 > gpurun_out/prof_configs.log 2>&1
find gpurun_out/prof_stream gpurun_out/prof_configs -name "*kernel_stats.csv" | head
