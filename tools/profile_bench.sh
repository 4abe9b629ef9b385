cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_pmc1 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_pmc2 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1
find gpurun_out -name "*.csv" | head -40
