#!/bin/bash
# placement experiment: the same bench command in separate processes, each with a different amount of device memory
# allocated before the ring of frames, logging the ring's base address beside the result.
# usage (on the GPU box): bash tools/ab_placement.sh > gpurun_out/placement.txt
for pre in 0 0 0 2 6 14 30 62 64 126 254 510 1022 1024 2046 4096 0 0 8190 16384 0 34 98 0; do
  timeout -k 10 150 python bench.py --no-cpu-baseline --steps 12 --warmup 4 --pre-alloc-mb $pre --arena-align-mb 0 --report-base 2>gpurun_out/placement.err \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pre %5d MiB' % $pre, d['roofline']['frac'], open('gpurun_out/placement.err').read().strip().splitlines()[-1])" || exit 1
done
