#!/bin/bash
# placement experiment: does the distance between the frames of one job (layer A, layer B, output) matter to HBM throughput?
# usage (on the GPU box): bash tools/ab_slot_pad.sh > gpurun_out/slot_pad.txt
for pad in 0 256 4096 36864 69632 1052672 3149824; do
  echo "### slot pad $pad"
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --slot-pad $pad | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['same_run_dtod_copy_GBps'])" || exit 1
done
