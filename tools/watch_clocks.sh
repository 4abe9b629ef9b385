#!/bin/bash
# the chain kernel's level beside the GPU's clocks while it runs: N processes one after the other, rocm-smi polled meanwhile
for i in 1 2 3 4; do
  timeout -k 10 120 python tools/time_steady_state.py 3000 > gpurun_out/steady_$i.txt 2>&1 &
  pid=$!
  sleep 6
  for k in 1 2; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "fclk|mclk|sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 1; done
  wait $pid
  grep "launches 1500" gpurun_out/steady_$i.txt
done
