#!/bin/bash
# sweep the blur kernel's strip width and rows-per-workgroup on the config 5 stream (prints launch split per setting)
for w in 256 128; do for r in 0 16 24 32 48 64 96 128; do
  echo "W=$w R=$r $(CVS_BLUR_WIDTH=$w CVS_BLUR_ROWS=$r timeout -k 10 120 python tools/bench_stream.py --frames 200 --ring 2 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["launch_ms"], d["ms_per_frame"])')"
done; done
echo "generic $(CVS_BLUR_GENERIC=1 timeout -k 10 120 python tools/bench_stream.py --frames 200 --ring 2 2>&1 | tail -1)"
