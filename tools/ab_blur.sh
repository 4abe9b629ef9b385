#!/bin/bash
# sweep the blur kernel's resident workgroups per CU on the config 5 stream (prints launch split per setting)
for n in 2 3 4 5 6 8; do
  echo "per_cu=$n $(CVS_BLUR_WGS_PER_CU=$n timeout -k 10 120 python tools/bench_stream.py --frames 200 --ring 2 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["launch_ms"], d["ms_per_frame"])')"
done
