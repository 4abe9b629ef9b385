#!/bin/bash
# sweep the register-window FIR kernel's launch shape on config 3 (blur 4K f32, Lanczos halving 4K -> 1080p)
for w in 256 128; do for n in 1 2 3 4 6; do
  echo "W=$w per_cu=$n $(CVS_BLUR_WIDTH=$w CVS_BLUR_WGS_PER_CU=$n timeout -k 10 120 python tools/bench_configs.py --which 3 2>&1 | python -c 'import sys,json; d=json.loads(sys.stdin.read())["config3"]; print(round(d["blur_ms"],4), round(d["scale_ms"],4), round(d["pipeline_f16_ms"],4))')"
done; done
