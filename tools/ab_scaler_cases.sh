#!/bin/bash
# A/B of library builds over every case of tools/time_scaler.py, interleaved.   usage: bash tools/ab_scaler_cases.sh <tag> libA.so libB.so [-- extra time_scaler args]
tag=$1; shift
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" = "--" ] && shift
out=gpurun_out/r4/ab_cases_$tag.txt; mkdir -p gpurun_out/r4; : > $out
for round in 1 2; do
  for lib in "${libs[@]}"; do
    echo "=== round $round  $lib" >> $out
    CANVAS_LIB=$PWD/$lib timeout -k 10 300 python3 tools/time_scaler.py --reps 60 "$@" >> $out 2>&1 || exit 1
  done
done
python3 - $out <<'PY'
import re, sys, collections
cur, res, order = None, collections.defaultdict(lambda: collections.defaultdict(list)), []
for l in open(sys.argv[1]):
    m = re.match(r"=== round \d+\s+(\S+)", l)
    if m: cur = m.group(1); continue
    m = re.match(r"(\S+)\s+(f16|f32)\s+([\d.]+) ms.*kernel=(\S+)", l)
    if m:
        key = (m.group(1), m.group(2))
        if key not in order: order.append(key)
        res[key][cur].append((float(m.group(3)), m.group(4)))
for key in order:
    print("%-16s %s  " % key + "   ".join("%s: %.4f ms (%s)" % (lib.split("/")[-1], min(v[0] for v in vs), vs[0][1]) for lib, vs in res[key].items()))
PY
