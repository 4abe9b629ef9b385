#!/bin/bash
# A/B of library builds on the enlarging scaler, interleaved so that both see the same box and the same minute.
#   usage: bash tools/ab_scaler.sh <tag> libA.so libB.so [...]     (paths from the repo root; CANVAS_LIB selects the build)
tag=$1; shift
out=gpurun_out/r4/ab_scaler_$tag.txt
mkdir -p gpurun_out/r4
: > $out
for round in 1 2 3; do
  for lib in "$@"; do
    for st in 1 2; do
      echo "=== round $round  $lib  streams $st" >> $out
      CANVAS_LIB=$PWD/$lib timeout -k 10 120 python3 tools/time_scaler.py --only "1080p->4K" --streams $st --reps 80 >> $out 2>&1 || exit 1
    done
  done
done
python3 - $out <<'PY'
import re, sys, collections
cur, res = None, collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"=== round \d+\s+(\S+)\s+streams (\d)", l)
    if m: cur = (m.group(1), m.group(2)); continue
    m = re.match(r"\S+\s+(f16|f32)\s+([\d.]+) ms", l)
    if m: res[cur + (m.group(1),)].append(float(m.group(2)))
for k in sorted(res): print("%-34s streams %s %s  median %.4f ms   all %s" % (k[0], k[1], k[2], sorted(res[k])[len(res[k]) // 2], " ".join("%.4f" % v for v in res[k])))
PY
