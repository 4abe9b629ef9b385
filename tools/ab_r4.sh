#!/bin/bash
# A/B of library builds x arithmetic flavours on the three shapes round 4 works on (config 3 as the bench calls it, the blur +
# over launch alone, config 5 as the bench calls it).   usage: bash tools/ab_r4.sh <tag> lib1.so [lib2.so ...]
tag=$1; shift
out=gpurun_out/r4/ab_$tag.txt
mkdir -p gpurun_out/r4
: > $out
for lib in "$@"; do
  for fl in separate contracted; do
    echo "=== $lib $fl" >> $out
    CANVAS_LIB=$PWD/$lib CVS_ARITHMETIC=$fl timeout -k 10 120 python3 tools/time_config3_batches.py 4 2 >> $out 2>&1 || exit 1
    CANVAS_LIB=$PWD/$lib CVS_ARITHMETIC=$fl timeout -k 10 120 python3 tools/time_blur_over.py 9 >> $out 2>&1 || exit 1
    CANVAS_LIB=$PWD/$lib CVS_ARITHMETIC=$fl timeout -k 10 180 python3 tools/time_config5_batches.py quick >> $out 2>&1 || exit 1
  done
done
cat $out
