#!/bin/bash
# The enlarging scaler in its two forms (tiles: k_fir_tile_vh, strips: k_fir_vh), one and two streams -> profiles/r04/scaler_forms.txt
out=gpurun_out/r4/scaler_forms.txt; mkdir -p gpurun_out/r4; : > $out
for st in 1 2; do
  for form in tiles strips; do
    echo "=== $form, $st stream(s)" >> $out
    a="--tiles"; [ $form = strips ] && a="--strips"
    for c in "1080p->4K" "4K->1.5" "4K->2x" "4K->1.25x1.125"; do timeout -k 10 120 python3 tools/time_scaler.py --only "$c" --streams $st --reps 60 $a >> $out 2>&1 || exit 1; done
  done
done
cat $out
