#!/bin/bash
# The colour node's two forms (CVS_COLOR_HALF: 0 = sixteen waves and the whole table per CU; 1, 2 = four waves and the
# (an experiment: needs tools/experiments/color_half_table.patch applied -- it adds the CVS_COLOR_HALF switch -- and a rebuild)
# non-negative half of it, that many workgroups per CU) in config 5 as the bench calls it, both arithmetic flavours.
out=gpurun_out/r4/ab_color_half.txt; mkdir -p gpurun_out/r4; : > $out
for round in 1 2; do
  for form in 0 1 2; do
    for fl in separate contracted; do
      echo "=== round $round  CVS_COLOR_HALF=$form  $fl" >> $out
      CVS_COLOR_HALF=$form CVS_ARITHMETIC=$fl timeout -k 10 180 python3 tools/time_config5_batches.py quick >> $out 2>&1 || exit 1
    done
  done
done
cat $out
