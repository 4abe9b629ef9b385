out=gpurun_out/r4/scaler_batch.txt; mkdir -p gpurun_out/r4; : > $out
for st in 1 2; do for b in 1 2 4 8; do echo "=== batch $b, $st stream(s)" >> $out
  for c in "1080p->4K" "4K->2x" "4K->1.5"; do timeout -k 10 150 python3 tools/time_scaler.py --only "$c" --streams $st --batch $b --reps 40 >> $out 2>&1 || exit 1; done; done; done
cat $out
