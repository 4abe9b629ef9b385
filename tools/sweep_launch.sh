export CANVAS_SYNTH_CACHE=/tmp/cs
run() { timeout -k 10 100 python3 tools/bench_diag.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'], d['verified_against_fixture'])" || exit 1; }
for rep in 1 2; do
for mb in 200 400 600 800 1000 1200 1519 2000 3000 6000; do CVS_CHAIN_LAUNCH_MB=$mb run "launch_mb $mb"; done
done
for b in 256 1024; do CVS_CHAIN_BLOCK=$b run "block $b"; done
for v in 10 12; do CVS_CHAIN_VARIANT=$v run "variant $v"; CVS_CHAIN_VARIANT=$v CVS_CHAIN_BLOCK=1024 run "variant $v block 1024"; done
