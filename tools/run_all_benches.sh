#!/bin/bash
# Every measurement DESIGN.md quotes, in one go (1 GPU).  Output: gpurun_out/all_benches.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/all_benches.txt
mkdir -p gpurun_out
{
  echo "### bench.py (config 2, default invocation)"; timeout -k 10 300 python bench.py 2>&1 | tail -1
  echo "### bench.py --translucent-base --no-cpu-baseline"; timeout -k 10 200 python bench.py --translucent-base --no-cpu-baseline 2>&1 | tail -1
  echo "### tools/bench_configs.py (configs 3, 4, config 2 node by node)"; timeout -k 10 300 python tools/bench_configs.py 2>&1
  echo "### tools/bench_stream.py --frames 600 (config 5, two streams)"; timeout -k 10 200 python tools/bench_stream.py --frames 600 2>&1 | tail -1
  echo "### tools/bench_stream.py --frames 600 --streams 1"; timeout -k 10 200 python tools/bench_stream.py --frames 600 --streams 1 2>&1 | tail -1
  echo "### tools/time_ops.py (single 4K frames, every device entry point)"; timeout -k 10 200 python tools/time_ops.py 2>&1
  echo "### tools/time_scale.py (triangle scaler)"; timeout -k 10 200 python tools/time_scale.py 2>&1
  echo "### tools/time_config1.py (config 1 through the Python surface)"; timeout -k 10 200 python tools/time_config1.py 2>&1
  echo "### tools/time_host_frames.py (config 2 from host buffers: PCIe included)"; timeout -k 10 200 python tools/time_host_frames.py 2>&1
  echo "### tools/time_fixed_cost.py (per-launch fixed cost of the chain kernel)"; timeout -k 10 200 python tools/time_fixed_cost.py 2>&1
  echo "### tools/time_graph.py (HIP graph replay against direct enqueue)"; timeout -k 10 200 python tools/time_graph.py 2>&1
  echo "### tools/time_general_fir.py (general FIR paths at 4K)"; timeout -k 10 200 python tools/time_general_fir.py 2>&1
  echo "### tools/time_blur_over.py (blur node with 0..3 layers over it)"; timeout -k 10 200 python tools/time_blur_over.py 2>&1
  echo "### tools/time_timeline.py (preview pulls through the Python surface, 1280x720)"; timeout -k 10 200 python tools/time_timeline.py 2>&1
} > $out 2>&1
tail -5 $out
