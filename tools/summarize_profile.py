#!/usr/bin/env python3
"""gpurun_out/prof_* (from tools/profile_bench.sh) -> profiles/<round>/<tag>_{kernel_stats.csv,pmc.json} + profiles/hbm_traffic.json
usage: python tools/summarize_profile.py r01 bench_v2 "description of the kernel version" """
import collections, csv, glob, json, os, shutil, sys

rnd, tag, desc = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "profiles", rnd), exist_ok=True)
newest = lambda pattern: max(glob.glob(os.path.join(root, pattern)), key=os.path.getmtime)   # gpurun merges keep older runs beside the new one
stats = newest("gpurun_out/prof_trace/*/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(root, "profiles", rnd, tag + "_kernel_stats.csv"))
out = {}
for d in ["prof_pmc1", "prof_pmc2", "prof_fetch", "prof_write"]:
    f = newest("gpurun_out/%s/*/*_counter_collection.csv" % d)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_chain" in r["Kernel_Name"] and "k_chain_tail" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
FRAMES = 8             # frames per LAUNCH: bench.py's 64-frame step is cut into launches of 8 (chain_ops.hip kBytesPerLaunch)
px = FRAMES * 3840 * 2160
fetch = out["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
write = out["WRITE_SIZE"]["mean_per_launch"] * 1024
json.dump({"kernel": desc, "command": "tools/profile_bench.sh (rocprofv3 --kernel-trace --pmc <one group per run> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra; 64 frames per step = 8 launches of 8 frames)",
           "pixels_per_launch": px, "counters": out,
           "derived": {"fetch_bytes_per_launch_x2_corrected": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                       "algorithmic_bytes_per_launch": px * 24,
                       "note": "FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM): doubled. WRITE_SIZE is exact for 16-B-per-lane streaming stores."}},
          open(os.path.join(root, "profiles", rnd, tag + "_pmc.json"), "w"), indent=1)
json.dump({"k_chain_bytes_per_output_pixel": round((fetch + write) / px, 4), "k_chain_bytes_per_launch": fetch + write, "frames_per_launch": FRAMES,
           "pixels_per_launch": px, "source": "profiles/%s/%s_pmc.json" % (rnd, tag)},
          open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"))
print(open(os.path.join(root, "profiles", rnd, tag + "_kernel_stats.csv")).read()[:400])
print("traffic B/px", (fetch + write) / px)
