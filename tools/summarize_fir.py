#!/usr/bin/env python3
"""gpurun_out/{fir,stream}_{FETCH,WRITE}_SIZE (tools/profile_fir.sh) -> profiles/<round>/fir_traffic.json:
measured HBM bytes per launch of every FIR / colour kernel of configs 3 and 5 beside its algorithmic bytes."""
import collections, csv, glob, json, os, re, sys

rnd = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
newest = lambda pattern: max(glob.glob(os.path.join(root, pattern)), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for run in ("fir", "stream"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in csv.DictReader(open(newest("gpurun_out/%s_%s/*/*_counter_collection.csv" % (run, c)))):
            name = r["Kernel_Name"]
            m = re.search(r"(k_blur<[^>]*>|k_color_flat<[^>]*>)", name)
            if m:
                agg[m.group(1)][c].append(float(r["Counter_Value"]))
px4k, px1080 = 3840 * 2160, 1920 * 1080
algorithmic = {   # bytes per launch at 3840x2160
    "k_blur<9, 256, true, false, 1>": px4k * (8 + 16), "k_blur<9, 256, false, false, 1>": px4k * (16 + 16),
    "k_blur<11, 256, false, true, 2>": px4k * 16 + px1080 * 8, "k_blur<11, 256, false, false, 2>": px4k * 16 + px1080 * 16,
    "k_blur<9, 256, true, true, 1>": px4k * (8 + 3 * 8 + 8), "k_color_flat<true, false>": px4k * 16,
}
out = {}
for k, v in sorted(agg.items()):
    f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1) * 1024 * 2      # KiB, 64 B counted per 128-B request on gfx950
    w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1) * 1024
    out[k] = {"launches": len(v["FETCH_SIZE"]), "fetch_bytes_x2_corrected": round(f), "write_bytes": round(w), "hbm_bytes": round(f + w),
              "algorithmic_bytes": algorithmic.get(k), "ratio": round((f + w) / algorithmic[k], 3) if k in algorithmic else None}
os.makedirs(os.path.join(root, "profiles", rnd), exist_ok=True)
json.dump({"command": "tools/profile_fir.sh", "note": "FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM); f16 8-byte-per-lane loads are outside the calibrated 16-B case, read the ratios as indicative",
           "kernels": out}, open(os.path.join(root, "profiles", rnd, "fir_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(k, v)
