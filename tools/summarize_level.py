#!/usr/bin/env python3
"""gpurun_out/level/run_* (tools/level_ab.sh) -> one table: per process its chain-kernel duration and the PMC group it
carried; per counter the correlation of its per-launch value with the launch duration across the processes of its group.
usage: python3 tools/summarize_level.py gpurun_out/level [out.json]"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

root = sys.argv[1]
runs = []
for d in sorted(glob.glob(os.path.join(root, "run_*"))):
    rec = {"run": os.path.basename(d)}
    try:
        j = json.load(open(os.path.join(d, "bench.json")))
        rec["frac_events"] = j["roofline"]["frac"]
        rec["avg_launch_ms_events"] = j["roofline"]["avg_launch_ms"]
    except Exception as e:                                   # noqa: BLE001
        rec["bench_error"] = str(e)
    try:
        rec["ring_base"] = [l for l in open(os.path.join(d, "bench.err")).read().splitlines() if l.startswith("ring base")][-1]
    except Exception:                                        # noqa: BLE001
        pass
    if os.path.exists(os.path.join(d, "clocks.txt")):
        rec["clocks"] = [l.strip() for l in open(os.path.join(d, "clocks.txt")).read().splitlines() if l.strip()][2:8]
    kt = glob.glob(os.path.join(d, "prof", "**", "*kernel_trace.csv"), recursive=True)
    if kt:
        durs = []
        for r in csv.DictReader(open(kt[0])):
            if "k_chain" in r["Kernel_Name"]:
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        if durs:
            rec["kernel_ms_median"] = statistics.median(durs)
            rec["kernel_ms_min"] = min(durs)
            rec["launches"] = len(durs)
    cc = glob.glob(os.path.join(d, "prof", "**", "*counter_collection.csv"), recursive=True)
    if cc:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(cc[0])):
            if "k_chain" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        rec["counters"] = {k: statistics.mean(v) for k, v in agg.items()}
    if os.path.exists(os.path.join(d, "group.txt")):
        rec["group"] = open(os.path.join(d, "group.txt")).read().split()
    runs.append(rec)


def corr(xs, ys):
    if len(xs) < 3 or len(set(xs)) < 2 or len(set(ys)) < 2:
        return None
    mx, my = statistics.mean(xs), statistics.mean(ys)
    sx = sum((x - mx) ** 2 for x in xs) ** 0.5
    sy = sum((y - my) ** 2 for y in ys) ** 0.5
    return sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / (sx * sy)


per_counter = {}
names = sorted({c for r in runs for c in r.get("counters", {})})
for c in names:
    pts = [(r["kernel_ms_median"], r["counters"][c]) for r in runs if c in r.get("counters", {}) and "kernel_ms_median" in r]
    if not pts:
        continue
    ds, vs = [p[0] for p in pts], [p[1] for p in pts]
    lo, hi = min(pts), max(pts)            # fastest and slowest process that carried this counter
    per_counter[c] = {"n": len(pts), "corr_with_duration": corr(ds, vs),
                      "fastest": {"ms": lo[0], "value": lo[1]}, "slowest": {"ms": hi[0], "value": hi[1]},
                      "value_ratio_slowest_over_fastest": (hi[1] / lo[1]) if lo[1] else None,
                      "duration_ratio_slowest_over_fastest": hi[0] / lo[0]}

print("%-22s %-8s %-10s %-10s %s" % ("run", "frac", "events_ms", "kernel_ms", "ring base / clocks"))
for r in runs:
    print("%-22s %-8s %-10s %-10s %s" % (r["run"], r.get("frac_events"), r.get("avg_launch_ms_events"),
                                        ("%.4f" % r["kernel_ms_median"]) if "kernel_ms_median" in r else "-",
                                        r.get("ring_base", "") + ((" | " + r["clocks"][0]) if r.get("clocks") else "")))
print()
print("%-44s %3s %8s %14s %14s %8s %8s" % ("counter (mean per launch)", "n", "corr", "fastest", "slowest", "v ratio", "t ratio"))
for c, s in sorted(per_counter.items(), key=lambda kv: -abs(kv[1]["corr_with_duration"] or 0)):
    print("%-44s %3d %8s %14.4g %14.4g %8s %8.4f" % (c, s["n"], "-" if s["corr_with_duration"] is None else "%.3f" % s["corr_with_duration"],
                                                    s["fastest"]["value"], s["slowest"]["value"],
                                                    "-" if s["value_ratio_slowest_over_fastest"] is None else "%.4f" % s["value_ratio_slowest_over_fastest"],
                                                    s["duration_ratio_slowest_over_fastest"]))
if len(sys.argv) > 2:
    json.dump({"runs": runs, "per_counter": per_counter}, open(sys.argv[2], "w"), indent=1)
