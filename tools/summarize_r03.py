#!/usr/bin/env python3
"""gpurun_out/r3p/* (tools/profile_r03.sh on a GPU box) -> profiles/r03/*.  usage: python3 tools/summarize_r03.py [r03]

  bench_trace_steady.json   the chain kernel launch by launch in the SAME run whose JSON line is kept beside it:
                            warm-up launches dropped, steady mean / min / max, mean x launches_per_step against that
                            run's ms_per_step (the check VERDICT r02 asked for: the profile must fit the clock)
  bench_kernel_stats.csv    rocprofv3 --stats of that run (averages include the warm-up launches: not the figure to quote)
  bench_pmc.json            FETCH_SIZE / WRITE_SIZE of the chain kernel, per launch; profiles/hbm_traffic.json refreshed
  fetch_calibration.json    FETCH_SIZE / WRITE_SIZE against launches of known size at 8 and 16 bytes per lane
  extras_kernels.json       every kernel of the extras (configs 3 / 4 / 5, scaler, Lanczos): steady durations, FETCH / WRITE
                            bytes per launch with the calibrated factor for its access width, SQ picture per wave
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", "r3p")
DST = os.path.join(ROOT, "profiles", rnd)
os.makedirs(DST, exist_ok=True)


def one(pattern):
    fs = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not fs:
        raise SystemExit("missing " + pattern)
    return max(fs, key=os.path.getmtime)


def trace(dirname):
    rows = list(csv.DictReader(open(one("%s/**/*kernel_trace.csv" % dirname))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]


def counters(dirname):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one("%s/**/*counter_collection.csv" % dirname))):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)


def stats(v):
    return {"n": len(v), "mean_us": round(sum(v) / len(v), 2), "min_us": round(min(v), 2), "max_us": round(max(v), 2)}


# ------------------------------------------------------------------ 1. the chain kernel, launch by launch
bench = last_json(os.path.join(SRC, "trace_bench.json"))
lps = bench["roofline"]["launches_per_step"]
chain = [d for n, d in trace("trace") if "k_chain<" in n and "tail" not in n]
warm = bench["warmup"] * lps
steady = chain[warm:warm + bench["steps"] * lps]
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
st = stats(steady)
out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extra --steps %d --warmup %d" % (bench["steps"], bench["warmup"]),
       "kernel": bench["roofline"]["kernel"], "launches_in_trace": len(chain), "warmup_launches_dropped": warm,
       "launches_per_step": lps, "steady": st,
       "steady_mean_x_launches_per_step_ms": round(st["mean_us"] * lps / 1e3, 4),
       "same_run_ms_per_step_wall": bench["ms_per_step"], "same_run_step_ms_hip_events": bench["roofline"]["step_ms"],
       "fits_inside_the_step": st["mean_us"] * lps / 1e3 <= bench["ms_per_step"],
       "algorithmic_bytes_per_launch": algo,
       "achieved_GBps_steady_mean": round(algo / (st["mean_us"] * 1e-6) / 1e9, 1),
       "frac_of_8TBps_steady_mean": round(algo / (st["mean_us"] * 1e-6) / 8e12, 4),
       "frac_of_8TBps_fastest_launch": round(algo / (st["min_us"] * 1e-6) / 8e12, 4),
       "warmup_launches": stats(chain[:warm]) if warm else None,
       "same_run_bench_line": {k: bench[k] for k in ("value", "ms_per_step", "steps", "warmup")}}
json.dump(out, open(os.path.join(DST, "bench_trace_steady.json"), "w"), indent=1)
shutil.copy(one("trace/**/*kernel_stats.csv"), os.path.join(DST, "bench_kernel_stats.csv"))
print("chain: steady mean %.1f us x %d = %.4f ms against ms_per_step %.4f (%s)" % (st["mean_us"], lps, st["mean_us"] * lps / 1e3, bench["ms_per_step"], "fits" if out["fits_inside_the_step"] else "DOES NOT FIT"))

# ------------------------------------------------------------------ 2. chain traffic
pm = {}
for d in ("fetch", "write"):
    for name, cs in counters(d).items():
        if "k_chain<" in name and "tail" not in name:
            for c, v in cs.items():
                pm[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
px = 8 * 3840 * 2160
fetch = pm["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
write = pm["WRITE_SIZE"]["mean_per_launch"] * 1024
json.dump({"kernel": bench["roofline"]["kernel"], "pixels_per_launch": px, "counters": pm,
           "derived": {"fetch_bytes_per_launch_x2_corrected": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                       "algorithmic_bytes_per_launch": px * 24, "ratio": round((fetch + write) / (px * 24), 4),
                       "note": "FETCH_SIZE is in KiB and counts 64 B per 128-B request for 16-B-per-lane loads on gfx950 (MI355X_MICROARCH.md, HBM): doubled. "
                               "WRITE_SIZE is exact for 16-B-per-lane streaming stores."}},
          open(os.path.join(DST, "bench_pmc.json"), "w"), indent=1)
json.dump({"k_chain_bytes_per_output_pixel": round((fetch + write) / px, 4), "k_chain_bytes_per_launch": fetch + write, "frames_per_launch": 8,
           "pixels_per_launch": px, "source": "profiles/%s/bench_pmc.json" % rnd}, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"))
print("chain traffic: %.3f B/px" % ((fetch + write) / px))

# ------------------------------------------------------------------ 3. calibration of the byte counters at 8 B per lane
FRAME16, FRAME32 = 3840 * 2160 * 8, 3840 * 2160 * 16
known = {"k_copy16": (FRAME16, FRAME16, "8 B/lane loads, 8 B/lane stores"), "k_widen": (FRAME16, FRAME32, "8 B/lane loads, 16 B/lane stores"),
         "k_narrow": (FRAME32, FRAME16, "16 B/lane loads, 8 B/lane stores")}
cal = {}
cf, cw = counters("cal_fetch"), counters("cal_write")
for key, (rd, wr, what) in known.items():
    f = [v for n, cs in cf.items() if key in n for v in cs.get("FETCH_SIZE", [])]
    w_ = [v for n, cs in cw.items() if key in n for v in cs.get("WRITE_SIZE", [])]
    if not f or not w_:
        continue
    fm, wm = sum(f) / len(f) * 1024, sum(w_) / len(w_) * 1024
    cal[key] = {"access": what, "launches": len(f), "known_read_bytes": rd, "FETCH_SIZE_bytes": fm, "read_factor": round(rd / fm, 4),
                "known_written_bytes": wr, "WRITE_SIZE_bytes": wm, "write_factor": round(wr / wm, 4)}
json.dump({"what": "bytes really read / written by launches of known size divided by what FETCH_SIZE / WRITE_SIZE report (x 1024): the factor to apply",
           "kernels": cal}, open(os.path.join(DST, "fetch_calibration.json"), "w"), indent=1)
for k, v in cal.items():
    print("calibration %-9s read factor %.3f  write factor %.3f" % (k, v["read_factor"], v["write_factor"]))
rf8 = cal.get("k_copy16", {}).get("read_factor", 2.0)
wf8 = cal.get("k_copy16", {}).get("write_factor", 1.0)
rf16 = cal.get("k_narrow", {}).get("read_factor", 2.0)

# ------------------------------------------------------------------ 4. the extras' kernels
xbench = last_json(os.path.join(SRC, "x_trace_bench.json"))
tr = trace("x_trace")
fe, wr_, s1, s2 = counters("x_fetch"), counters("x_write"), counters("x_sq1"), counters("x_sq2")
# kernel -> (bytes per lane of its loads, of its stores, algorithmic bytes per launch as the extras run it, note)
PX4K, PX1080, PX8K = 3840 * 2160, 1920 * 1080, 7680 * 4320
# (the extras run config 3's sweep and config 5's blur + over four frames per launch: their FETCH / WRITE figures are per
# launch of four, their durations come from one-stream traces -- c3_trace: four frames per launch; c5_trace: one)
KERNELS = [
    ("k_blur_halve_pair<", 16, 8, 4 * (PX4K * 8 + PX1080 * 8), "config 3: four 4K f16 frames in, four 1080p f16 frames out per launch (16-byte loads, a lane stores one pixel: 8 B)"),
    ("k_color_flat", 16, 16, PX4K * 16, "config 5, launch 1: colour filter 8 r + 8 w per px (pixel pairs: 16 B per lane)"),
    ("k_blur_pair<9, 64, 3>", 16, 16, PX4K * 40, "config 5, launch 2: blur + 3 overlays + store = 8 r + 24 r + 8 w per px (two columns per lane: 16 B accesses)"),
    ("k_fir_vh<", 8, 16, PX1080 * 8 + PX4K * 8, "scaler 1080p -> 4K f16"),
    ("k_chain<3, 1", 16, 16, PX8K * 32, "config 4: three 8K layers in, one out"),
]
try:
    tr_c5 = trace("c5_trace")
except SystemExit:
    tr_c5 = None
try:
    tr_c3 = trace("c3_trace")
except SystemExit:
    tr_c3 = None
try:
    tr_vh = trace("vh_up16_p1")                           # tools/profile_kernel.sh on tools/time_scaler.py --only "1080p->4K": one stream
except SystemExit:
    tr_vh = None
xs = {}
for pat, lb, sb, algo_b, note in KERNELS:
    d = [t for n, t in tr if pat in n]
    if tr_c5 is not None and ("k_color_flat" in pat or "k_blur_pair<9" in pat):
        d = [t for n, t in tr_c5 if pat in n]            # one stream: the kernel alone on the chip
        note += "; durations from tools/bench_stream.py --streams 1 (one frame per launch; in the bench its frames alternate over two streams and overlap)"
    if tr_vh is not None and "k_fir_vh" in pat:
        d = [t for n, t in tr_vh if "k_fir_vh<2, 2, 2, true" in n]
        note += "; durations from tools/time_scaler.py --only 1080p->4K under tools/profile_kernel.sh (one stream, counters on; in the bench its frames alternate over two streams)"
    if tr_c3 is not None and "k_blur_halve" in pat:
        d = [t for n, t in tr_c3 if pat in n]
        note += "; durations from tools/time_config3_batches.py 4 1 (four frames per launch, one stream; in the bench two such launches overlap on two streams)"
    if not d:
        continue
    d = d[len(d) // 5:]                                   # the first fifth of a kernel's launches are its warm-up passes
    rec = {"note": note, "launches_steady": len(d), "duration": stats(d), "algorithmic_bytes_per_launch": algo_b,
           "frac_of_8TBps_on_algorithmic_bytes": round(algo_b / (sum(d) / len(d) * 1e-6) / 8e12, 4)}
    f = [v for n, cs in fe.items() if pat in n for v in cs.get("FETCH_SIZE", [])]
    w_ = [v for n, cs in wr_.items() if pat in n for v in cs.get("WRITE_SIZE", [])]
    if f and w_:
        rfac = rf16 if lb == 16 else rf8
        wfac = 1.0 if sb == 16 else wf8
        fb, wb = sum(f) / len(f) * 1024 * rfac, sum(w_) / len(w_) * 1024 * wfac
        if "k_blur_pair<9" in pat:                            # counted on the extras' launches of four frames
            fb, wb = fb / 4, wb / 4
        rec["traffic"] = {"FETCH_SIZE_KiB_mean": round(sum(f) / len(f), 1), "read_factor_applied": rfac, "read_bytes": round(fb),
                          "WRITE_SIZE_KiB_mean": round(sum(w_) / len(w_), 1), "write_factor_applied": wfac, "written_bytes": round(wb),
                          "hbm_bytes_per_launch": round(fb + wb), "ratio_to_algorithmic": round((fb + wb) / algo_b, 3),
                          "loads_bytes_per_lane": lb, "stores_bytes_per_lane": sb,
                          "note": "factors from fetch_calibration.json (launches of known size with the same bytes per lane)"}
    sq = {}
    for src in (s1, s2):
        for n, cs in src.items():
            if pat in n:
                for c, v in cs.items():
                    sq[c] = sum(v) / len(v)
    if sq.get("SQ_WAVES"):
        wv = sq["SQ_WAVES"]
        rec["per_wave"] = {k.replace("SQ_INSTS_", "").lower(): round(sq[k] / wv, 1) for k in sq if k.startswith("SQ_INSTS_")}
        rec["waves"] = wv
    if sq.get("SQ_WAVE_CYCLES"):
        wc = sq["SQ_WAVE_CYCLES"]
        rec["shares_of_wave_cycles"] = {"valu_active": round(sq.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "waiting": round(sq.get("SQ_WAIT_ANY", 0) / wc, 3),
                                        "issue_stall": round(sq.get("SQ_WAIT_INST_ANY", 0) / wc, 3), "lds_active": round(sq.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3)}
        if sq.get("SQ_ACTIVE_INST_LDS"):
            rec["lds_bank_conflict_share_of_lds_cycles"] = round(sq.get("SQ_LDS_BANK_CONFLICT", 0) / sq["SQ_ACTIVE_INST_LDS"], 3)
    xs[pat] = rec
    print("%-34s %7.1f us  frac(alg) %.3f  traffic x%s" % (pat, rec["duration"]["mean_us"], rec["frac_of_8TBps_on_algorithmic_bytes"],
                                                           rec.get("traffic", {}).get("ratio_to_algorithmic")))
json.dump({"command": "rocprofv3 --kernel-trace [--stats | --pmc <one group>] -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --extra-seconds 0.1",
           "kernels": xs, "extras_of_the_traced_run": [{k: e.get(k) for k in ("config", "ms_per_frame_per_gpu", "roofline")} for e in xbench.get("extra", [])]},
          open(os.path.join(DST, "extras_kernels.json"), "w"), indent=1)
shutil.copy(one("x_trace/**/*kernel_stats.csv"), os.path.join(DST, "extras_kernel_stats.csv"))
