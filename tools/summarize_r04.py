#!/usr/bin/env python3
"""gpurun_out/r4p/* (tools/profile_r04.sh on a GPU box) -> profiles/r04/*.      usage: python3 tools/summarize_r04.py

  bench_trace_steady.json   the chain kernel launch by launch in the SAME default run whose JSON line is kept beside it: steady
                            mean x launches_per_step against that run's ms_per_step (the profile must fit the clock)
  bench_kernel_stats.csv    rocprofv3 --stats of that run (averages include warm-up launches)
  bench_pmc.json            FETCH_SIZE / WRITE_SIZE of the chain kernel, per launch; profiles/hbm_traffic.json refreshed
  extras_kernels.json       the extras' kernels IN THE BENCH'S OWN LAUNCH SHAPE: for every extra its
                            trace per-frame time (span of the kernel's launches of one timed region / frames they processed, two
                            streams overlapping as they do in the bench) beside the same run's ms_per_frame_per_gpu, bytes per
                            FRAME (labelled; per-launch figures too), and ONE counter set per kernel; config 3 and config 5 in
                            both arithmetic flavours (the bench runs the default flavour first, the contracted one second)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r4p")
DST = os.path.join(ROOT, "profiles", "r04")
os.makedirs(DST, exist_ok=True)
SIMDS = 256 * 4


def one(pattern):
    fs = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not fs:
        raise SystemExit("missing " + pattern)
    return max(fs, key=os.path.getmtime)


def trace_rows(dirname):
    rows = list(csv.DictReader(open(one("%s/**/*kernel_trace.csv" % dirname))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def counters(dirname):
    """kernel name -> counter -> list of per-dispatch values, in dispatch order"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = list(csv.DictReader(open(one("%s/**/*counter_collection.csv" % dirname))))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)


def stats(v):
    return {"n": len(v), "mean_us": round(sum(v) / len(v), 2), "min_us": round(min(v), 2), "max_us": round(max(v), 2)}


def dur(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


# ------------------------------------------------------------------ 1. the chain kernel, launch by launch
bench = last_json(os.path.join(SRC, "t_trace_bench.json"))
lps = bench["roofline"]["launches_per_step"]
chain = [dur(r) for r in trace_rows("t_trace") if "k_chain<" in r["Kernel_Name"] and "tail" not in r["Kernel_Name"]]
warm = bench["warmup"] * lps
steady = chain[warm:warm + bench["steps"] * lps]
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
st = stats(steady)
out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extra   (default --steps %d --warmup %d)" % (bench["steps"], bench["warmup"]),
       "kernel": bench["roofline"]["kernel"], "launches_in_trace": len(chain), "warmup_launches_dropped": warm, "launches_per_step": lps, "steady": st,
       "steady_mean_x_launches_per_step_ms": round(st["mean_us"] * lps / 1e3, 4),
       "same_run_ms_per_step_wall": bench["ms_per_step"], "same_run_step_ms_hip_events": bench["roofline"]["step_ms"],
       "fits_inside_the_step": st["mean_us"] * lps / 1e3 <= bench["ms_per_step"],
       "algorithmic_bytes_per_launch": algo,
       "achieved_GBps_steady_mean": round(algo / (st["mean_us"] * 1e-6) / 1e9, 1),
       "frac_of_8TBps_steady_mean": round(algo / (st["mean_us"] * 1e-6) / 8e12, 4),
       "frac_of_8TBps_all_launches_incl_warmup": round(algo / (sum(chain) / len(chain) * 1e-6) / 8e12, 4),
       "same_run_bench_line": {k: bench[k] for k in ("value", "ms_per_step", "steps", "warmup")}, "same_run_roofline": bench["roofline"]}
json.dump(out, open(os.path.join(DST, "bench_trace_steady.json"), "w"), indent=1)
shutil.copy(one("t_trace/**/*kernel_stats.csv"), os.path.join(DST, "bench_kernel_stats.csv"))
print("chain: steady mean %.1f us x %d = %.4f ms against ms_per_step %.4f (%s); frac %.4f" % (
    st["mean_us"], lps, st["mean_us"] * lps / 1e3, bench["ms_per_step"], "fits" if out["fits_inside_the_step"] else "DOES NOT FIT", out["frac_of_8TBps_steady_mean"]))

# ------------------------------------------------------------------ 2. chain traffic
pm = {}
for d in ("t_fetch", "t_write"):
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
                       "note": "FETCH_SIZE is in KiB and counts 64 B per 128-B request for 16-B-per-lane loads on gfx950 (MI355X_MICROARCH.md, HBM; "
                               "profiles/r03/fetch_calibration.json): doubled.  WRITE_SIZE is exact for 16-B-per-lane streaming stores."}},
          open(os.path.join(DST, "bench_pmc.json"), "w"), indent=1)
json.dump({"k_chain_bytes_per_output_pixel": round((fetch + write) / px, 4), "k_chain_bytes_per_launch": fetch + write, "frames_per_launch": 8,
           "pixels_per_launch": px, "source": "profiles/r04/bench_pmc.json"}, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"))
print("chain traffic: %.3f B/px" % ((fetch + write) / px))

# ------------------------------------------------------------------ 3. the extras in the bench's own launch shape
xbench = last_json(os.path.join(SRC, "x_trace_bench.json"))
extras = {e["config"]: e for e in xbench.get("extra", [])}
rows = trace_rows("x_trace")
PX4K, PX1080, PX8K = 3840 * 2160, 1920 * 1080, 7680 * 4320
rf16 = rf8 = 2.0                                          # FETCH_SIZE factor (profiles/r03/fetch_calibration.json: 2.0 at 8 and at 16 B per lane)


def clusters(pat, parts):
    """The dispatches of the kernel named `pat`, split into `parts` runs at the largest gaps between consecutive launches (the
    bench runs an extra's flavours one after the other with a download and a hash in between)."""
    rs = [r for r in rows if pat in r["Kernel_Name"]]
    if not rs:
        return []
    gaps = sorted(((int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"]), i) for i in range(len(rs) - 1)), reverse=True)
    cuts = sorted(i for _g, i in gaps[:parts - 1])
    out_, a = [], 0
    for c in cuts:
        out_.append(rs[a:c + 1])
        a = c + 1
    out_.append(rs[a:])
    return out_


def steady_span(rs, frames_per_launch, drop=0.25):
    """per-frame time from the trace: span of the last (1 - drop) of the launches / frames they processed (launches overlap on
    two streams: the span, not the sum of durations, is what the chip took)"""
    rs = rs[int(len(rs) * drop):]
    span = (max(int(r["End_Timestamp"]) for r in rs) - min(int(r["Start_Timestamp"]) for r in rs)) / 1e3
    return span / (len(rs) * frames_per_launch), [dur(r) for r in rs]


def counter_means(pat, part, parts):
    """ONE counter set of the kernel `pat`: mean per dispatch over the same cluster (flavour) in each counter run"""
    m = {}
    for d in ("x_fetch", "x_write", "x_sq1", "x_sq2", "x_sq3"):
        for name, cs in counters(d).items():
            if pat not in name:
                continue
            for c, v in cs.items():
                k = len(v) // parts
                seg = v[part * k:(part + 1) * k] if parts > 1 else v
                seg = seg[len(seg) // 4:] or seg
                m[c] = sum(seg) / len(seg)
    return m


# (kernel pattern, extra record, flavour parts, frames per launch, algorithmic bytes per FRAME moved by this kernel, lanes per workgroup note)
PLAN = [
    ("k_blur_halve_pair<9, 11, 128>", ["config3", "config3_contracted"], 4, PX4K * 8 + PX1080 * 8, "config 3: 4K f16 in, 1080p f16 out; 128-lane workgroups, four frames per launch, two streams"),
    ("k_blur_pair<9, 64, 3>", ["config5", "config5_contracted"], 4, PX4K * 40, "config 5, blur + 3 overlays: 8 r + 24 r + 8 w per px; 64-lane workgroups, four frames per launch, two streams"),
    ("k_color_flat<true, false>", ["config5", "config5_contracted"], 1, PX4K * 16, "config 5, colour launch: 8 r + 8 w per px, one frame per launch"),
    ("k_fir_tile_vh<2, 2, true>", ["scaler_x2.00"], 1, PX1080 * 8 + PX4K * 8, "scaler 1080p -> 4K f16, one frame per launch, two streams"),
    ("k_chain<3, 1, false, false, 0>", ["config4"], 1, PX8K * 32, "config 4: three 8K layers in, one out, one frame per launch"),
]
xs = {}
for pat, recs, fpl, algo_b, note in PLAN:
    cl = clusters(pat, len(recs))
    for part, (name, rs) in enumerate(zip(recs, cl)):
        if not rs or name not in extras:
            continue
        per_frame_us, durs = steady_span(rs, fpl)
        e = extras[name]
        bench_us = e["ms_per_frame_per_gpu"] * 1e3
        cm = counter_means(pat, part, len(recs))
        rec = {"kernel": pat, "note": note, "arithmetic": e.get("arithmetic", "separate"),
               "launches_in_timed_region": len(rs), "frames_per_launch": fpl,
               "launch_duration": stats(durs),
               "trace_us_per_frame": round(per_frame_us, 2), "bench_us_per_frame_same_run": round(bench_us, 2)}
        if pat.startswith("k_color_flat"):
            rec["nests"] = "n/a: the colour launch overlaps the blur + over launch of the other stream; config 5's per-frame time is the blur + over kernel's record"
        else:
            rec["trace_fits_inside_bench"] = per_frame_us <= bench_us * 1.001
        rec["algorithmic_bytes_per_frame"] = algo_b
        rec["frac_of_8TBps_on_these_bytes_at_trace_rate"] = round(algo_b / (per_frame_us * 1e-6) / 8e12, 4)
        if "FETCH_SIZE" in cm and "WRITE_SIZE" in cm:
            lanes16 = not pat.startswith("k_fir_")
            fb = cm["FETCH_SIZE"] * 1024 * (rf16 if lanes16 else rf8)
            wb = cm["WRITE_SIZE"] * 1024
            rec["traffic"] = {"per": "LAUNCH of %d frame(s)" % fpl, "FETCH_SIZE_KiB_per_launch": round(cm["FETCH_SIZE"], 1), "read_factor_applied": 2.0,
                              "read_bytes_per_launch": round(fb), "WRITE_SIZE_KiB_per_launch": round(cm["WRITE_SIZE"], 1), "written_bytes_per_launch": round(wb),
                              "hbm_bytes_per_FRAME": round((fb + wb) / fpl), "ratio_to_algorithmic": round((fb + wb) / fpl / algo_b, 3)}
        if cm.get("SQ_WAVES"):
            wv = cm["SQ_WAVES"]
            rec["waves_per_launch"] = round(wv)
            rec["per_wave_instructions"] = {k.replace("SQ_INSTS_", "").lower(): round(cm[k] / wv, 1) for k in cm if k.startswith("SQ_INSTS_")}
        if cm.get("SQ_WAVE_CYCLES"):
            wc = cm["SQ_WAVE_CYCLES"]
            vgpr = int(rs[0]["VGPR_Count"]) + int(rs[0].get("Accum_VGPR_Count") or 0)
            lanes = int(rs[0]["Workgroup_Size_X"])
            per_simd_by_regs = max(1, 512 // max(vgpr, 1))
            resident = min(per_simd_by_regs, cm.get("SQ_WAVES", 0) / SIMDS)
            sh = {"valu_active": round(cm.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "waiting": round(cm.get("SQ_WAIT_ANY", 0) / wc, 3),
                  "issue_stall": round(cm.get("SQ_WAIT_INST_ANY", 0) / wc, 3), "lds_active": round(cm.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3)}
            rec["shares_of_a_waves_cycles"] = sh
            rec["vgprs"] = vgpr
            rec["workgroup_lanes"] = lanes
            rec["waves_resident_per_simd"] = round(resident, 2)
            rec["valu_busy_share_of_simd_time"] = round(sh["valu_active"] * resident, 3)
            rec["valu_busy_note"] = ("a wave's VALU-active share x the waves a SIMD holds at once (%d by registers; the launch has %.1f waves per SIMD in all, "
                                     "so the SIMD is full for the whole launch)" % (per_simd_by_regs, cm.get("SQ_WAVES", 0) / SIMDS))
        if cm.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_cycles_per_lds_active_cycle"] = round(cm.get("SQ_LDS_BANK_CONFLICT", 0) / cm["SQ_LDS_IDX_ACTIVE"], 3)
        xs["%s [%s]" % (name, pat.split("<")[0])] = rec
        print("%-44s trace %7.2f us/frame  bench %7.2f  %s  valu busy %s" % (name + " " + pat.split("<")[0], per_frame_us, bench_us,
              "nests" if rec.get("trace_fits_inside_bench", True) else "DOES NOT NEST", rec.get("valu_busy_share_of_simd_time")))
json.dump({"command": "rocprofv3 --kernel-trace [--stats | --pmc <one group>] -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --extra-seconds 0.5 (0.1 for the counter runs)",
           "how": "per-frame time from the trace = span of the kernel's launches in the timed region (first quarter dropped) / frames processed, two streams overlapping as in the bench; "
                  "counters: mean per launch of the same region; bytes are labelled per launch or per frame",
           "kernels": xs, "extras_of_the_traced_run": [{k: e.get(k) for k in ("config", "arithmetic", "ms_per_frame_per_gpu", "roofline", "ranks_verified")} for e in xbench.get("extra", [])]},
          open(os.path.join(DST, "extras_kernels.json"), "w"), indent=1)
shutil.copy(one("x_trace/**/*kernel_stats.csv"), os.path.join(DST, "extras_kernel_stats.csv"))
