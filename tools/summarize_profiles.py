#!/usr/bin/env python3
"""Condenses gpurun_out/profiles_<round>/ (rocprofv3 output of tools/collect_profiles.sh) into profiles/<round>_*: the
kernel_stats CSVs as they are, one JSON per arithmetic mode with the dominant kernel's counter averages and calibrated HBM
traffic (640x480 bench workload and C4), the launches per batch of the timed workload, and the depth kernels against their
algorithmic bytes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "profiles_" + rnd)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
PEAK = 8000.0


def newest(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return f[-1] if f else None


def last_json_line(path):
    run = None
    if path and os.path.exists(path):
        for line in open(path):
            if line.startswith("{"):
                try:
                    run = json.loads(line)
                except Exception:
                    pass
    return run


def counters(name, match):
    """per-kernel-family averages of every counter in one --pmc pass"""
    f = newest(os.path.join(name, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        k = match(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: dict({c: sum(x) / len(x) for c, x in v.items()}, dispatches=max(len(x) for x in v.values())) for k, v in agg.items()}


def gn_match(n):
    return "gn_fca_level0" if "gn_fca" in n else ("calib_read_f32" if "calib_read" in n else None)


for name in ("bench_trace", "c4_bench_trace", "kernel_trace_fast", "kernel_trace_exact", "c4_kernel_trace_fast", "c4_kernel_trace_exact", "depth_trace"):
    f = newest(os.path.join(name, "*", "*_kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, name)))


def pmc_summary(prefix, arith):
    fetch = counters("%spmc_fetch_%s" % (prefix, arith), gn_match)
    write = counters("%spmc_write_%s" % (prefix, arith), gn_match)
    sq = counters("%spmc_sq_%s" % (prefix, arith), gn_match) if not prefix else {}
    run = last_json_line(os.path.join(src, "%skernel_trace_%s.log" % (prefix, arith)))
    out = {"round": rnd, "source": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python3 tools/profile_kernel.py (args in profile_kernel_run)",
           "kernels": {}, "profile_kernel_run": run}
    for k in set(fetch) | set(write) | set(sq):
        d = {}
        for part, tag in ((fetch, "pmc_fetch"), (write, "pmc_write"), (sq, "pmc_sq")):
            for c, v in part.get(k, {}).items():
                d[c if c not in ("duration_us", "dispatches") else "%s_%s" % (c, tag)] = v
        out["kernels"][k] = d
    kern = out["kernels"].get("gn_fca_level0", {})
    cal = out["kernels"].get("calib_read_f32", {})
    if run and "FETCH_SIZE" in kern and "FETCH_SIZE" in cal:
        factor = run["calib_bytes_per_launch"] / (cal["FETCH_SIZE"] * 1024.0)   # known bytes / reported bytes (gfx950: 2.0)
        fb = kern["FETCH_SIZE"] * 1024.0 * factor
        wb = kern.get("WRITE_SIZE", 0.0) * 1024.0
        out["hbm_traffic"] = {"fetch_correction_factor": factor, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                              "traffic_bytes_per_launch": fb + wb, "algorithmic_bytes_per_launch": run["algorithmic_bytes"],
                              "traffic_over_algorithmic": (fb + wb) / run["algorithmic_bytes"]}
    return out


for arith in ("fast", "exact"):
    s = pmc_summary("", arith)
    if s["profile_kernel_run"]:
        json.dump(s, open(os.path.join(dst, "%s_pmc_summary_%s.json" % (rnd, arith)), "w"), indent=1)
        print(arith, json.dumps(s.get("hbm_traffic")))
c4 = {a: pmc_summary("c4_", a) for a in ("fast", "exact")}
if any(v["profile_kernel_run"] for v in c4.values()):
    json.dump(dict(round=rnd, note="level 0 of 1280x960 dense, 16 alignments (BASELINE configs[4] per GPU)", **c4),
              open(os.path.join(dst, "%s_c4_pmc_summary.json" % rnd), "w"), indent=1)
    for a, v in c4.items():
        print("c4", a, json.dumps(v.get("hbm_traffic")))

# ---- launches per batch of the timed workload (kernel trace of bench.py --steps 10 --warmup 3: 13 batches of 32 + uploads)
f = newest(os.path.join("bench_trace", "*", "*_kernel_trace.csv"))
if f:
    rows = list(csv.DictReader(open(f)))
    by = collections.Counter()
    dur = collections.defaultdict(float)
    for r in rows:
        n = r["Kernel_Name"].split("(")[0]
        by[n] += 1
        dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seqs = sum(c for n, c in by.items() if "gn_fused_finish" in n)   # one final solve per launch sequence
    # a launch sequence covers a group of batches side by side (cfg.coalesce): the finish kernel runs one block per alignment
    fin = [int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) for r in rows if "gn_fused_finish" in r["Kernel_Name"]]
    per_seq = (sum(fin) / len(fin)) if fin else 0
    rec = {"round": rnd, "source": "rocprofv3 --kernel-trace -- python3 bench.py --steps 10 --warmup 3 --trace-only",
           "launch_sequences": seqs, "alignments_per_sequence_mean": per_seq, "batches_of_32_per_sequence_mean": per_seq / 32.0, "kernels": {}}
    for n, c in sorted(by.items(), key=lambda kv: -dur[kv[0]]):
        rec["kernels"][n] = {"launches": c, "launches_per_sequence": (c / seqs) if seqs else None, "avg_us": dur[n] / c, "total_us": dur[n]}
    rec["alignment_launches_per_sequence"] = sum(v["launches_per_sequence"] for k, v in rec["kernels"].items()
                                                 if seqs and any(t in k for t in ("gn_", "prep_", "stage_in", "ica_hinv")))
    rec["alignment_launches_per_batch_of_32"] = rec["alignment_launches_per_sequence"] / (per_seq / 32.0) if per_seq else None
    json.dump(rec, open(os.path.join(dst, "%s_launches_per_batch.json" % rnd), "w"), indent=1)
    print("launches per sequence:", rec["alignment_launches_per_sequence"], "alignments per sequence:", per_seq)

# ---- depth kernels: average duration from the kernel trace, algorithmic bytes per pixel, PMC traffic
f = newest(os.path.join("depth_trace", "*", "*_kernel_trace.csv"))
if f:
    N = 640 * 480
    bpp = {"dm_regularize": 50.0, "dm_fill_holes": 50.0, "dm_observe_walk": 94.0, "dm_observe_select": None, "dm_export_level0": 21.0, "depth_pyr_level": None,
           "dm_prop_project": 61.0, "dm_prop_fold": None, "dm_rescale": None, "dm_sum_stage1": None,
           # the one-launch chains: a full map in and out (25 + 25 B/px), the export's planes where it rides along
           "dm_reg_fill_reg": 50.0, "dm_fill_reg": 50.0, "dm_export_pyramid": 9.0 + 12.0 + 8.0 / 3.0}
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("ellc::", "").replace("void ", "")
        d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

    def dm_match(n):
        n = n.split("(")[0].split("<")[0].replace("ellc::", "").replace("void ", "")
        return n if (n.startswith("dm_") or n.startswith("depth_pyr")) else None
    fe = counters("depth_pmc_fetch", dm_match)
    wr = counters("depth_pmc_write", dm_match)
    out = {"round": rnd, "source": "rocprofv3 --kernel-trace [--pmc FETCH_SIZE | WRITE_SIZE] -- python3 tools/bench_depth.py (640x480, 67 k valid hypotheses)",
           "peak_GBps": PEAK, "note": "FETCH_SIZE doubled (gfx950 calibration, see the *_pmc_summary files); 15 MB of state per stage is Infinity-Cache "
           "resident, so counter traffic can fall below the algorithmic bytes", "kernels": {}}
    for n, v in sorted(d.items()):
        if not (n.startswith("dm_") or n.startswith("depth_pyr")):
            continue
        us = sum(v) / len(v)
        k = {"launches": len(v), "avg_us": us}
        b = bpp.get(n)
        if b:
            k.update(algorithmic_bytes_per_px=b, algorithmic_bytes=b * N, achieved_GBps=b * N / us / 1e3, frac_of_hbm_peak=b * N / us / 1e3 / PEAK)
        if n in fe and "FETCH_SIZE" in fe[n]:
            k["pmc_fetch_bytes"] = fe[n]["FETCH_SIZE"] * 1024.0 * 2.0
        if n in wr and "WRITE_SIZE" in wr[n]:
            k["pmc_write_bytes"] = wr[n]["WRITE_SIZE"] * 1024.0
        out["kernels"][n] = k
    json.dump(out, open(os.path.join(dst, "%s_depth_roofline.json" % rnd), "w"), indent=1)
    print("depth kernels:", {k: round(v["avg_us"], 1) for k, v in out["kernels"].items()})

# ---- C4 level-0 kernel: where the waves wait (tools/pmc_c4.sh: SQ / TCP / TCC passes)
f = os.path.join(src, "pmc_c4", "summary.json")
if os.path.exists(f):
    d = json.load(open(f))
    d = {"round": rnd, "source": "tools/pmc_c4.sh: rocprofv3 --kernel-trace --pmc <one set per pass> -- python3 tools/profile_kernel.py --width 1280 --height 960 "
         "--levels 5 --dense --batch 16 --arith fast (gn_fca_fused at level 0 over 64 alignments, averages per launch; SQ_* in quad-cycles summed over waves)", "counters": d}
    json.dump(d, open(os.path.join(dst, "%s_c4_wait_counters.json" % rnd), "w"), indent=1)
