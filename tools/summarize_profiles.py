#!/usr/bin/env python3
"""Condenses gpurun_out/profiles_<round>/ (rocprofv3 output) into profiles/<round>_*: the kernel_stats CSVs as they are
plus one JSON with per-kernel counter averages and the calibrated HBM traffic of the dominant kernel."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "profiles_" + rnd)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
for name in ("bench_trace", "kernel_trace"):
    f = sorted(glob.glob(os.path.join(src, name, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if f:
        shutil.copy(f[-1], os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, name)))
summary = {"round": rnd, "source": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python3 tools/profile_kernel.py", "kernels": {}}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = sorted(glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "gn_fca_" in r["Kernel_Name"] or "calib_read" in r["Kernel_Name"]:
            k = "gn_fca_level0" if "gn_fca" in r["Kernel_Name"] else "calib_read_f32"
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["duration_us_" + name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        summary["kernels"].setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
        summary["kernels"][k]["dispatches_" + name] = max(len(x) for x in v.values())
log = os.path.join(src, "kernel_trace.log")
run = None
for line in open(log):
    if line.startswith("{"):
        run = json.loads(line)
summary["profile_kernel_run"] = run
k = summary["kernels"].get("gn_fca_level0", {})
c = summary["kernels"].get("calib_read_f32", {})
if run and "FETCH_SIZE" in k and "FETCH_SIZE" in c:
    factor = run["calib_bytes_per_launch"] / (c["FETCH_SIZE"] * 1024.0)   # known bytes / reported bytes (gfx950: 2.0)
    fetch = k["FETCH_SIZE"] * 1024.0 * factor
    write = k.get("WRITE_SIZE", 0.0) * 1024.0
    summary["hbm_traffic"] = {"fetch_correction_factor": factor, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                              "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": run["algorithmic_bytes"],
                              "traffic_over_algorithmic": (fetch + write) / run["algorithmic_bytes"]}
json.dump(summary, open(os.path.join(dst, "%s_pmc_summary.json" % rnd), "w"), indent=1)
print(json.dumps(summary.get("hbm_traffic"), indent=1))
