#!/bin/bash
# SQ counters of the depth kernels for one or more library builds (bench_depth.py under rocprofv3 --pmc; counters only).
# usage (on the GPU box): bash tools/pmc_depth.sh name1 name2 ...   ("tree" = the in-tree library)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = tree ]; then unset ELLC_LIB_PATH; else export ELLC_LIB_PATH=$R/build/libellc_hip_$v.so; fi
  rm -rf /tmp/pmcd_$v
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmcd_$v -o d -- python3 $R/tools/bench_depth.py > $R/gpurun_out/pmcd_$v.log 2>&1
  f=$(find /tmp/pmcd_$v -name "*counter_collection.csv" | head -1)
  echo "== $v"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "dm_" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in acc:
    print(k, "launches", n[k], " ".join("%s=%.0f" % (c, v / max(1, n[k])) for c, v in sorted(acc[k].items())))
PY
  else echo "no counters written"; fi
done
