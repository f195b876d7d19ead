#!/bin/bash
# r04 verdict item 3 (LDS tap tile for the semi-dense list kernel): the level-0 launch of the bench workload (640x480 semi-dense,
# 4 x 32 alignments per launch) as shipped against the -DELLC_X_LDSTAPS variant (the four tap rows read as two aligned LDS dwords +
# v_alignbit each from a window that is "already there": garbage values, no staging — the ceiling of what a window can give), with
# the vector-cache and LDS counters of both. Counter passes only (--kernel-trace + --pmc).
# usage (GPU box, repo root): tools/pmc_ldstaps.sh OUTNAME     (needs csrc/variants/libellc_hip_ldstaps.so: make variant NAME=ldstaps
#                                                               DEFS=-DELLC_X_LDSTAPS VARDIR=variants)
set -o pipefail
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/${1:-pmc_ldstaps}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PK=$ROOT/tools/profile_kernel.py
ARGS="--reps 20 --arith fast"
VAR=$ROOT/egomotion_with_local_loop_closures_amd/csrc/variants/libellc_hip_ldstaps.so
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $PK $ARGS > $OUT/$1.log 2>&1; echo "$1 rc=$?"; }
for lib in head ldstaps; do
  if [ $lib = head ]; then unset ELLC_LIB_PATH; else export ELLC_LIB_PATH=$VAR; fi
  for i in 1 2 3; do python3 $PK $ARGS | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib live avg_us %.2f' % (d['avg_ms'] * 1e3))" | tee -a $OUT/live.txt; done
  pmc ${lib}_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
  pmc ${lib}_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM"
  pmc ${lib}_tcp1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
res = {"head": {}, "ldstaps": {}}
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    name = os.path.basename(d.rstrip("/")); lib = name.split("_")[0]
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list); dur = []
        for r in csv.DictReader(open(f)):
            if "gn_fca" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in agg.items():
            res[lib][k] = sum(v) / len(v)
        if dur:
            res[lib]["duration_us_" + name] = sum(dur) / len(dur)
res["live"] = open(os.path.join(out, "live.txt")).read().split("\n")
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
