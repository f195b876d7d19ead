#!/bin/bash
# Counters of the compaction kernels (prep_count / prep_scatter) in the pipelined bench workload.
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r03/pmc_prep; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $ROOT/bench.py --steps 8 --warmup 2 --trace-only --inflight 4 --coalesce 4 > $OUT/$1.log 2>&1; echo "$1 rc=$?"; }
pmc sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
pmc tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
python3 - $OUT <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "prep_scatter" if "prep_scatter" in n else ("prep_count" if "prep_count" in n else None)
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in agg.items():
            print(os.path.basename(d.rstrip("/")), k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
