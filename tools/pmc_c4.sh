#!/bin/bash
# C4 level-0 kernel (1280x960 dense, 4 x 16 alignments per launch): where the waves wait. Counter passes only (--kernel-trace + --pmc).
# usage (GPU box, repo root): tools/pmc_c4.sh OUTNAME [extra profile_kernel args]
set -o pipefail
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/${1:-pmc_c4}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PK=$ROOT/tools/profile_kernel.py
C4="--width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10 --arith fast"
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $PK $C4 "${@:3}" > $OUT/$1.log 2>&1; echo "$1 rc=$?"; }
pmc sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "$@"
[ -n "$PMC_ALL" ] && pmc sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "$@"
[ -n "$PMC_ALL" ] && pmc sq3 "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "$@"
pmc tcp1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "$@"
[ -n "$PMC_ALL" ] && pmc tcp2 "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "$@"
# (a pass with TA_* counters hung rocprofv3 on this pool in r03: not collected)
pmc tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "$@"
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "*/"))):
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        dur = []
        for r in csv.DictReader(open(f)):
            if "gn_fca" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in agg.items():
            res[k] = sum(v) / len(v)
        if dur:
            res["duration_us_" + os.path.basename(d.rstrip("/"))] = sum(dur) / len(dur)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
