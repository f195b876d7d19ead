#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: writes rocprofv3 output under gpurun_out/profiles_rNN/, which
# tools/summarize_profiles.py rNN condenses into profiles/. Counters are collected in their own passes (never together with
# trace domains other than --kernel-trace); the program itself follows `--` (no env / shell hop).
set -o pipefail
R=${1:-r03}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PK=$ROOT/tools/profile_kernel.py
trace() { rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -- python3 "${@:2}" > $OUT/$1.log 2>&1; echo "$1 rc=$?"; }
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 "${@:3}" > $OUT/$1.log 2>&1; echo "$1 rc=$?"; }
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
# the timed workload of bench.py (what the launches per batch look like) and C4 at 16 per GPU
trace bench_trace $ROOT/bench.py --steps 10 --warmup 3 --trace-only
trace c4_bench_trace $ROOT/bench.py --steps 6 --warmup 2 --width 1280 --height 960 --levels 5 --dense --batch 16 --trace-only
# the dominant kernel, both arithmetic modes, on the grid bench.py runs
for A in fast exact; do
  trace kernel_trace_$A $PK --arith $A
  pmc pmc_fetch_$A FETCH_SIZE $PK --arith $A
  pmc pmc_write_$A WRITE_SIZE $PK --arith $A
  pmc pmc_sq_$A "$SQ" $PK --arith $A
done
# C4: level 0 of 1280x960 dense, 16 alignments
C4="--width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10"
for A in fast exact; do
  trace c4_kernel_trace_$A $PK $C4 --arith $A
  pmc c4_pmc_fetch_$A FETCH_SIZE $PK $C4 --arith $A
  pmc c4_pmc_write_$A WRITE_SIZE $PK $C4 --arith $A
done
# depth-map kernels
trace depth_trace $ROOT/tools/bench_depth.py
pmc depth_pmc_fetch FETCH_SIZE $ROOT/tools/bench_depth.py
pmc depth_pmc_write WRITE_SIZE $ROOT/tools/bench_depth.py
# C4 level-0 kernel: where the waves wait
$ROOT/tools/pmc_c4.sh profiles_$R/pmc_c4 > $OUT/pmc_c4.log 2>&1; echo "pmc_c4 rc=$?"
ls $OUT | head -60
