#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: writes rocprofv3 summaries under gpurun_out/profiles_rNN/.
# Counters are collected in their own passes (never together with trace domains other than --kernel-trace).
set -o pipefail
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_trace -- python3 $GRAFT_REPO_ROOT/tools/profile_kernel.py > $OUT/kernel_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/profile_kernel.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/profile_kernel.py > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/tools/profile_kernel.py > $OUT/pmc_sq.log 2>&1
ls -R $OUT | head -40
