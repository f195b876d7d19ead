#!/bin/bash
# Block counts per alignment of the fused launches over a whole launch group (4 x 32 alignments), diag build (ELLC_NBLK=l0,l1,l2,l3):
# the level-0 launch alone (tools/profile_kernel.py) and the batch pipeline (bench.py, 40 steps after 200), interleaved.
# usage: tools/sweep_nblk_group.sh OUT "8,4,4,2 16,4,4,2 ..."
out=${1:-gpurun_out/sweep_nblk_group.txt}; pts=${2:-"8,4,4,2 12,4,4,2 16,4,4,2 8,8,4,2 16,8,4,2"}
mkdir -p $(dirname $out); : > $out
export ELLC_LIB_PATH=$PWD/egomotion_with_local_loop_closures_amd/csrc/variants/libellc_hip_envdiag.so
for rep in 1 2; do
  for p in $pts; do
    k=$(ELLC_NBLK=$p python3 tools/profile_kernel.py --calib-mb 16 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (1e3*d['avg_ms']))")
    b=$(ELLC_NBLK=$p python3 bench.py --lib $ELLC_LIB_PATH --no-extras --no-cpu-baseline --blocks 0 --sustained 0 --steps 40 --warmup 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
    echo "nblk $p rep $rep: level-0 launch $k us, pipeline $b ms per step" | tee -a $out
  done
done
