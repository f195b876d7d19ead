# the batch pipeline against the block counts per level (diagnostic build: ELLC_NBLK=l0,l1,..). GPU box, repo root.
# usage: tools/dbg/nblk_bench.sh [diag lib]
export ELLC_LIB_PATH=$PWD/${1:-build/libellc_hip_envdiag.so}
for rep in 1 2; do
for N in "" "8,8" "8,4,4,2" "8,4,2,2" "8,4,2,1" "8,6,4,2" "8,4,4,4" "10,5,4,2" "8,2,2,2" "6,4,2,2"; do
  echo -n "bench NBLK='$N': "; ELLC_NBLK=$N python3 bench.py --lib $ELLC_LIB_PATH --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.3fM ms/step %.4f k0 %.1f us' % (d['value']/1e6, d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))"
done
done
