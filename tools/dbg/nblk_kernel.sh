# level-0 kernel time against the block count per alignment (diagnostic build: ELLC_NBLK). GPU box, repo root.
# usage: tools/dbg/nblk_kernel.sh [diag lib] ; default build/libellc_hip_envdiag.so
export ELLC_LIB_PATH=$PWD/${1:-build/libellc_hip_envdiag.so}
for N in ${C4_NBLK:-8 16 20 24 32}; do
  echo -n "c4 nblk $N: "; ELLC_NBLK=$N python3 tools/profile_kernel.py --arith fast --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f us frac %.3f' % (1e3*d['avg_ms'], d['achieved_GBps']/8000))"
done
for N in ${K640_NBLK:-4 8 10 12 16}; do
  echo -n "640 nblk $N: "; ELLC_NBLK=$N python3 tools/profile_kernel.py --arith fast 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f us frac %.3f' % (1e3*d['avg_ms'], d['achieved_GBps']/8000))"
done
