# the tracked-frame loop (tools/bench_track.py) against the block counts per level (diagnostic build: ELLC_NBLK). GPU box, repo root.
export ELLC_LIB_PATH=$PWD/${1:-build/libellc_hip_envdiag.so}
for rep in 1 2; do
for N in "" "256,64,16,4" "128,32,8,2" "64,16,4,1" "32,8,2,1" "128,64,16,4" "64,32,16,4" "16,8,4,1"; do
  echo -n "track NBLK='$N': "; ELLC_NBLK=$N python3 tools/bench_track.py 600 fast 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/frame %.4f align %.4f' % (d['ms_per_frame'], d['host_ms']['align']))"
done
done
