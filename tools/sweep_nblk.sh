#!/bin/bash
# level-0 block-count sweep of the production (fused) kernel, B = 32
for nb in 8 12 16 20 24 28 30 32 36 40 48; do
  echo -n "nblk0=$nb  "; ELLC_NBLK=$nb python3 tools/profile_kernel.py --calib-mb 16 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg_ms %.4f  achieved %.0f GB/s' % (d['avg_ms'], d['achieved_GBps']))"
done
