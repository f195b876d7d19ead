#!/bin/bash
# The driver's 20-step window and the steady state against the launch-group size (cfg.coalesce) and the batches in flight.
# (groups of more than four batches need a build with ellc_ctx::MAX_COALESCE raised and bench.py's clamp with it: NOTEBOOK 5.6)
# usage: tools/sweep_groups.sh OUT "c:inflight c:inflight ..."   (one box; every point twice, interleaved)
out=${1:-gpurun_out/sweep_groups.txt}; pts=${2:-"4:16 5:20 6:24 7:21 7:28 8:24 8:32"}
mkdir -p $(dirname $out); : > $out
for rep in 1 2; do
  for p in $pts; do
    c=${p%%:*}; f=${p##*:}
    for st in 20 200; do
      python3 bench.py --no-extras --no-cpu-baseline --blocks 0 --sustained 0 --coalesce $c --inflight $f --steps $st --warmup 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('coalesce $c inflight $f steps $st rep $rep ms_per_step %.4f value %.3f M' % (d['ms_per_step'], d['value'] / 1e6))
" >> $out
    done
  done
done
cat $out
