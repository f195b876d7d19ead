#!/bin/bash
# bench.py under a list of environment settings (one per line on stdin, "-" = none); prints value and ms/step for each.
# usage (on the GPU box): bash tools/sweep_env.sh [bench args] < settings.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
while read -r line; do
  [ -z "$line" ] && continue
  if [ "$line" = "-" ]; then envs=""; else envs="$line"; fi
  out=$(env $envs python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1)
  echo "$line :: $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('%.0f it/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" "$out")"
done
