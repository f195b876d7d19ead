#!/bin/bash
# A/B of the depth kernels across library builds: rocprofv3 kernel stats of tools/bench_depth.py per library.
# usage (on the GPU box): bash tools/ab_depth.sh name1 name2 ...   (build/libellc_hip_<name>.so; "tree" = the in-tree library)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = tree ]; then unset ELLC_LIB_PATH; else if [ -f "$R/build/libellc_hip_$v.so" ]; then export ELLC_LIB_PATH=$R/build/libellc_hip_$v.so; else export ELLC_LIB_PATH=$R/egomotion_with_local_loop_closures_amd/csrc/variants/libellc_hip_$v.so; fi; fi
  rm -rf /tmp/abd_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abd_$v -o d -- python3 $R/tools/bench_depth.py > $R/gpurun_out/abd_$v.log 2>&1
  f=$(find /tmp/abd_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  if [ -n "$f" ]; then grep -E "dm_observe|dm_regularize|dm_fill" "$f" | cut -d, -f1-4,6,7; else echo "no kernel stats written"; fi
done
