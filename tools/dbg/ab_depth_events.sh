for v in "$@"; do
  if [ "$v" = tree ]; then unset ELLC_LIB_PATH; else export ELLC_LIB_PATH=$PWD/build/libellc_hip_$v.so; fi
  echo "== $v"; timeout -k 10 200 python3 tools/dbg/depth_events.py 2>&1 | tail -5
done
