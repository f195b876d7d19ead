#!/bin/bash
# Interleaved A/B of library builds on one box: bench.py (contract fields only; 200 warm-up steps put the HIP runtime's one-time pool growth, a 37 ms pause near the 120th graph launch when torch's runtime is loaded, before the timed 40) under each ELLC_LIB_PATH,
# several rounds, so that box-to-box and warm-up differences cancel. usage: tools/ab_libs.sh OUT ROUNDS lib1.so lib2.so ... [-- bench flags]
OUT=$1; ROUNDS=$2; shift 2
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p "$OUT"
for r in $(seq 1 "$ROUNDS"); do
  for lib in "${LIBS[@]}"; do
    name=$(basename "$lib" .so)
    python bench.py --lib $lib --no-extras --no-cpu-baseline --steps 40 --warmup 200 "$@" > "$OUT/${name}_r$r.json" 2>> "$OUT/err.log" || exit 1
  done
done
python - "$OUT" <<'PY'
import glob, json, os, statistics, sys
out = sys.argv[1]
by = {}
for f in sorted(glob.glob(os.path.join(out, "*_r*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    name = os.path.basename(f).rsplit("_r", 1)[0]
    by.setdefault(name, []).append((d["ms_per_step"], d["roofline"]["frac"]))
for name, v in by.items():
    print(f"{name}: ms/step median over runs {statistics.median(x[0] for x in v):.4f} (runs {[round(x[0], 4) for x in v]}), roofline frac {statistics.median(x[1] for x in v):.3f}")
PY
