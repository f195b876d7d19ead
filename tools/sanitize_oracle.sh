#!/bin/bash
# CPU-only check: the oracle built with AddressSanitizer + UBSan runs its whole CPU suite without a report.
# (GPU sanitizers are not available on the MI355X pool; the HIP library has no CPU build to sanitize.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/oracle"
cp libellc_oracle.so /tmp/libellc_oracle_keep.so 2>/dev/null || true
trap 'cp /tmp/libellc_oracle_keep.so "$ROOT/oracle/libellc_oracle.so" 2>/dev/null || make -s -C "$ROOT/oracle"' EXIT
g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o libellc_oracle.so ellc_oracle_core.cpp ellc_oracle_gn.cpp ellc_oracle_depth.cpp ellc_oracle_capi.cpp
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests -x -q -m "not gpu" -k oracle -s 2>&1 | grep -iE "runtime error|AddressSanitizer|passed|failed"
