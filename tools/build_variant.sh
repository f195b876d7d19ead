#!/bin/bash
# Build the library as of a git revision into build/libellc_hip_<name>.so (for A/B timing against the working tree).
# usage: tools/build_variant.sh <git-rev> <name>
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" egomotion_with_local_loop_closures_amd/csrc include | tar -x -C "$TMP"
cd "$TMP/egomotion_with_local_loop_closures_amd/csrc"
COMM=""
if [ -f ellc_comm.cpp ]; then   # the multi-GPU layer (round 2 on): part of the library, bound by _lib.py
  /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -w -DELLC_WITH_RCCL -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -x c++ -c ellc_comm.cpp -o ellc_comm_rccl.o
  COMM="ellc_comm_rccl.o"
fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w -DELLC_DIAG_ABI -mllvm -amdgpu-kernarg-preload-count=16 -shared -o "${VARDIR:-$ROOT/build}/libellc_hip_$NAME.so" $COMM ellc_hip.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf "$TMP"
ls -la "${VARDIR:-$ROOT/build}/libellc_hip_$NAME.so"
