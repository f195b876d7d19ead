#!/bin/bash
# Device assembly of the library, and the fused Gauss-Newton kernel cut out of it (for instruction counting).
set -e
OUT=${1:-/tmp/asm}
mkdir -p $OUT
cd /root/repo/egomotion_with_local_loop_closures_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w -S --cuda-device-only -mllvm -amdgpu-kernarg-preload-count=16 -o $OUT/ellc.s ellc_hip.hip
awk '/^_ZN4ellc12gn_fca_fusedILb1ELb1EEEvPKNS_10AlignStateEPKfiiiNS_9FusedArgsE:/{f=1} f{print} /s_endpgm/{if(f){exit}}' $OUT/ellc.s > $OUT/fused.s
grep -E "^\s+(v_|s_|global_|ds_|flat_|buffer_)" $OUT/fused.s | wc -l
grep -E "vgpr_count|sgpr_count|scratch|Occupancy|NumVgprs|ScratchSize" $OUT/ellc.s | awk '/gn_fca_fusedILb1ELi1/{f=1} f' | head -0
