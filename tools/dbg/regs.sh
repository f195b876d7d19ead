#!/bin/bash
# registers / scratch / LDS / occupancy of the Gauss-Newton and compaction kernels (cross-compiled; no GPU needed)
cd "$(dirname "$0")/../../egomotion_with_local_loop_closures_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w -S --cuda-device-only -mllvm -amdgpu-kernarg-preload-count=16 $EXTRA -o /tmp/ellc.s ellc_hip.hip || exit 1
awk '/^_ZN4ellc[0-9]+(gn_fca_fused|gn_ica_fused|gn_fca_adaptive|prep_build|gn_fca_dense)I.*:/ {name = $1} /^; NumVgprs:/ {if (name) v = $3} /^; ScratchSize:/ {if (name) s = $3} /^; LDSByteSize:/ {if (name) l = $3}
     /^; Occupancy:/ {if (name) {print name, "vgpr", v, "scratch", s, "lds", l, "occupancy", $3; name = ""}}' /tmp/ellc.s | sed 's/EEvPK.*FusedArgsE//'
