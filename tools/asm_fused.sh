#!/bin/bash
# Device assembly of the library ($OUT/ellc.s), the fused Gauss-Newton kernels cut out of it, their register / scratch use
# and the instruction count of the tolerance-mode pixel loop (one step = one pixel). usage: tools/asm_fused.sh [outdir]
set -e
OUT=${1:-/tmp/asm}
mkdir -p $OUT
cd "$(dirname "$0")/../egomotion_with_local_loop_closures_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w -S --cuda-device-only -mllvm -amdgpu-kernarg-preload-count=16 -o $OUT/ellc.s ellc_hip.hip
cut_kernel() { awk -v n="$1" 'index($0, n ":") == 1 {f = 1} f {print} /^\.Lfunc_end/ {if (f) exit}' $OUT/ellc.s > "$2"; }
cut_kernel _ZN4ellc12gn_fca_fusedILb0ELb1ELb1ELi0EEEvPKNS_10AlignStateEPKfiiiNS_9FusedArgsE $OUT/fused_fast.s
cut_kernel _ZN4ellc12gn_fca_fusedILb1ELb1ELb0ELin1EEEvPKNS_10AlignStateEPKfiiiNS_9FusedArgsE $OUT/fused_exact.s
# registers, scratch and occupancy of every Gauss-Newton kernel
awk '/^_ZN4ellc[0-9]+gn_[a-z_]+I.*:/ {name = $1} /^; NumVgprs:/ {if (name) v = $3} /^; ScratchSize:/ {if (name) s = $3}
     /^; Occupancy:/ {if (name) {print name, "vgpr", v, "scratch", s, "occupancy", $3; name = ""}}' $OUT/ellc.s
# tolerance mode: instructions of one pixel step on the interior path = loop head up to the interior branch + interior taps + tail
python3 - $OUT/fused_fast.s <<'PY'
import re, sys
lines = open(sys.argv[1]).read().splitlines()
L = max(i for i, l in enumerate(lines) if "Inner Loop Header" in l)
is_instr = lambda l: re.match(r"\s+(v_|s_|global_|ds_|scratch_)", l) is not None
valu = total = 0
i = L
while "s_cbranch_vccz" not in lines[i]:      # head: record decode, warp, interior test
    total += is_instr(lines[i]); valu += lines[i].lstrip().startswith("v_"); i += 1
total += 1
target = lines[i].split()[1] + ":"
i = next(k for k in range(i, len(lines)) if lines[k].startswith(target))
seen_fma = False
while True:                                   # interior taps, Jacobian, weight, accumulation, loop control
    l = lines[i]
    total += is_instr(l); valu += l.lstrip().startswith("v_")
    seen_fma = seen_fma or "v_pk_fma_f32" in l
    if seen_fma and "s_cbranch_execz" in l:
        break
    i += 1
print("fast pixel step (interior path): %d VALU instructions, %d in all" % (valu, total))
PY
