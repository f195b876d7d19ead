#!/bin/bash
# Device assembly of the library ($OUT/ellc.s), the fused Gauss-Newton kernels cut out of it, their register / scratch use
# and the instruction count of the tolerance-mode pixel loop (one step = one pixel). usage: tools/asm_fused.sh [outdir]
set -e
OUT=${1:-/tmp/asm}
TOOLS=$(cd "$(dirname "$0")" && pwd)
mkdir -p $OUT
cd "$(dirname "$0")/../egomotion_with_local_loop_closures_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w -S --cuda-device-only -mllvm -amdgpu-kernarg-preload-count=16 -o $OUT/ellc.s ellc_hip.hip
cut_kernel() { awk -v n="$1" 'index($0, n ":") == 1 {f = 1} f {print} /^\.Lfunc_end/ {if (f) exit}' $OUT/ellc.s > "$2"; }
cut_kernel _ZN4ellc12gn_fca_fusedILb0ELb1ELb1ELi0EEEvPKNS_10AlignStateEPKfiiiNS_9FusedArgsE $OUT/fused_fast.s
cut_kernel _ZN4ellc12gn_fca_fusedILb1ELb1ELb0ELin1EEEvPKNS_10AlignStateEPKfiiiNS_9FusedArgsE $OUT/fused_exact.s
# registers, scratch and occupancy of every Gauss-Newton kernel
awk '/^_ZN4ellc[0-9]+gn_[a-z_]+I.*:/ {name = $1} /^; NumVgprs:/ {if (name) v = $3} /^; ScratchSize:/ {if (name) s = $3}
     /^; Occupancy:/ {if (name) {print name, "vgpr", v, "scratch", s, "occupancy", $3; name = ""}}' $OUT/ellc.s
# tolerance mode: instructions of one pixel step on the interior path = the pixel loop's body (two steps, unrolled) without the basic
# blocks of the general (per-tap) path, halved; with the weighted issue cost of tools/isa_cost.py
python3 - $OUT/fused_fast.s "$TOOLS" <<'PY'
import re, sys
sys.path.insert(0, sys.argv[2])
import isa_cost
lines = open(sys.argv[1]).read().splitlines()
L = max(i for i, l in enumerate(lines) if "Inner Loop Header" in l)
label = lines[L].split(":")[0]
# the loop ends at the last branch back to its header
E = max(i for i, l in enumerate(lines) if re.search(r"s_cbranch_\w+\s+" + re.escape(label) + r"\b|s_branch\s+" + re.escape(label) + r"\b", l))
body = lines[L:E + 1]
blocks, cur = [], []
for l in body:
    if (l.startswith(".LBB") or l.startswith("; %bb.")) and cur:
        blocks.append(cur); cur = []
    cur.append(l)
blocks.append(cur)
gen = [i for i, b in enumerate(blocks) if any(("global_load_ubyte" in l) or ("v_cmp_o_f32" in l) or ("v_cmp_u_f32" in l) for l in b)]
# the general path of each of the two steps is a contiguous run of blocks: from a step's first general block to its last
drop = set()
if gen:
    runs, start, prev = [], gen[0], gen[0]
    for i in gen[1:]:
        if i - prev > 6: runs.append((start, prev)); start = i
        prev = i
    runs.append((start, prev))
    for a, b in runs: drop.update(range(a, b + 1))
keep = [l for i, b in enumerate(blocks) if i not in drop for l in b]
tot, cyc, slow = isa_cost.cost(keep)
n = tot["fast"] + tot["slow"] + tot["trans"] + tot["cnd32"]
print("fast pixel step (interior path, per pixel): %.1f VALU instructions (fast class %.1f, slow %.1f, transcendental %.1f), %.1f others; modelled issue %.0f cycles per wave-step" %
      (n / 2, tot["fast"] / 2, (tot["slow"] + tot["cnd32"]) / 2, tot["trans"] / 2, tot["other"] / 2, cyc / 2))
PY
