#!/usr/bin/env python3
"""Weighted VALU issue cost of a stretch of gfx950 assembly (r04).

tools/micro/valu_rates.hip measured, with >= 4 waves per SIMD (cycles per wave-instruction per SIMD, gpurun_out/r04/valu_rates.txt):
  fast  ~2.4 : v_add/sub/mul/fma/fmac/fmaak/fmamk_f32, v_mov_b32, v_and/or/xor_b32, v_add/sub_u32, v_lshrrev_b32, v_ashrrev_i32
               -- with VGPR, inline-constant or literal operands only, no SDWA/DPP, and (fma) no register named twice among the sources
  slow  ~4.4 : everything else (cvt, floor/fract, min/max/med3, cmp, cndmask_e64, 3-operand integer ops, v_lshlrev_b32, pk_*, f64,
               SDWA, DPP, and ANY instruction with an SGPR operand)
  trans ~8.5 : v_rcp/rsq/sqrt/exp/log/sin/cos_f32
  (v_cndmask_b32_e32 back to back reads vcc at ~22.8 cycles each in the microbenchmark; paired with the v_cmp that feeds it, or spaced by
   other instructions, it costs what a slow instruction costs: counted as slow)
usage: isa_cost.py file.s [first_line last_line]   (1-based, inclusive; default: whole file)
"""
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32"}
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
C_FAST, C_SLOW, C_TRANS, C_CND32 = 2.4, 4.4, 8.5, 4.4


def classify(line):
    l = line.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        return None
    l = l.split(";")[0].strip()
    m = re.match(r"([a-z_0-9]+)\s*(.*)", l)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    if not op.startswith("v_"):
        return ("other", 0.0, op)
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith(TRANS):
        return ("trans", C_TRANS, op)
    if base == "v_cndmask_b32" and not op.endswith("_e64") and "sdwa" not in op and "dpp" not in op:
        return ("cnd32", C_CND32, op)
    ops = [o.strip() for o in rest.split(",")]
    has_sgpr = any(re.match(r"^-?\|?(s\d+|s\[|vcc|exec|m0|scc|ttmp)", o) for o in ops[1:])
    if base in FAST and not op.endswith(("_sdwa", "_dpp")) and "sdwa" not in rest and "row_" not in rest and "quad_perm" not in rest and not has_sgpr:
        if base in ("v_fma_f32",):
            srcs = [re.sub(r"[-|]", "", o) for o in ops[1:4]]
            regs = [s for s in srcs if s.startswith("v")]
            if len(set(regs)) != len(regs):
                return ("slow", C_SLOW, op)
        return ("fast", C_FAST, op)
    return ("slow", C_SLOW, op)


def cost(lines):
    tot = {"fast": 0, "slow": 0, "trans": 0, "cnd32": 0, "other": 0}
    cyc = 0.0
    slow_ops = {}
    for ln in lines:
        c = classify(ln)
        if c is None:
            continue
        tot[c[0]] += 1
        cyc += c[1]
        if c[0] in ("slow", "cnd32", "trans"):
            slow_ops[c[2]] = slow_ops.get(c[2], 0) + 1
    return tot, cyc, slow_ops


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().splitlines()
    if len(sys.argv) >= 4:
        lines = lines[int(sys.argv[2]) - 1:int(sys.argv[3])]
    tot, cyc, slow_ops = cost(lines)
    n = tot["fast"] + tot["slow"] + tot["trans"] + tot["cnd32"]
    print("VALU %d (fast %d, slow %d, trans %d, cndmask_e32 %d), other %d; weighted issue cost %.0f cycles per wave-step" %
          (n, tot["fast"], tot["slow"], tot["trans"], tot["cnd32"], tot["other"], cyc))
    print("not-fast opcodes:", ", ".join("%s x%d" % kv for kv in sorted(slow_ops.items(), key=lambda kv: -kv[1])))
