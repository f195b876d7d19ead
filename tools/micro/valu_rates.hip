// Issue cost of the vector instructions the pixel passes are made of, on one SIMD with 1 / 2 / 4 waves resident (r04).
// Every kernel runs REPS rounds of 32 independent instances of ONE instruction (eight register sets, no dependences inside a
// round closer than eight instructions) and stamps s_memtime around the loop: cycles per wave-instruction per SIMD =
// (cycles * 100 MHz-clock ratio) ... reported as ns per wave-instruction per SIMD from hipEvents over the whole grid, and as
// "relative to v_fma_f32". A grid of 256 CUs x 4 SIMDs x W waves: blocks of 64 * 4 * W threads, one block per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o build/valu_rates tools/micro/valu_rates.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REPS 16384

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R32(X) R8(X) R8(X) R8(X) R8(X)

// each OP(k) must be one instruction on register set k: v[8+2k : 9+2k] destination / accumulator, v[40+2k:41+2k], v[60+2k:61+2k] sources
#define DEF_KERNEL(NAME, ...)                                                                                          \
  __global__ __launch_bounds__(1024) void NAME(float* out, float seed) {                                              \
    float a0 = seed + threadIdx.x, a1 = a0 * 1.0001f, a2 = a0 * 0.5f, a3 = a1 + 2.0f;                                  \
    unsigned t_shader, t_real;                                                                                        \
    asm volatile(                                                                                                     \
        "v_mov_b32 v8, %0\n v_mov_b32 v9, %1\n v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n"                               \
        "v_mov_b32 v12, %0\n v_mov_b32 v13, %1\n v_mov_b32 v14, %2\n v_mov_b32 v15, %3\n"                             \
        "v_mov_b32 v16, %0\n v_mov_b32 v17, %1\n v_mov_b32 v18, %2\n v_mov_b32 v19, %3\n"                             \
        "v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, %3\n"                             \
        "v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n v_mov_b32 v42, %2\n v_mov_b32 v43, %3\n"                             \
        "v_mov_b32 v44, %0\n v_mov_b32 v45, %1\n v_mov_b32 v46, %2\n v_mov_b32 v47, %3\n"                             \
        "v_mov_b32 v48, %0\n v_mov_b32 v49, %1\n v_mov_b32 v50, %2\n v_mov_b32 v51, %3\n"                             \
        "v_mov_b32 v52, %0\n v_mov_b32 v53, %1\n v_mov_b32 v54, %2\n v_mov_b32 v55, %3\n"                             \
        "v_mov_b32 v60, 0x3f7fff00\n v_mov_b32 v61, 0x3f7fff00\n v_mov_b32 v62, 0x3f7fff00\n v_mov_b32 v63, 0x3f7fff00\n" \
        "v_mov_b32 v64, 0x3f7fff00\n v_mov_b32 v65, 0x3f7fff00\n v_mov_b32 v66, 0x3f7fff00\n v_mov_b32 v67, 0x3f7fff00\n" \
        "v_mov_b32 v68, 0x3f7fff00\n v_mov_b32 v69, 0x3f7fff00\n v_mov_b32 v70, 0x3f7fff00\n v_mov_b32 v71, 0x3f7fff00\n" \
        "v_mov_b32 v72, 0x3f7fff00\n v_mov_b32 v73, 0x3f7fff00\n v_mov_b32 v74, 0x3f7fff00\n v_mov_b32 v75, 0x3f7fff00\n" \
        "v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n" \
        "s_movk_i32 s20, %6\n s_mov_b32 s21, 0x3f7fff00\n s_mov_b64 s[30:31], 0x55\n s_mov_b64 vcc, 0x33\n s_memtime s[22:23]\n s_memrealtime s[24:25]\n s_waitcnt lgkmcnt(0)\n"                      \
        "1:\n" __VA_ARGS__                                                                                            \
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                           \
        "s_memtime s[26:27]\n s_memrealtime s[28:29]\n s_waitcnt lgkmcnt(0)\n"                                        \
        "s_sub_u32 s22, s26, s22\n s_sub_u32 s24, s28, s24\n v_mov_b32 %4, s22\n v_mov_b32 %5, s24\n"                 \
        "v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11\n v_add_f32 %0, %0, v12\n"              \
        "v_add_f32 %0, %0, v13\n v_add_f32 %0, %0, v14\n v_add_f32 %0, %0, v15\n v_add_f32 %0, %0, v16\n"             \
        "v_add_f32 %0, %0, v17\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v19\n v_add_f32 %0, %0, v20\n"             \
        "v_add_f32 %0, %0, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n"                                     \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=v"(t_shader), "=v"(t_real)                                           \
        : "n"(REPS)                                                                                                   \
        : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22",     \
          "v23", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",    \
          "v54", "v55", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",    \
          "v73", "v74", "v75", "v76", "v77", "s20", "s21", "s30", "s31", "s32", "s33", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc", "vcc", "memory");                                                        \
    if (a0 == 12345.678f) out[threadIdx.x] = a0 + a1 + a2 + a3;                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0) { ((unsigned*)out)[1024] = t_shader; ((unsigned*)out)[1025] = t_real; } \
  }

// 32 instructions per round: 4 x (register sets 0..7). Macro arithmetic is not evaluated inside strings: the sets are spelled out.
#define SETS(F) F("v8", "v40", "v60", "v[8:9]", "v[40:41]", "v[60:61]") F("v10", "v42", "v62", "v[10:11]", "v[42:43]", "v[62:63]") \
  F("v12", "v44", "v64", "v[12:13]", "v[44:45]", "v[64:65]") F("v14", "v46", "v66", "v[14:15]", "v[46:47]", "v[66:67]")           \
  F("v16", "v48", "v68", "v[16:17]", "v[48:49]", "v[68:69]") F("v18", "v50", "v70", "v[18:19]", "v[50:51]", "v[70:71]")           \
  F("v20", "v52", "v72", "v[20:21]", "v[52:53]", "v[72:73]") F("v22", "v54", "v74", "v[22:23]", "v[54:55]", "v[74:75]")
#define X4(x) x x x x
#define K(NAME, F) DEF_KERNEL(NAME, X4(SETS(F)))

#define F_FMA(d, a, b, D, A, B) "v_fma_f32 " d ", " a ", " b ", v41\n"
#define F_MUL(d, a, b, D, A, B) "v_mul_f32 " d ", " a ", " b "\n"
#define F_FMAC(d, a, b, D, A, B) "v_fmac_f32 " d ", " a ", " b "\n"
#define F_PKFMA(d, a, b, D, A, B) "v_pk_fma_f32 " D ", " A ", " B ", v[76:77]\n"
#define F_PKFMAB(d, a, b, D, A, B) "v_pk_fma_f32 " D ", " A ", " B ", v[76:77] op_sel_hi:[0,1,1]\n"
#define F_PKFMAACC(d, a, b, D, A, B) "v_pk_fma_f32 " D ", " A ", " B ", " D "\n"
#define F_PKMUL(d, a, b, D, A, B) "v_pk_mul_f32 " D ", " A ", " B "\n"
#define F_PKADD(d, a, b, D, A, B) "v_pk_add_f32 " D ", " A ", " B "\n"
#define F_RCP(d, a, b, D, A, B) "v_rcp_f32 " d ", " a "\n"
#define F_RSQ(d, a, b, D, A, B) "v_rsq_f32 " d ", " a "\n"
#define F_FLOOR(d, a, b, D, A, B) "v_floor_f32 " d ", " a "\n"
#define F_CVTUB1(d, a, b, D, A, B) "v_cvt_f32_ubyte1 " d ", " a "\n"
#define F_CVTI2F(d, a, b, D, A, B) "v_cvt_f32_i32 " d ", " a "\n"
#define F_CVTF2I(d, a, b, D, A, B) "v_cvt_i32_f32 " d ", " a "\n"
#define F_MED3(d, a, b, D, A, B) "v_med3_f32 " d ", " a ", " b ", 1.0\n"
#define F_MADU24(d, a, b, D, A, B) "v_mad_u32_u24 " d ", " a ", " b ", v41\n"
#define F_MULLO(d, a, b, D, A, B) "v_mul_lo_u32 " d ", " a ", " b "\n"
#define F_ADD3(d, a, b, D, A, B) "v_add3_u32 " d ", " a ", " b ", v41\n"
#define F_BFE(d, a, b, D, A, B) "v_bfe_u32 " d ", " a ", 12, 12\n"
#define F_SUBSDWA(d, a, b, D, A, B) "v_sub_u32_sdwa " d ", " a ", " b " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
#define F_CVTSDWA(d, a, b, D, A, B) "v_cvt_f32_ubyte0_sdwa " d ", " a " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
#define F_CNDMASK(d, a, b, D, A, B) "v_cndmask_b32 " d ", " a ", " b ", vcc\n"
#define F_CMP(d, a, b, D, A, B) "v_cmp_lt_f32 vcc, " a ", " b "\n"
#define F_MOVDPP(d, a, b, D, A, B) "v_mov_b32_dpp " d ", " a " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_ADDDPP(d, a, b, D, A, B) "v_add_f32_dpp " d ", " a ", " b " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_FMAF64(d, a, b, D, A, B) "v_fma_f64 " D ", " A ", " B ", v[76:77]\n"
#define F_PKMOV(d, a, b, D, A, B) "v_pk_mov_b32 " D ", " A ", " B "\n"
#define F_PERM(d, a, b, D, A, B) "v_perm_b32 " d ", " a ", " b ", v41\n"
#define F_SAD(d, a, b, D, A, B) "v_sad_u8 " d ", " a ", " b ", v41\n"
#define F_PKSUBI16(d, a, b, D, A, B) "v_pk_sub_i16 " d ", " a ", " b "\n"
#define F_CVTF16(d, a, b, D, A, B) "v_cvt_f32_f16 " d ", " a "\n"
#define F_FMAMIX(d, a, b, D, A, B) "v_fma_mix_f32 " d ", " a ", " b ", v41 op_sel_hi:[1,0,0]\n"
#define F_LSHLADD(d, a, b, D, A, B) "v_lshl_add_u32 " d ", " a ", 4, " b "\n"
#define F_MIN(d, a, b, D, A, B) "v_min_f32 " d ", " a ", " b "\n"
#define F_ALIGNBYTE(d, a, b, D, A, B) "v_alignbyte_b32 " d ", " a ", " b ", v41\n"
#define F_SUBF32SDWA(d, a, b, D, A, B) "v_sub_f32_sdwa " d ", " a ", " b " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"

#define F_ADD(d, a, b, D, A, B) "v_add_f32 " d ", " a ", " b "\n"
#define F_SUB(d, a, b, D, A, B) "v_sub_f32 " d ", " a ", " b "\n"
#define F_MAX(d, a, b, D, A, B) "v_max_f32 " d ", " a ", " b "\n"
#define F_MOV(d, a, b, D, A, B) "v_mov_b32 " d ", " a "\n"
#define F_AND(d, a, b, D, A, B) "v_and_b32 " d ", " a ", " b "\n"
#define F_ADDU(d, a, b, D, A, B) "v_add_u32 " d ", " a ", " b "\n"
#define F_LSHR(d, a, b, D, A, B) "v_lshrrev_b32 " d ", 8, " a "\n"
#define F_FMA_SAME(d, a, b, D, A, B) "v_fma_f32 " d ", " a ", " a ", " a "\n"
#define F_FMA_NEG(d, a, b, D, A, B) "v_fma_f32 " d ", " a ", -" b ", v41\n"
#define F_FMA_SGPR(d, a, b, D, A, B) "v_fma_f32 " d ", s21, " b ", v41\n"
#define F_FMAAK(d, a, b, D, A, B) "v_fmaak_f32 " d ", " a ", " b ", 0x41800000\n"
#define F_MUL_E64(d, a, b, D, A, B) "v_mul_f32_e64 " d ", " a ", -" b "\n"
#define F_MUL_SGPR(d, a, b, D, A, B) "v_mul_f32 " d ", s21, " b "\n"
#define F_CNDMASK_S(d, a, b, D, A, B) "v_cndmask_b32_e64 " d ", " a ", " b ", s[30:31]\n"
#define F_CNDMASK_0(d, a, b, D, A, B) "v_cndmask_b32_e64 " d ", 0, " b ", s[30:31]\n"
#define F_CMP_S(d, a, b, D, A, B) "v_cmp_lt_f32_e64 s[32:33], " a ", " b "\n"
#define F_CVTUB0(d, a, b, D, A, B) "v_cvt_f32_ubyte0 " d ", " a "\n"
#define F_CVTU2F(d, a, b, D, A, B) "v_cvt_f32_u32 " d ", " a "\n"
#define F_FRACT(d, a, b, D, A, B) "v_fract_f32 " d ", " a "\n"
#define F_MAD_I24(d, a, b, D, A, B) "v_mad_i32_i24 " d ", " a ", " b ", v41\n"
#define F_MUL_U24(d, a, b, D, A, B) "v_mul_u32_u24 " d ", " a ", " b "\n"
#define F_PKMUL_B(d, a, b, D, A, B) "v_pk_mul_f32 " D ", " A ", " B " op_sel_hi:[1,0]\n"
#define F_MULF64(d, a, b, D, A, B) "v_mul_f64 " D ", " A ", " B "\n"
#define F_ADDF64(d, a, b, D, A, B) "v_add_f64 " D ", " A ", " B "\n"
#define F_FMA_C1(d, a, b, D, A, B) "v_fma_f32 " d ", " a ", " b ", 1.0\n"
#define F_FMA_LIT(d, a, b, D, A, B) "v_fma_f32 " d ", " a ", " b ", 0x41800000\n"
#define F_FMAC_IND(d, a, b, D, A, B) "v_fmac_f32 " a ", " b ", v41\n"
#define F_FMA_ACC(d, a, b, D, A, B) "v_fma_f32 " a ", " b ", v41, " a "\n"
#define F_MUL_C(d, a, b, D, A, B) "v_mul_f32 " d ", 0.5, " b "\n"
#define F_MUL_LIT(d, a, b, D, A, B) "v_mul_f32 " d ", 0x3fc00000, " b "\n"
#define F_OR(d, a, b, D, A, B) "v_or_b32 " d ", " a ", " b "\n"
#define F_XOR(d, a, b, D, A, B) "v_xor_b32 " d ", " a ", " b "\n"
#define F_LSHL(d, a, b, D, A, B) "v_lshlrev_b32 " d ", 4, " a "\n"
#define F_SUBU(d, a, b, D, A, B) "v_sub_u32 " d ", " a ", " b "\n"
#define F_CND_VCC(d, a, b, D, A, B) "v_cndmask_b32_e32 " d ", " a ", " b ", vcc\n"
#define F_CND_VCC0(d, a, b, D, A, B) "v_cndmask_b32_e32 " d ", 0, " b ", vcc\n"
#define F_CND_VCC64(d, a, b, D, A, B) "v_cndmask_b32_e64 " d ", " a ", " b ", vcc\n"
#define F_CVTFLR(d, a, b, D, A, B) "v_cvt_flr_i32_f32 " d ", " a "\n"
#define F_ADD_ABS(d, a, b, D, A, B) "v_add_f32_e64 " d ", |" a "|, " b "\n"
#define F_MUL_SELF(d, a, b, D, A, B) "v_mul_f32 " a ", " a ", " b "\n"
#define F_FMA_SQ(d, a, b, D, A, B) "v_fma_f32 " d ", " a ", " a ", " b "\n"
#define F_MUL_SQ(d, a, b, D, A, B) "v_mul_f32 " d ", " a ", " a "\n"
#define F_CMP_U32(d, a, b, D, A, B) "v_cmp_lt_u32 vcc, " a ", " b "\n"
#define F_SUB_SGPR(d, a, b, D, A, B) "v_sub_f32 " d ", s21, " b "\n"
#define F_ADD_C(d, a, b, D, A, B) "v_add_f32 " d ", 1.0, " b "\n"
#define F_MOV_S(d, a, b, D, A, B) "v_mov_b32 " d ", s21\n"
#define F_LDEXP(d, a, b, D, A, B) "v_ldexp_f32 " d ", " a ", " b "\n"
#define F_ADD_NEG(d, a, b, D, A, B) "v_sub_f32_e64 " d ", " a ", -" b "\n"
#define F_MIN3(d, a, b, D, A, B) "v_min3_f32 " d ", " a ", " b ", v41\n"
#define F_ANDOR(d, a, b, D, A, B) "v_and_or_b32 " d ", " a ", " b ", v41\n"
#define F_ADDLSHL(d, a, b, D, A, B) "v_add_lshl_u32 " d ", " a ", " b ", 2\n"
#define F_MIN_I32(d, a, b, D, A, B) "v_min_i32 " d ", " a ", " b "\n"
#define F_ADD_CO(d, a, b, D, A, B) "v_add_co_u32 " d ", vcc, " a ", " b "\n"
#define F_ASHR(d, a, b, D, A, B) "v_ashrrev_i32 " d ", 8, " a "\n"
#define F_FMA_2S(d, a, b, D, A, B) "v_fma_f32 " d ", " a ", " b ", " b "\n"
#define F_CVT_UB0_B(d, a, b, D, A, B) "v_cvt_f32_ubyte0 " d ", " a "\n v_sub_f32 " a ", " d ", " b "\n"
#define F_PAIR_VCC(d, a, b, D, A, B) "v_cmp_lt_f32 vcc, " a ", " b "\n v_cndmask_b32_e32 " d ", " a ", " b ", vcc\n"
#define F_PAIR_S(d, a, b, D, A, B) "v_cmp_lt_f32_e64 s[30:31], " a ", " b "\n v_cndmask_b32_e64 " d ", " a ", " b ", s[30:31]\n"
#define F_PAIR_VCC64(d, a, b, D, A, B) "v_cmp_lt_f32 vcc, " a ", " b "\n v_cndmask_b32_e64 " d ", " a ", " b ", vcc\n"
#define F_CND32_SPACED(d, a, b, D, A, B) "v_cndmask_b32_e32 " d ", " a ", " b ", vcc\n v_add_f32 " d ", " b ", v41\n v_add_f32 " d ", " b ", v41\n v_add_f32 " d ", " b ", v41\n"
K(k_pair_vcc, F_PAIR_VCC) K(k_pair_s, F_PAIR_S) K(k_pair_vcc64, F_PAIR_VCC64) K(k_cnd32_spaced, F_CND32_SPACED)
K(k_fma_c1, F_FMA_C1) K(k_fmac_ind, F_FMAC_IND) K(k_fma_acc, F_FMA_ACC) K(k_mul_c, F_MUL_C) K(k_mul_lit, F_MUL_LIT) K(k_or, F_OR)
K(k_xor, F_XOR) K(k_lshl, F_LSHL) K(k_subu, F_SUBU) K(k_cnd_vcc, F_CND_VCC) K(k_cnd_vcc0, F_CND_VCC0) K(k_cnd_vcc64, F_CND_VCC64) K(k_cvtflr, F_CVTFLR)
K(k_add_abs, F_ADD_ABS) K(k_mul_self, F_MUL_SELF) K(k_fma_sq, F_FMA_SQ) K(k_mul_sq, F_MUL_SQ) K(k_cmp_u32, F_CMP_U32) K(k_sub_sgpr, F_SUB_SGPR)
K(k_add_c, F_ADD_C) K(k_mov_s, F_MOV_S) K(k_ldexp, F_LDEXP) K(k_add_neg, F_ADD_NEG) K(k_min3, F_MIN3) K(k_andor, F_ANDOR) K(k_addlshl, F_ADDLSHL)
K(k_min_i32, F_MIN_I32) K(k_add_co, F_ADD_CO) K(k_ashr, F_ASHR) K(k_fma_2s, F_FMA_2S)
K(k_add, F_ADD) K(k_sub, F_SUB) K(k_max, F_MAX) K(k_mov, F_MOV) K(k_and, F_AND) K(k_addu, F_ADDU) K(k_lshr, F_LSHR) K(k_fma_same, F_FMA_SAME)
K(k_fma_neg, F_FMA_NEG) K(k_fma_sgpr, F_FMA_SGPR) K(k_fmaak, F_FMAAK) K(k_mul_e64, F_MUL_E64) K(k_mul_sgpr, F_MUL_SGPR) K(k_cndmask_s, F_CNDMASK_S)
K(k_cndmask_0, F_CNDMASK_0) K(k_cmp_s, F_CMP_S) K(k_cvt_ub0, F_CVTUB0) K(k_cvt_u2f, F_CVTU2F) K(k_fract, F_FRACT) K(k_mad_i24, F_MAD_I24)
K(k_mul_u24, F_MUL_U24) K(k_pk_mul_b, F_PKMUL_B) K(k_mul_f64, F_MULF64) K(k_add_f64, F_ADDF64)
K(k_fma, F_FMA) K(k_mul, F_MUL) K(k_fmac, F_FMAC) K(k_pk_fma, F_PKFMA) K(k_pk_fma_bcast, F_PKFMAB) K(k_pk_fma_acc, F_PKFMAACC)
K(k_pk_mul, F_PKMUL) K(k_pk_add, F_PKADD) K(k_rcp, F_RCP) K(k_rsq, F_RSQ) K(k_floor, F_FLOOR) K(k_cvt_ub1, F_CVTUB1)
K(k_cvt_i32, F_CVTI2F) K(k_cvt_f2i, F_CVTF2I) K(k_med3, F_MED3) K(k_mad_u24, F_MADU24) K(k_mul_lo, F_MULLO) K(k_add3, F_ADD3)
K(k_bfe, F_BFE) K(k_sub_sdwa, F_SUBSDWA) K(k_cvt_sdwa, F_CVTSDWA) K(k_cndmask, F_CNDMASK) K(k_cmp, F_CMP) K(k_mov_dpp, F_MOVDPP)
K(k_add_dpp, F_ADDDPP) K(k_fma_f64, F_FMAF64) K(k_pk_mov, F_PKMOV) K(k_perm, F_PERM) K(k_sad_u8, F_SAD) K(k_pk_sub_i16, F_PKSUBI16)
K(k_cvt_f32_f16, F_CVTF16) K(k_fma_mix, F_FMAMIX) K(k_lshl_add, F_LSHLADD) K(k_min, F_MIN) K(k_alignbyte, F_ALIGNBYTE)
K(k_sub_f32_sdwa, F_SUBF32SDWA)

typedef void (*kern_t)(float*, float);
struct Entry { const char* name; kern_t k; };

int main() {
  float* out;
  hipMalloc(&out, 8192);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int dev_clock_khz = 0;
  hipDeviceGetAttribute(&dev_clock_khz, hipDeviceAttributeClockRate, 0);
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  std::printf("device: %d CUs, clock %d kHz\n", cus, dev_clock_khz);
  std::vector<Entry> es = {
      {"PAIR v_cmp vcc + v_cndmask_e32 vcc (per pair)", k_pair_vcc}, {"PAIR v_cmp_e64 s[] + v_cndmask_e64 s[] (per pair)", k_pair_s},
      {"PAIR v_cmp vcc + v_cndmask_e64 vcc (per pair)", k_pair_vcc64}, {"QUAD v_cndmask_e32 vcc + 3 v_add_f32 (per quad)", k_cnd32_spaced},
      {"v_fma_f32 a,b,1.0", k_fma_c1}, {"v_fmac_f32 independent", k_fmac_ind}, {"v_fma_f32 dst=src2", k_fma_acc},
      {"v_mul_f32 0.5,b", k_mul_c}, {"v_mul_f32 literal,b", k_mul_lit}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshl}, {"v_sub_u32", k_subu},
      {"v_cndmask_b32_e32 a,b,vcc (vcc set)", k_cnd_vcc}, {"v_cndmask_b32_e32 0,b,vcc", k_cnd_vcc0}, {"v_cndmask_b32_e64 a,b,vcc", k_cnd_vcc64},
      {"v_cvt_flr_i32_f32", k_cvtflr}, {"v_add_f32_e64 |a|,b", k_add_abs}, {"v_mul_f32 a,a,b (dst=src0)", k_mul_self}, {"v_fma_f32 a,a,b", k_fma_sq},
      {"v_mul_f32 a,a", k_mul_sq}, {"v_cmp_lt_u32 vcc", k_cmp_u32}, {"v_sub_f32 s,b", k_sub_sgpr}, {"v_add_f32 1.0,b", k_add_c}, {"v_mov_b32 v,s", k_mov_s},
      {"v_ldexp_f32", k_ldexp}, {"v_sub_f32_e64 a,-b", k_add_neg}, {"v_min3_f32", k_min3}, {"v_and_or_b32", k_andor}, {"v_add_lshl_u32", k_addlshl},
      {"v_min_i32", k_min_i32}, {"v_add_co_u32", k_add_co}, {"v_ashrrev_i32", k_ashr}, {"v_fma_f32 a,b,b", k_fma_2s},
      {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_max_f32", k_max}, {"v_mov_b32", k_mov}, {"v_and_b32", k_and}, {"v_add_u32", k_addu},
      {"v_lshrrev_b32", k_lshr}, {"v_fma_f32 a,a,a", k_fma_same}, {"v_fma_f32 a,-b,c", k_fma_neg}, {"v_fma_f32 s,b,c", k_fma_sgpr}, {"v_fmaak_f32", k_fmaak},
      {"v_mul_f32_e64 a,-b", k_mul_e64}, {"v_mul_f32 s,b", k_mul_sgpr}, {"v_cndmask_b32_e64 sgpr mask", k_cndmask_s}, {"v_cndmask_b32_e64 0,b", k_cndmask_0},
      {"v_cmp_lt_f32_e64 sgpr dst", k_cmp_s}, {"v_cvt_f32_ubyte0", k_cvt_ub0}, {"v_cvt_f32_u32", k_cvt_u2f}, {"v_fract_f32", k_fract},
      {"v_mad_i32_i24", k_mad_i24}, {"v_mul_u32_u24", k_mul_u24}, {"v_pk_mul_f32 op_sel_hi:[1,0]", k_pk_mul_b}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
      {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_fmac_f32 (dependent on itself every 8th)", k_fmac}, {"v_pk_fma_f32", k_pk_fma},
      {"v_pk_fma_f32 op_sel_hi:[0,1,1]", k_pk_fma_bcast}, {"v_pk_fma_f32 acc=dst", k_pk_fma_acc}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add},
      {"v_rcp_f32", k_rcp}, {"v_rsq_f32", k_rsq}, {"v_floor_f32", k_floor}, {"v_cvt_f32_ubyte1", k_cvt_ub1}, {"v_cvt_f32_i32", k_cvt_i32},
      {"v_cvt_i32_f32", k_cvt_f2i}, {"v_med3_f32", k_med3}, {"v_min_f32", k_min}, {"v_mad_u32_u24", k_mad_u24}, {"v_mul_lo_u32", k_mul_lo},
      {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshl_add}, {"v_bfe_u32", k_bfe}, {"v_sub_u32_sdwa", k_sub_sdwa}, {"v_sub_f32_sdwa", k_sub_f32_sdwa},
      {"v_cvt_f32_ubyte0_sdwa", k_cvt_sdwa}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32", k_cmp}, {"v_mov_b32_dpp", k_mov_dpp},
      {"v_add_f32_dpp", k_add_dpp}, {"v_fma_f64", k_fma_f64}, {"v_pk_mov_b32", k_pk_mov}, {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_alignbyte},
      {"v_sad_u8", k_sad_u8}, {"v_pk_sub_i16", k_pk_sub_i16}, {"v_cvt_f32_f16", k_cvt_f32_f16}, {"v_fma_mix_f32", k_fma_mix},
  };
  const int waves_per_simd[] = {2, 4, 8};
  std::printf("%-52s", "cycles per wave-instruction per SIMD, waves/SIMD =");
  for (int w : waves_per_simd) std::printf(" %11d", w);
  std::printf("\n");
  for (const Entry& e : es) {
    std::printf("%-52s", e.name);
    for (int w : waves_per_simd) {
      const int threads = 64 * 4 * (w > 4 ? 4 : w);
      const int grid = cus * (w > 4 ? w / 4 : 1);
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(threads), 0, 0, out, 1.0f);   // warm-up
      hipDeviceSynchronize();
      float best = 1e30f;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(e.k, dim3(grid), dim3(threads), 0, 0, out, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      // instructions per SIMD = REPS * 32 * w ; cycles = ms * clock
      unsigned st[2];
      hipMemcpy(st, (unsigned*)out + 1024, 8, hipMemcpyDeviceToHost);
      // in-kernel: s_memtime ticks (shader clock) and s_memrealtime ticks (100 MHz) of block 0's first wave around its loop
      // wall time of the whole grid (every SIMD runs w waves), in shader cycles at the clock the kernel itself observed. (The
      // stamps of one wave alone say nothing about throughput: the oldest wave of a SIMD is served first and finishes early.)
      const double mhz = st[1] ? (double)st[0] / ((double)st[1] / 100.0) : (double)dev_clock_khz * 1e-3;
      const double cyc = (double)best * 1e-3 * mhz * 1e6 / ((double)REPS * 32.0 * w);
      std::printf(" %6.2f@%4.0f", cyc, mhz);
    }
    std::printf("\n");
  }
  return 0;
}
