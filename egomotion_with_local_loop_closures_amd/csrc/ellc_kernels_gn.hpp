// Gauss-Newton kernels for gfx950: per-pixel warp / bilinear taps / 1x6 Jacobian / robust weight and the
// 6x6 normal-equation reduction (reference: PixelWisePyramid::calculatePixelWise, PixelWisePyramid.cpp:58-413;
// constant-weight variant :561-913), and the per-alignment solve + se(3) update (:441-491).
//
// Arithmetic contract: every per-pixel quantity (warped point, taps, J, residual, weight) is computed with
// the reference's expression order in IEEE f32 — or f64 where the reference's pow() promotes — with
// contraction disabled for the translation unit, so it is bit-identical to the CPU path. Only the
// *summation order* of the 27 accumulators differs (per-thread FMA chains, wave DPP tree, fixed-order
// f64 combine of block partials instead of three serial row bands).
#pragma once
#include "ellc_device.hpp"
#include "ellc_se3.hpp"

namespace ellc {

// In-kernel cycle stamps: diagnostic builds only (make STAMPS=1); the stamp buffer is read by nothing else.
#ifdef ELLC_STAMPS
__device__ unsigned long long g_stamps[64];
__device__ unsigned long long g_block_stamps[4 * 8192];   // per block: start, prologue end, pixels done, end
#define ELLC_BSTAMP(slot)                                                                                    \
  do {                                                                                                       \
    if (threadIdx.x == 0) {                                                                                  \
      unsigned long long t__;                                                                                \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                        \
      const unsigned bid__ = blockIdx.y * gridDim.x + blockIdx.x;                                            \
      if (bid__ < 8192) g_block_stamps[bid__ * 4 + slot] = t__;                                              \
    }                                                                                                        \
  } while (0)
#define ELLC_STAMP(id)                                                                                       \
  do {                                                                                                       \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                                            \
      unsigned long long t__;                                                                                \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                            \
      g_stamps[id] = t__;                                                                                    \
    }                                                                                                        \
  } while (0)
// gn_fca_persist, per round of eight of its blocks (alignment 0), 12 words: 0 loop top, 1 level tables read + first record requested,
// 2 records gathered, 3 solved, 4 pass: constants formed, 5 taps used, 6 pixel done, 7 pass returned, 8 wave sums, 9 block barrier,
// 10 record stored, 11 level << 8 | works  (tools/dbg/persist_trace.py)
__device__ unsigned long long g_ptrace[8 * 64 * 12];
__device__ __forceinline__ int ptrace_slot(int sub) {
  return sub == 0 ? 0 : sub == 1 ? 1 : sub == 8 ? 2 : sub == 16 ? 3 : sub == 48 ? 4 : sub == 100 ? 5 : sub == 200 ? 6 : sub == 255 ? 7 : -1;
}
__device__ int* ptrace_base_ptr() { __shared__ int base; return &base; }
#define ELLC_PTRACE_ROUND(sub, round)                                                                        \
  do {                                                                                                       \
    const int ps__ = ptrace_slot(sub);                                                                       \
    if (threadIdx.x == 0) *ptrace_base_ptr() = (blockIdx.y == 0 && ps__ >= 0 && (round) < 64) ? (ps__ * 64 + (round)) * 12 : -1; \
  } while (0)
#define ELLC_PTRACE(k, val)                                                                                  \
  do {                                                                                                       \
    if (threadIdx.x == 0) {                                                                                  \
      const int pb__ = *ptrace_base_ptr();                                                                   \
      if (pb__ >= 0 && pb__ <= (8 * 64 - 1) * 12) {                                                          \
        unsigned long long t__;                                                                              \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                     \
        g_ptrace[pb__ + (k)] = (k) == 11 ? (unsigned long long)(val) : t__;                                  \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
#else
#define ELLC_STAMP(id) do { } while (0)
#define ELLC_BSTAMP(slot) do { } while (0)
#define ELLC_PTRACE(k, val) do { } while (0)
#define ELLC_PTRACE_ROUND(sub, round) do { } while (0)
#endif
// -DELLC_SEQ_STAMPS (tools/dbg/seq_stamps.py): the launches of ONE sequence on one clock — block (0, 0)'s entry into launch n of the
// schedule in g_seq_stamps[n], its exit in g_seq_stamps[32 + n] (gn_fca_fused only)
#ifdef ELLC_SEQ_STAMPS
__device__ unsigned long long g_seq_stamps[64];
#define ELLC_SEQSTAMP(off, seq)                                                                              \
  do {                                                                                                       \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                                            \
      unsigned long long t__;                                                                                \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                        \
      g_seq_stamps[(off) + ((seq) & 31)] = t__;                                                              \
    }                                                                                                        \
  } while (0)
#else
#define ELLC_SEQSTAMP(off, seq) do { } while (0)
#endif

// Pointers read out of device-resident tables are "flat" to the compiler (it emits flat_load and 64-bit address
// arithmetic per lane). Everything this library indexes lives in global memory, so the hot kernels say so.
#define ELLC_GLOBAL __attribute__((address_space(1)))
typedef const ELLC_GLOBAL uint8_t* g_u8;
typedef const ELLC_GLOBAL float* g_f32;
typedef const ELLC_GLOBAL double* g_f64;
typedef const ELLC_GLOBAL uint32_t* g_u32;
template <class T>
__device__ __forceinline__ const ELLC_GLOBAL T* as_global(const T* p) { return (const ELLC_GLOBAL T*)p; }
template <class T>
__device__ __forceinline__ ELLC_GLOBAL T* as_global_rw(T* p) { return (ELLC_GLOBAL T*)p; }

// ---------------------------------------------------------------------------------------------------
// one DPP step of a wave-wide sum (see wave_sum_rows)
template <int CTRL, int RMASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false);
  return v + __builtin_bit_cast(float, x);
}
// ExternVariable.h:232 clamps (double)v away from zero at +-1e-10 and rounds back to f32. For an f32 argument that is
// a pure f32 select: with C = RN_f32(1e-10) = 0x1.b7cdfep-34 > 1e-10 and its predecessor below 1e-10,
// "(double)v < 1e-10" <=> "v < C" and the clamped result (float)1e-10 is C (negative side mirrored; NaN passes).
__device__ __forceinline__ float unzero_f(float v) {
  const float C = 0x1.b7cdfep-34f;
  return (v < 0.0f) ? ((v > -C) ? -C : v) : ((v < C) ? C : v);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 32-bit load at any byte address of a global buffer (gfx9+ global memory accepts unaligned dword accesses; the
// align-1 type lets the compiler choose the instruction, so this stays correct if a target did not)
typedef uint32_t u32_align1 __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t load_u32_unaligned(g_u8 base, unsigned byte_off) {
  return *(const ELLC_GLOBAL u32_align1*)(base + byte_off);
}
template <int N>
__device__ __forceinline__ float byte_f32(uint32_t w) { return (float)((w >> (8 * N)) & 0xffu); }   // v_cvt_f32_ubyteN

// (r03 built an LDS-staged window variant of the taps behind -DELLC_WINDOWS: measured slower; removed from the
// tree in r04 — it is in the history at 99afb34.)
#define ELLC_LDS __attribute__((address_space(3)))

struct Taps {
  float I;      // u8 tap (Frame.h:181-279), -1 when all four taps are out of bounds
  float gx, gy; // gradient taps (Frame.h:283-394) on frame::calculateGradient's planes (Frame.cpp:185-285)
};

// The general path of the taps: per-tap bounds tests of the reference (Frame.h:211-275). FAST: the gradients are returned twice
// their value, as the interior branches of that mode return them.
template <bool WANT_GRAD, bool FAST, bool LAT = false>
__device__ __forceinline__ Taps tap_general(g_u8 img, int sw, int cols, int rows, float x1, float y1) {
  Taps o;
  const float fx0 = floorf(x1), fy0 = floorf(y1);
  const float wx = x1 - fx0, wy = y1 - fy0;
  const float omx = 1.0f - wx, omy = 1.0f - wy;
  if (x1 != x1 || y1 != y1) {  // NaN: reference behaviour undefined; treated as out of bounds
    o.I = -1.0f; o.gx = 0.0f; o.gy = 0.0f;
    return o;
  }
  const float nC = (float)(cols - 1), nR = (float)(rows - 1);
  const bool xf_bad = (fx0 < 0.0f) || (fx0 > nC);
  const bool xc_bad = (x1 < 0.0f) || (x1 > nC);
  const bool yf_bad = (fy0 < 0.0f) || (fy0 > nR);
  const bool yc_bad = (y1 < 0.0f) || (y1 > nR);
  const bool v00 = !(xf_bad || yf_bad), v01 = !(xc_bad || yf_bad), v10 = !(xf_bad || yc_bad), v11 = !(xc_bad || yc_bad);
  if (!(v00 || v01 || v10 || v11)) {
    o.I = -1.0f; o.gx = 0.0f; o.gy = 0.0f;   // gradient taps: four zero samples interpolate to 0
    return o;
  }
  const int x0 = (int)fminf(fmaxf(fx0, -4.0f), nC + 4.0f);
  const int y0 = (int)fminf(fmaxf(fy0, -4.0f), nR + 4.0f);
  const int xb = clampi(x0, 0, cols - 1), xc = clampi(x0 + 1, 0, cols - 1);
  const int yb = clampi(y0, 0, rows - 1), yc = clampi(y0 + 1, 0, rows - 1);
  // uniform base pointer + unsigned 32-bit lane offsets (SGPR-base global loads, no 64-bit lane arithmetic)
  const unsigned rb = __umul24((unsigned)yb, (unsigned)sw), rc = __umul24((unsigned)yc, (unsigned)sw);   // rows are clamped to >= 0
  if constexpr (LAT) {
    // The latency regime (gn_fca_persist: one pixel per thread and round, nothing else in flight): all the samples are requested
    // TOGETHER, whatever the register allocator would like (the barrier) — in the 168-register resident kernel it had turned the
    // first four into load, wait, load, wait, and a wave on the image border, i.e. every wave of the two coarse levels (the depth
    // pyramid's border shrinks with the level), took 0.65 us longer per pixel than an interior one (tools/dbg/persist_trace.py,
    // r06). The same loads and the same arithmetic as below: the same taps. NOT for the batch pipeline's kernels: with the
    // barrier their coarse-level launches were 12 % slower (and 4.5 % of a whole step), NOTEBOOK 6.8.
    const int xa = clampi(x0 - 1, 0, cols - 1), xd = clampi(x0 + 2, 0, cols - 1);
    const int ya = clampi(y0 - 1, 0, rows - 1), yd = clampi(y0 + 2, 0, rows - 1);
    const unsigned ra = __umul24((unsigned)ya, (unsigned)sw), rd = __umul24((unsigned)yd, (unsigned)sw);
    const uint32_t bbb = img[rb + (unsigned)xb], bbc = img[rb + (unsigned)xc], bcb = img[rc + (unsigned)xb], bcc = img[rc + (unsigned)xc];
    uint32_t bba = 0, bbd = 0, bca = 0, bcd = 0, bab = 0, bac = 0, bdb = 0, bdc = 0;
    if (WANT_GRAD) {
      bba = img[rb + (unsigned)xa]; bbd = img[rb + (unsigned)xd];
      bca = img[rc + (unsigned)xa]; bcd = img[rc + (unsigned)xd];
      bab = img[ra + (unsigned)xb]; bac = img[ra + (unsigned)xc];
      bdb = img[rd + (unsigned)xb]; bdc = img[rd + (unsigned)xc];
    }
    __builtin_amdgcn_sched_barrier(0);
    const float Pbb = (float)bbb, Pbc = (float)bbc, Pcb = (float)bcb, Pcc = (float)bcc;
    {
      const float p00 = v00 ? Pbb : 0.0f, p01 = v01 ? Pbc : 0.0f, p10 = v10 ? Pcb : 0.0f, p11 = v11 ? Pcc : 0.0f;
      const float top = (omx * p00) + (wx * p01);
      const float btm = (omx * p10) + (wx * p11);
      o.I = (omy * top) + (wy * btm);
    }
    if (WANT_GRAD) {
      const float Pba = (float)bba, Pbd = (float)bbd, Pca = (float)bca, Pcd = (float)bcd;
      const float Pab = (float)bab, Pac = (float)bac, Pdb = (float)bdb, Pdc = (float)bdc;
      const float sx0 = (x0 <= 0 || x0 >= cols - 1) ? 1.0f : 0.5f;
      const float sx1 = (x0 + 1 <= 0 || x0 + 1 >= cols - 1) ? 1.0f : 0.5f;
      const float sy0 = (y0 <= 0 || y0 >= rows - 1) ? 1.0f : 0.5f;
      const float sy1 = (y0 + 1 <= 0 || y0 + 1 >= rows - 1) ? 1.0f : 0.5f;
      float g00 = sx0 * (Pbc - Pba), g01 = sx1 * (Pbd - Pbb), g10 = sx0 * (Pcc - Pca), g11 = sx1 * (Pcd - Pcb);
      g00 = v00 ? g00 : 0.0f; g01 = v01 ? g01 : 0.0f; g10 = v10 ? g10 : 0.0f; g11 = v11 ? g11 : 0.0f;
      float top = (omx * g00) + (wx * g01);
      float btm = (omx * g10) + (wx * g11);
      o.gx = (omy * top) + (wy * btm);
      float h00 = sy0 * (Pcb - Pab), h01 = sy0 * (Pcc - Pac), h10 = sy1 * (Pdb - Pbb), h11 = sy1 * (Pdc - Pbc);
      h00 = v00 ? h00 : 0.0f; h01 = v01 ? h01 : 0.0f; h10 = v10 ? h10 : 0.0f; h11 = v11 ? h11 : 0.0f;
      top = (omx * h00) + (wx * h01);
      btm = (omx * h10) + (wx * h11);
      o.gy = (omy * top) + (wy * btm);
      if (FAST) { o.gx *= 2.0f; o.gy *= 2.0f; }
    } else {
      o.gx = 0.0f; o.gy = 0.0f;
    }
    return o;
  }
  const float Pbb = (float)img[rb + (unsigned)xb], Pbc = (float)img[rb + (unsigned)xc];
  const float Pcb = (float)img[rc + (unsigned)xb], Pcc = (float)img[rc + (unsigned)xc];
  {
    const float p00 = v00 ? Pbb : 0.0f, p01 = v01 ? Pbc : 0.0f, p10 = v10 ? Pcb : 0.0f, p11 = v11 ? Pcc : 0.0f;
    const float top = (omx * p00) + (wx * p01);
    const float btm = (omx * p10) + (wx * p11);
    o.I = (omy * top) + (wy * btm);
  }
  if (WANT_GRAD) {
    const int xa = clampi(x0 - 1, 0, cols - 1), xd = clampi(x0 + 2, 0, cols - 1);
    const int ya = clampi(y0 - 1, 0, rows - 1), yd = clampi(y0 + 2, 0, rows - 1);
    const unsigned ra = __umul24((unsigned)ya, (unsigned)sw), rd = __umul24((unsigned)yd, (unsigned)sw);
    const float Pba = (float)img[rb + (unsigned)xa], Pbd = (float)img[rb + (unsigned)xd];
    const float Pca = (float)img[rc + (unsigned)xa], Pcd = (float)img[rc + (unsigned)xd];
    const float Pab = (float)img[ra + (unsigned)xb], Pac = (float)img[ra + (unsigned)xc];
    const float Pdb = (float)img[rd + (unsigned)xb], Pdc = (float)img[rd + (unsigned)xc];
    // scale 1 on the border column/row of the tap itself, 0.5 inside
    const float sx0 = (x0 <= 0 || x0 >= cols - 1) ? 1.0f : 0.5f;
    const float sx1 = (x0 + 1 <= 0 || x0 + 1 >= cols - 1) ? 1.0f : 0.5f;
    const float sy0 = (y0 <= 0 || y0 >= rows - 1) ? 1.0f : 0.5f;
    const float sy1 = (y0 + 1 <= 0 || y0 + 1 >= rows - 1) ? 1.0f : 0.5f;
    // d/dx at (yb,x0) (yb,x0+1) (yc,x0) (yc,x0+1)
    float g00 = sx0 * (Pbc - Pba), g01 = sx1 * (Pbd - Pbb), g10 = sx0 * (Pcc - Pca), g11 = sx1 * (Pcd - Pcb);
    g00 = v00 ? g00 : 0.0f; g01 = v01 ? g01 : 0.0f; g10 = v10 ? g10 : 0.0f; g11 = v11 ? g11 : 0.0f;
    float top = (omx * g00) + (wx * g01);
    float btm = (omx * g10) + (wx * g11);
    o.gx = (omy * top) + (wy * btm);
    // d/dy at the same four positions
    float h00 = sy0 * (Pcb - Pab), h01 = sy0 * (Pcc - Pac), h10 = sy1 * (Pdb - Pbb), h11 = sy1 * (Pdc - Pbc);
    h00 = v00 ? h00 : 0.0f; h01 = v01 ? h01 : 0.0f; h10 = v10 ? h10 : 0.0f; h11 = v11 ? h11 : 0.0f;
    top = (omx * h00) + (wx * h01);
    btm = (omx * h10) + (wx * h11);
    o.gy = (omy * top) + (wy * btm);
    if (FAST) { o.gx *= 2.0f; o.gy *= 2.0f; }   // as the interior branch of this mode returns them
  } else {
    o.gx = 0.0f; o.gy = 0.0f;
  }
  return o;
}

// The three bilinear taps of one warped point. The gradient planes are never materialised: the four
// gradient samples are rebuilt from the u8 image with the reference's border rules (interior central
// difference x0.5, one-sided without 0.5 on the border), which is exact in f32.
// Fast path: when every active lane of the wave samples the interior (all 16 neighbours in range, none of the
// four taps on a border column/row) the validity selects, clamps and border scales drop out; the arithmetic
// that remains is the same expression, so the results are bit-identical to the general path.
// FAST (cfg.arith = ELLC_ARITH_FAST): the interior path interpolates in the fused form a + w (b - a) and applies the 0.5 of
// the central differences once to the interpolated gradient: same values to within a few ulp, 19 instructions fewer.
// after_issue() is called exactly once, on the interior path right after the tap loads have been issued: the pixel loops
// request the next pixel's record there. Vector loads return in order, so a record load issued BEFORE the taps would have
// to come back from HBM before the (cache-resident) taps count as complete; issued behind them it stays in flight while
// this pixel's arithmetic runs.
struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };
// A prefetch functor that also says "latency regime" (tap_general<.., LAT>): how gn_fca_persist's passes mark their taps
template <class F> struct LatPF { F f; __device__ __forceinline__ void operator()() const { f(); } };
template <class T> struct is_lat_pf { static constexpr bool value = false; };
template <class F> struct is_lat_pf<LatPF<F>> { static constexpr bool value = true; };
template <class F> __device__ __forceinline__ LatPF<F> lat_pf(F f) { return LatPF<F>{f}; }
template <bool WANT_GRAD, bool FAST = false, class AfterIssue = NoPrefetch>
__device__ __forceinline__ Taps tap_point(g_u8 img, int sw, int cols, int rows, float x1, float y1, AfterIssue after_issue = AfterIssue()) {
  Taps o;
  const float fx0 = floorf(x1), fy0 = floorf(y1);
  const float wx = x1 - fx0, wy = y1 - fy0;
  const float omx = 1.0f - wx, omy = 1.0f - wy;
  // fx0 in [1, cols-3] and fy0 in [1, rows-3], false for NaN: a value equals its clamp (v_med3_f32) exactly when it is in range.
  const bool interior = (__builtin_amdgcn_fmed3f(fx0, 1.0f, (float)(cols - 3)) == fx0) & (__builtin_amdgcn_fmed3f(fy0, 1.0f, (float)(rows - 3)) == fy0);
  if (__builtin_amdgcn_ballot_w64(!interior) == 0ull) {
    // The 4x4 neighbourhood is fetched as one (byte-unaligned) 32-bit word per image row: columns x0-1 .. x0+2.
    const int x0 = (int)fx0, y0 = (int)fy0;
    uint32_t wa = 0, wb, wc, wd = 0;
    const unsigned ob = __umul24((unsigned)y0, (unsigned)sw) + (unsigned)x0 - 1u, oc = ob + (unsigned)sw;   // y0 >= 1, sw < 2^24: full-rate v_mad_u32_u24
    wb = load_u32_unaligned(img, ob);
    wc = load_u32_unaligned(img, oc);
    if (WANT_GRAD) { wa = load_u32_unaligned(img, ob - (unsigned)sw); wd = load_u32_unaligned(img, oc + (unsigned)sw); }
    __builtin_amdgcn_sched_barrier(0);
    after_issue();
    __builtin_amdgcn_sched_barrier(0);
    const float Pbb = byte_f32<1>(wb), Pbc = byte_f32<2>(wb), Pcb = byte_f32<1>(wc), Pcc = byte_f32<2>(wc);
    if (FAST) {
      const float top = __builtin_fmaf(wx, Pbc - Pbb, Pbb);
      const float btm = __builtin_fmaf(wx, Pcc - Pcb, Pcb);
      o.I = __builtin_fmaf(wy, btm - top, top);
      if (WANT_GRAD) {
        const float Pba = byte_f32<0>(wb), Pbd = byte_f32<3>(wb), Pca = byte_f32<0>(wc), Pcd = byte_f32<3>(wc);
        const float Pab = byte_f32<1>(wa), Pac = byte_f32<2>(wa), Pdb = byte_f32<1>(wd), Pdc = byte_f32<2>(wd);
        const float g00 = Pbc - Pba, g01 = Pbd - Pbb, g10 = Pcc - Pca, g11 = Pcd - Pcb;   // twice the central differences
        float t2 = __builtin_fmaf(wx, g01 - g00, g00);
        float b2 = __builtin_fmaf(wx, g11 - g10, g10);
        o.gx = __builtin_fmaf(wy, b2 - t2, t2);   // FAST: TWICE the gradient (the caller folds the 0.5 into fx, fy)
        const float h00 = Pcb - Pab, h01 = Pcc - Pac, h10 = Pdb - Pbb, h11 = Pdc - Pbc;
        t2 = __builtin_fmaf(wx, h01 - h00, h00);
        b2 = __builtin_fmaf(wx, h11 - h10, h10);
        o.gy = __builtin_fmaf(wy, b2 - t2, t2);
      } else {
        o.gx = 0.0f; o.gy = 0.0f;
      }
      return o;
    }
    {
      const float top = (omx * Pbb) + (wx * Pbc);
      const float btm = (omx * Pcb) + (wx * Pcc);
      o.I = (omy * top) + (wy * btm);
    }
    if (WANT_GRAD) {
      const float Pba = byte_f32<0>(wb), Pbd = byte_f32<3>(wb), Pca = byte_f32<0>(wc), Pcd = byte_f32<3>(wc);
      const float Pab = byte_f32<1>(wa), Pac = byte_f32<2>(wa), Pdb = byte_f32<1>(wd), Pdc = byte_f32<2>(wd);
      const float g00 = 0.5f * (Pbc - Pba), g01 = 0.5f * (Pbd - Pbb), g10 = 0.5f * (Pcc - Pca), g11 = 0.5f * (Pcd - Pcb);
      float top = (omx * g00) + (wx * g01);
      float btm = (omx * g10) + (wx * g11);
      o.gx = (omy * top) + (wy * btm);
      const float h00 = 0.5f * (Pcb - Pab), h01 = 0.5f * (Pcc - Pac), h10 = 0.5f * (Pdb - Pbb), h11 = 0.5f * (Pdc - Pbc);
      top = (omx * h00) + (wx * h01);
      btm = (omx * h10) + (wx * h11);
      o.gy = (omy * top) + (wy * btm);
    } else {
      o.gx = 0.0f; o.gy = 0.0f;
    }
    return o;
  }
  after_issue();
  return tap_general<WANT_GRAD, FAST, is_lat_pf<AfterIssue>::value>(img, sw, cols, rows, x1, y1);
}

// ---------------------------------------------------------------------------------------------------
// The taps of the tolerance mode (cfg.arith = ELLC_ARITH_FAST), written against what the vector ALU of gfx950 issues at which
// rate (tools/micro/valu_rates.hip, r04; cycles per wave-instruction with four or more waves per SIMD):
//   ~2.4  v_add / sub / mul / fma / fmac_f32, v_mov, v_and / or / xor, v_add / sub_u32, v_lshrrev, v_ashrrev — with VGPR, inline or
//         literal operands only;
//   ~4.4  every other vector instruction — conversions, floor / fract, min / max / med3, compares, selects, the three-operand
//         integer forms, v_lshlrev, SDWA and DPP forms, f64, the packed f32 forms (v_pk_fma_f32 does two fmas for the price of 1.8) —
//         AND any instruction of the first group that names an SGPR;
//   ~8.5  v_rcp / v_rsq_f32.
// Hence: the per-block constants live in VGPRs (FcafConst); the four rows of the 4 x 4 neighbourhood are four uniform base
// pointers with ONE lane offset (no per-row address arithmetic in the vector ALU); floor and fraction come from v_cvt_flr_i32_f32
// and v_fract_f32 (the interior test is two unsigned compares on the integers); the twelve bytes are converted by twelve
// v_cvt_f32_ubyteN and differenced in f32 (left to itself the compiler subtracts the bytes with SDWA integer instructions and
// converts the differences: 26 slow instructions for these 12 + 8 fast ones).
// What it bought (r04, tools/ab_levels.sh, tools/ab_libs.sh, interleaved on one box): 124 -> 110 vector instructions per pixel,
// 460 -> 396 cycles of modelled issue (tools/isa_cost.py); the level-0 launch over 128 alignments 45.8 -> 43.6 us, the batch
// pipeline 0.1331 -> 0.1312 ms per step. Far less than the instructions saved: the pass is bound by the CU's memory pipeline as much
// as by its vector ALU (the same loop without its loads: 214 of 389 us at 1280x960 dense, 33 of 44 us at 640x480).
struct TapRows { g_u8 ra, rb, rc, rd; };   // image - 1 + (-1, 0, 1, 2) * pitch: column x0 - 1 of rows y0 - 1 .. y0 + 2 sits at row pointer + y0 * pitch + x0
__device__ __forceinline__ TapRows tap_rows(g_u8 img, int sw) {
  TapRows r;
  r.rb = img - 1;
  r.ra = r.rb - sw;
  r.rc = r.rb + sw;
  r.rd = r.rc + sw;
  return r;
}
template <int N>
__device__ __forceinline__ float cvt_ubyte(uint32_t w) {   // opaque to the optimiser on purpose, see above
  float f;
  if (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w));
  else if (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w));
  else if (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w));
  else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w));
  return f;
}
__device__ __forceinline__ int cvt_floor_i32(float x) {   // floor, then the saturating conversion (NaN gives 0)
  int i;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(x));
  return i;
}
// Two halves: tap_request_f decides interior / general for the wave and requests the four rows; tap_finish_f turns them (or, on the
// general path, the synchronous per-tap loads) into the taps. (r04 put the next pixel's request in front of the current pixel's
// finish — a software pipeline over pixels: 1280x960 dense 387 against 389 us, 640x480 5 % slower, 125 registers: not kept.)
struct TapReq {
  uint32_t wa, wb, wc, wd;   // rows y0 - 1 .. y0 + 2, columns x0 - 1 .. x0 + 2 (valid when interior)
  bool interior;             // wave-uniform: every lane that was active at the request samples the interior
};
template <bool WANT_GRAD, class AfterIssue = NoPrefetch>
__device__ __forceinline__ TapReq tap_request_f(const TapRows& tr, int sw, int cols, int rows, float x1, float y1, AfterIssue after_issue = AfterIssue()) {
  TapReq q;
  q.wa = 0; q.wd = 0;
  const int x0 = cvt_floor_i32(x1), y0 = cvt_floor_i32(y1);
  // x0 in [1, cols - 3] and y0 in [1, rows - 3] (all 16 neighbours in range, none of the four taps on a border column / row); a
  // NaN coordinate converts to 0 and an infinite one saturates: neither is interior
  const bool interior = ((unsigned)(x0 - 1) <= (unsigned)(cols - 4)) & ((unsigned)(y0 - 1) <= (unsigned)(rows - 4));
  q.interior = (__builtin_amdgcn_ballot_w64(!interior) == 0ull);
  // (r04 measured the straight-line form — the rows requested unconditionally, a lane that is not interior asking for the image's
  // first bytes — which a software pipeline over pixels needs: 5 % slower on the batch pipeline, the coarse levels' waves on the
  // image border pay for four requests they do not use)
  q.wb = 0; q.wc = 0;
  if (q.interior) {
    const unsigned off = __umul24((unsigned)y0, (unsigned)sw) + (unsigned)x0;
#ifdef ELLC_X_LDSTAPS   // variant builds only (tools/pmc_ldstaps.sh; WRONG values): what the four rows would cost as reads of an LDS
                        // window that is already there — two aligned dwords + v_alignbit per row, no staging, no window arithmetic
    __shared__ uint32_t xl[4 * 1024 + 4];
    const unsigned o = (off >> 2) & 1023u, shb = (off & 3u) * 8u;
    q.wb = __builtin_amdgcn_alignbit(xl[o + 1], xl[o], shb);
    q.wc = __builtin_amdgcn_alignbit(xl[1024 + o + 1], xl[1024 + o], shb);
    if (WANT_GRAD) { q.wa = __builtin_amdgcn_alignbit(xl[2048 + o + 1], xl[2048 + o], shb); q.wd = __builtin_amdgcn_alignbit(xl[3072 + o + 1], xl[3072 + o], shb); }
#else
    q.wb = load_u32_unaligned(tr.rb, off);
    q.wc = load_u32_unaligned(tr.rc, off);
    if (WANT_GRAD) { q.wa = load_u32_unaligned(tr.ra, off); q.wd = load_u32_unaligned(tr.rd, off); }
#endif
    __builtin_amdgcn_sched_barrier(0);
    after_issue();   // behind the row requests: vector loads return in issue order
    __builtin_amdgcn_sched_barrier(0);
  } else {
    after_issue();
  }
  return q;
}
template <bool WANT_GRAD, bool LAT = false>
__device__ __forceinline__ Taps tap_finish_f(const TapReq& q, g_u8 img, int sw, int cols, int rows, float x1, float y1) {
  if (q.interior) {
    Taps o;
    const float wx = __builtin_amdgcn_fractf(x1), wy = __builtin_amdgcn_fractf(y1);   // x - floor(x), exact for x >= 1
    const float Pbb = cvt_ubyte<1>(q.wb), Pbc = cvt_ubyte<2>(q.wb), Pcb = cvt_ubyte<1>(q.wc), Pcc = cvt_ubyte<2>(q.wc);
    const float top = __builtin_fmaf(wx, Pbc - Pbb, Pbb);
    const float btm = __builtin_fmaf(wx, Pcc - Pcb, Pcb);
    o.I = __builtin_fmaf(wy, btm - top, top);
    if (WANT_GRAD) {
      const float Pba = cvt_ubyte<0>(q.wb), Pbd = cvt_ubyte<3>(q.wb), Pca = cvt_ubyte<0>(q.wc), Pcd = cvt_ubyte<3>(q.wc);
      const float Pab = cvt_ubyte<1>(q.wa), Pac = cvt_ubyte<2>(q.wa), Pdb = cvt_ubyte<1>(q.wd), Pdc = cvt_ubyte<2>(q.wd);
      const float g00 = Pbc - Pba, g01 = Pbd - Pbb, g10 = Pcc - Pca, g11 = Pcd - Pcb;   // twice the central differences
      float t2 = __builtin_fmaf(wx, g01 - g00, g00);
      float b2 = __builtin_fmaf(wx, g11 - g10, g10);
      o.gx = __builtin_fmaf(wy, b2 - t2, t2);   // TWICE the gradient (the caller folds the 0.5 in)
      const float h00 = Pcb - Pab, h01 = Pcc - Pac, h10 = Pdb - Pbb, h11 = Pdc - Pbc;
      t2 = __builtin_fmaf(wx, h01 - h00, h00);
      b2 = __builtin_fmaf(wx, h11 - h10, h10);
      o.gy = __builtin_fmaf(wy, b2 - t2, t2);
    } else {
      o.gx = 0.0f; o.gy = 0.0f;
    }
    return o;
  }
  return tap_general<WANT_GRAD, true, LAT>(img, sw, cols, rows, x1, y1);
}

// a / b for a per-level constant b with rb = RN(1/b): q = RN(a rb), e = a - b q (exact, fma), RN(q + e rb).
// Correctly rounded (Markstein's final-step theorem); verified exhaustively on the host for the context's fx, fy
// over all 2^23 mantissas of a before LevelGeom::divc_ok is set (ellc_hip.hip: verify_div_const).
__device__ __forceinline__ float div_const(float a, float b, float rb) {
  const float q = a * rb;
  const float e = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(e, rb, q);
}

// Two IEEE f32 divisions a0/b0, a1/b1 with the refinement arithmetic of both packed (v_pk_fma_f32). Operation for
// operation this is the sequence the compiler emits for `/` on gfx9 with f32 denormals enabled (div_scale x2, rcp,
// three-step Newton refinement, div_fmas, div_fixup), so the quotients are the correctly rounded ones `/` gives;
// checked against `/` on the device over special values and random bit patterns (tests/test_gpu_gn.py).
__device__ __forceinline__ void div_pair_ieee(float a0, float b0, float a1, float b1, float& q0, float& q1) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  bool vd0, vd1, vn0, vn1;
  const v2 ds = {__builtin_amdgcn_div_scalef(a0, b0, false, &vd0), __builtin_amdgcn_div_scalef(a1, b1, false, &vd1)};
  const v2 ns = {__builtin_amdgcn_div_scalef(a0, b0, true, &vn0), __builtin_amdgcn_div_scalef(a1, b1, true, &vn1)};
  const v2 r = {__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
  const v2 nds = -ds;
  const v2 f0 = __builtin_elementwise_fma(nds, r, (v2)(1.0f));
  const v2 f1 = __builtin_elementwise_fma(f0, r, r);
  const v2 mul = ns * f1;
  const v2 f2 = __builtin_elementwise_fma(nds, mul, ns);
  const v2 f3 = __builtin_elementwise_fma(f2, f1, mul);
  const v2 f4 = __builtin_elementwise_fma(nds, f3, ns);
  q0 = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(f4.x, f1.x, f3.x, vn0), b0, a0);
  q1 = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(f4.y, f1.y, f3.y, vn1), b1, a1);
}

struct Warp { float px, py, pz, wx, wy; };

// PixelWisePyramid.cpp:241-262: rigid transform of the back-projected pixel (X, Y, Z) and projection
template <bool PAIR>
__device__ __forceinline__ Warp warp_point(float X, float Y, float Z, const LevelGeom& g, const float* S) {
  Warp o;
  o.px = (S[0] * X) + (S[1] * Y) + (S[2] * Z) + (S[3]);
  o.py = (S[4] * X) + (S[5] * Y) + (S[6] * Z) + (S[7]);
  o.pz = (S[8] * X) + (S[9] * Y) + (S[10] * Z) + (S[11]);
  o.pz = unzero_f(o.pz);
  float qx, qy;
  if (PAIR) div_pair_ieee(o.px, o.pz, o.py, o.pz, qx, qy);
  else { qx = o.px / o.pz; qy = o.py / o.pz; }
  o.wx = (qx * g.fx) + g.cx;
  o.wy = (qy * g.fy) + g.cy;
  return o;
}

// PixelWisePyramid.cpp:236-240: back-projection of pixel (x, y) at depth Z, then the warp
template <bool DIVC>
__device__ __forceinline__ Warp warp_pixel(int x, int y, float Z, const LevelGeom& g, const float* S) {
  const float aX = ((float)x - g.cx) * Z, aY = ((float)y - g.cy) * Z;
  const float X = DIVC ? div_const(aX, g.fx, g.rfx) : aX / g.fx;
  const float Y = DIVC ? div_const(aY, g.fy, g.rfy) : aY / g.fy;
  return warp_point<false>(X, Y, Z, g, S);
}

// PixelWisePyramid.cpp:296-320: 1x6 steepest-descent row at the reference pixel / reference depth.
// invZ = pow(depth,-1) evaluated in double (the reference's promotion), shared with the weight term.
template <bool DIVC>
__device__ __forceinline__ void jacobian_row(float gradx, float grady, int x, int y, double invZ, const LevelGeom& g, float J[6]) {
  const float u = -g.cx + (float)x;
  const float v = -g.cy + (float)y;
  const float vu = v * u;
  const float jb0 = (float)((double)grady * as_global(g.rowA)[y]);
  const float jt0 = gradx * (DIVC ? div_const(-vu, g.fy, g.rfy) : (-vu / g.fy));
  const float jb1 = grady * (DIVC ? div_const(vu, g.fx, g.rfx) : (vu / g.fx));
  const float jt1 = (float)((double)gradx * as_global(g.colA)[x]);
  const float jb2 = grady * as_global(g.colB)[x];
  const float jt2 = gradx * as_global(g.rowB)[y];
  const float jt3 = (float)((double)gradx * ((double)g.fx * invZ));
  const float jb4 = (float)((double)grady * ((double)g.fy * invZ));
  const float jb5 = (float)((double)grady * ((double)(-v) * invZ));
  const float jt5 = (float)((double)gradx * ((double)(-u) * invZ));
  J[0] = jt0 + jb0;
  J[1] = jt1 + jb1;
  J[2] = jt2 + jb2;
  J[3] = jt3 + 0.0f;
  J[4] = 0.0f + jb4;
  J[5] = jt5 + jb5;
}

// PixelWisePyramid.cpp:341-358
// d = 1.0f / Z is passed in: it equals (float)(1.0 / (double)Z) exactly (the reciprocal of a 24-bit number cannot sit
// within 2^-54 of a 25-bit midpoint unless it is itself representable, so the double rounding is innocuous).
__device__ __forceinline__ float fca_weight(const Warp& w, float d, float residual, float gradx, float grady, float s,
                                            const LevelGeom& g, float tx, float ty, float tz) {
  const float gx = g.fx * gradx;
  const float gy = g.fy * grady;
  const float den = (w.pz * w.pz) * d;
  float g0, g1;
  div_pair_ieee(tx * w.pz - tz * w.px, den, ty * w.pz - tz * w.py, den, g0, g1);
  const float drpdd = gx * g0 + gy * g1;
  const float w_p = 1.0f / (16.0f + (s * drpdd) * drpdd);
  const float weighted_rp = fabsf(residual * sqrtf(w_p));
  const float wh = fabsf(weighted_rp < 1.5f ? 1.0f : 1.5f / weighted_rp);
  return wh * w_p;
}

struct GnArgs {
  const LevelGeom* geom;        // [levels]
  const KfLevelDev* kf_tab;     // [levels][max_kf]
  const FrLevelDev* fr_tab;     // [levels][max_fr]
  const int* kf_slot;           // [B]
  const int* fr_slot;           // [B]
  AlignState* state;            // [B]
  float* partials;              // [B][ELLC_NBLK_MAX][ELLC_PART_STRIDE]
  float* planes;                // debug: 10 planes of n floats (B must be 1), else null
  int level, max_kf, max_fr, nblk;
  int save_w;                   // write per-pixel weights of this iteration into kf.wlast
};

// Wave-wide sums of NV values by a halving transpose: v_permlane32_swap exchanges the upper half of one register with the
// lower half of another, so ONE swap and ONE add turn two values into one register whose halves carry one value each (summed
// over lane pairs l, l + 32); v_permlane16_swap does the same with the 16-lane rows. After the two stages ceil(NV / 4)
// registers hold four values each — one per row — and four DPP steps finish the sums inside the rows: for 27 values 14 + 7
// swaps, 21 + 28 adds instead of the 162 DPP adds of wave_sum_all (r02: 1.2 us per launch, at the tail of every launch).
// Row q of out[j] holds value 4 j + 2 (q & 1) + (q >> 1), the same total in every lane of the row. Fixed order: deterministic.
template <int NV>
struct WaveRows {
  static constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2;
};
template <int NV>
__device__ __forceinline__ void wave_sum_rows(const float (&v)[NV], float (&out)[WaveRows<NV>::N2]) {
  constexpr int N1 = WaveRows<NV>::N1, N2 = WaveRows<NV>::N2;
  float s1[N1];
#pragma unroll
  for (int j = 0; j < N1; j++) {
    const float a = v[2 * j], b = (2 * j + 1 < NV) ? v[2 * j + 1] : 0.0f;   // (an odd value out pairs with zero: a register swapped with itself is a no-op)
    const auto t = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned t0 = t[0], t1 = t[1];   // (copied to scalars first: __builtin_bit_cast of a vector element expression reads element 0)
    s1[j] = __builtin_bit_cast(float, t0) + __builtin_bit_cast(float, t1);
  }
#pragma unroll
  for (int j = 0; j < N2; j++) {
    const float a = s1[2 * j], b = (2 * j + 1 < N1) ? s1[2 * j + 1] : 0.0f;
    const auto t = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned t0 = t[0], t1 = t[1];
    out[j] = __builtin_bit_cast(float, t0) + __builtin_bit_cast(float, t1);
  }
#pragma unroll
  for (int j = 0; j < N2; j++) out[j] = dpp_add<0xB1, 0xf>(out[j]);    // quad_perm [1,0,3,2]
#pragma unroll
  for (int j = 0; j < N2; j++) out[j] = dpp_add<0x4E, 0xf>(out[j]);    // quad_perm [2,3,0,1]
#pragma unroll
  for (int j = 0; j < N2; j++) out[j] = dpp_add<0x141, 0xf>(out[j]);   // row_half_mirror
#pragma unroll
  for (int j = 0; j < N2; j++) out[j] = dpp_add<0x140, 0xf>(out[j]);   // row_mirror: every lane of a row holds the row's total
}

// block reduction of NV per-thread accumulators; thread 0..NV-1 of the block ends up writing value j.
template <int NV>
__device__ __forceinline__ void block_reduce_store(float (&acc)[NV], float* __restrict__ out) {
  __shared__ float red[ELLC_GN_THREADS / 64][32];
  static_assert(NV <= 28, "one partial record holds 32 floats");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float rows[WaveRows<NV>::N2];
  wave_sum_rows<NV>(acc, rows);
  if ((lane & 15) == 0) {   // the first lane of each row stores the row's value of every register
    const int q = lane >> 4, col = 2 * (q & 1) + (q >> 1);
#pragma unroll
    for (int j = 0; j < WaveRows<NV>::N2; j++) red[wave][4 * j + col] = rows[j];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float s = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < ELLC_GN_THREADS / 64; w++) s += red[w][threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// One pixel of the FCA pass (PixelWisePyramid.cpp:236-361): J, residual, weight.
struct FcaPix { float J[6]; float residual, wgt; };

// one entry of the keyframe's compact pixel list as the exact pass works on it (FcaRec, with the back-projection formed again)
struct FcaIn { uint32_t xy; float Z, var, Ikf, X, Y; double invZ; };   // xy: y << 16 | x

template <bool DIVC>
__device__ __forceinline__ FcaIn fca_load(const KfLevelDev& K, const LevelGeom& g, unsigned i) {
  // uniform base + 32-bit byte offset (20 * i < 2^32 for any image this library accepts): SGPR-base addressing; 16 + 4 bytes
  typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
  const ELLC_GLOBAL char* r = (const ELLC_GLOBAL char*)K.crec + i * (unsigned)sizeof(FcaRec);
  const u32x4a v = *(const ELLC_GLOBAL u32x4a*)r;
  const uint32_t hi = *(const ELLC_GLOBAL uint32_t*)(r + 16);
  const uint32_t w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;   // (copied to scalars first: bit_cast of a vector element expression reads element 0)
  FcaIn in;
  const int x = (int)(w0 & 0xfffu), y = (int)((w0 >> 12) & 0xfffu);
  in.xy = ((uint32_t)y << 16) | (uint32_t)x;
  in.Ikf = byte_f32<3>(w0);
  in.Z = __builtin_bit_cast(float, w1);
  in.var = __builtin_bit_cast(float, w2);
  in.invZ = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)w3);
  // PixelWisePyramid.cpp:236-240, the expressions prep_scatter used to store: ((x - cx) Z) / fx
  const float aX = ((float)x - g.cx) * in.Z, aY = ((float)y - g.cy) * in.Z;
  in.X = DIVC ? div_const(aX, g.fx, g.rfx) : aX / g.fx;
  in.Y = DIVC ? div_const(aY, g.fy, g.rfy) : aY / g.fy;
  return in;
}
__device__ __forceinline__ FcaIn fca_in_empty() {
  FcaIn in;
  in.xy = 0; in.Z = 1.0f; in.var = 0.0f; in.Ikf = 0.0f; in.X = 0.0f; in.Y = 0.0f; in.invZ = 1.0;
  return in;
}

// Everything the Jacobian and the weight need from the keyframe pixel alone (no pose, no tap): the per-column /
// per-row table entries and the products with 1/Z of PixelWisePyramid.cpp:296-303. The fused kernel evaluates this
// for a thread's first pixel while the solve of the previous iteration is still running.
struct FcaPre {
  float c_t0, c_b1, colB, rowB, d;
  double colA, rowA, fxz, fyz, nvz, nuz;
};

template <bool DIVC>
__device__ __forceinline__ FcaPre fca_prepare(const LevelGeom& g, const FcaIn& in) {
  const int x = (int)(in.xy & 0xffffu), y = (int)(in.xy >> 16);
  const float u = -g.cx + (float)x;
  const float v = -g.cy + (float)y;
  const float vu = v * u;
  FcaPre p;
  p.c_t0 = DIVC ? div_const(-vu, g.fy, g.rfy) : (-vu / g.fy);
  p.c_b1 = DIVC ? div_const(vu, g.fx, g.rfx) : (vu / g.fx);
  p.colA = as_global(g.colA)[x];
  p.colB = as_global(g.colB)[x];
  p.rowA = as_global(g.rowA)[y];
  p.rowB = as_global(g.rowB)[y];
  p.fxz = (double)g.fx * in.invZ;
  p.fyz = (double)g.fy * in.invZ;
  p.nvz = (double)(-v) * in.invZ;
  p.nuz = (double)(-u) * in.invZ;
  p.d = (float)in.invZ;
  return p;
}

// PixelWisePyramid.cpp:296-320 with the pose-independent factors of fca_prepare (same expressions as jacobian_row)
__device__ __forceinline__ void jacobian_row_pre(float gradx, float grady, const FcaPre& p, float J[6]) {
  const float jb0 = (float)((double)grady * p.rowA);
  const float jt0 = gradx * p.c_t0;
  const float jb1 = grady * p.c_b1;
  const float jt1 = (float)((double)gradx * p.colA);
  const float jb2 = grady * p.colB;
  const float jt2 = gradx * p.rowB;
  const float jt3 = (float)((double)gradx * p.fxz);
  const float jb4 = (float)((double)grady * p.fyz);
  const float jb5 = (float)((double)grady * p.nvz);
  const float jt5 = (float)((double)gradx * p.nuz);
  J[0] = jt0 + jb0;
  J[1] = jt1 + jb1;
  J[2] = jt2 + jb2;
  J[3] = jt3 + 0.0f;
  J[4] = 0.0f + jb4;
  J[5] = jt5 + jb5;
}

template <bool DEBUG, class PF = NoPrefetch>
__device__ __forceinline__ FcaPix fca_pixel_pre(const GnArgs& a, const KfLevelDev& K, const LevelGeom& g, g_u8 cur,
                                                const float* S, unsigned i, const FcaIn& in, const FcaPre& pre, PF pf = PF()) {
  const Warp w = warp_point<true>(in.X, in.Y, in.Z, g, S);
  const Taps t = tap_point<true, false, PF>(cur, g.sw, g.cols, g.rows, w.wx, w.wy, pf);
  FcaPix o;
  jacobian_row_pre(t.gx, t.gy, pre, o.J);
  const bool oob = (t.I == -1.0f);
  o.residual = oob ? 0.0f : (t.I - in.Ikf);
  o.wgt = oob ? 0.0f : fca_weight(w, pre.d, o.residual, t.gx, t.gy, 1.0f * in.var, g, S[3], S[7], S[11]);
  if (a.save_w) *(ELLC_GLOBAL float*)((ELLC_GLOBAL char*)K.wlast + i * 4u) = o.wgt;
  if (DEBUG) {
    const int x = (int)(in.xy & 0xffffu), y = (int)(in.xy >> 16);
    const size_t n = (size_t)g.n, p = (size_t)y * g.cols + x;
    a.planes[0 * n + p] = o.residual;
    a.planes[1 * n + p] = o.wgt;
    a.planes[2 * n + p] = oob ? -1.0f : w.wx;
    a.planes[3 * n + p] = oob ? -1.0f : w.wy;
#pragma unroll
    for (int k = 0; k < 6; k++) a.planes[(4 + k) * n + p] = o.J[k];
  }
  return o;
}

template <bool DEBUG, bool DIVC, class PF = NoPrefetch>
__device__ __forceinline__ FcaPix fca_pixel_in(const GnArgs& a, const KfLevelDev& K, const LevelGeom& g, g_u8 cur,
                                               const float* S, unsigned i, const FcaIn& in, PF pf = PF()) {
  return fca_pixel_pre<DEBUG, PF>(a, K, g, cur, S, i, in, fca_prepare<DIVC>(g, in), pf);
}

template <bool DEBUG, bool DIVC>
__device__ __forceinline__ FcaPix fca_pixel(const GnArgs& a, const KfLevelDev& K, const LevelGeom& g, g_u8 cur,
                                            const float* S, unsigned i) {
  return fca_pixel_in<DEBUG, DIVC>(a, K, g, cur, S, i, fca_load<DIVC>(K, g, i));
}

// ---------------------------------------------------------------------------------------------------
// Tolerance-mode pixel pass (cfg.arith = ELLC_ARITH_FAST). Same formulas as PixelWisePyramid.cpp:236-361, evaluated in f32
// with fused multiply-adds and the hardware reciprocal / reciprocal square root (1 ulp) in place of the IEEE division and
// sqrt sequences, and f32 products where the reference's pow() promotes to double. Per-pixel values agree with the
// exact path to a few 1e-7 relative (tests/test_gpu_fast.py states the bounds); the final pose to well below the 1e-5 bar.
// 12-byte records (FcaRecF, ellc_device.hpp; r05); r04: written for the issue classes of the vector ALU (see tap_request_f).
// (a record slot is kept as ONE vector value: it is carried around the pixel loop while its load is in flight, and a slot made of
// scalars makes the register allocator copy them at the loop's back edge — copies that wait for the load)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
struct FcaInF { u32x4_t v; };   // FcaRecF in registers: x | y << 12 | I << 24, variance, d = 1 / Z (the fourth word is unused)
__device__ __forceinline__ FcaInF fcaf_empty() { FcaInF in; in.v = (u32x4_t){0u, 0u, 0x3f800000u, 0u}; return in; }   // x = y = 0, I = 0, var = 0, d = 1

#define ELLC_FREC 12u
typedef uint32_t Rec12 __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ FcaInF fcaf_load_off(const KfLevelDev& K, unsigned byte_off) {
  FcaInF in;
  const Rec12 r = *(const ELLC_GLOBAL Rec12*)((const ELLC_GLOBAL char*)K.crec + byte_off);
  const uint32_t a = r.x, b = r.y, c = r.z;
  in.v = (u32x4_t){a, b, c, 0u};
  return in;
}
__device__ __forceinline__ FcaInF fcaf_load(const KfLevelDev& K, unsigned i) { return fcaf_load_off(K, i * ELLC_FREC); }
// position of a tolerance-mode record
__device__ __forceinline__ void fcaf_position_at(const KfLevelDev& K, const LevelGeom& g, int i, int& x, int& y) {
  const uint32_t w = *(const ELLC_GLOBAL uint32_t*)((const ELLC_GLOBAL char*)K.crec + (unsigned)i * 12u);
  x = (int)(w & 0xfffu); y = (int)((w >> 12) & 0xfffu);
}

// What the pixel step needs of the level and of the pose, in VECTOR registers (an SGPR operand halves the issue rate of the f32
// multiply-adds, tools/micro/valu_rates.hip). P = K S with K = [fx 0 cx; 0 fy cy; 0 0 1] and S = exp(pose) (3 x 4): the warped point
// comes out in pixel units times its depth, so the projection is one multiplication per coordinate.
struct FcafConst {
  float P[12];
  float rfy, qc;     // q = (y - cy) / fy = y rfy + qc
  float rfx, pc;     // p = (x - cx) / fx = x rfx + pc
  float hfx, hfy;    // fx / 2, fy / 2 (the taps return twice the gradient)
};
__device__ __forceinline__ FcafConst fcaf_const(const LevelGeom& g, const float* S) {
  FcafConst c;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    c.P[k] = __builtin_fmaf(g.fx, S[k], g.cx * S[8 + k]);
    c.P[4 + k] = __builtin_fmaf(g.fy, S[4 + k], g.cy * S[8 + k]);
    c.P[8 + k] = S[8 + k];
  }
  c.rfy = g.rfy; c.qc = -(g.cy * g.rfy);
  c.rfx = g.rfx; c.pc = -(g.cx * g.rfx);
  asm volatile("" : "+v"(c.rfx), "+v"(c.pc));
  c.hfx = 0.5f * g.fx; c.hfy = 0.5f * g.fy;
#pragma unroll
  for (int k = 0; k < 12; k++) asm volatile("" : "+v"(c.P[k]));
  asm volatile("" : "+v"(c.rfy), "+v"(c.qc), "+v"(c.hfx), "+v"(c.hfy));
  return c;
}

// The pixel step in two stages (see tap_request_f): stage A decodes the record, warps the point and requests its rows; stage B
// interpolates, forms the Jacobian row and the weight. FcafStage is what B needs of A.
struct FcafStage {
  TapReq tq;
  float x1, y1;              // warped position in the current image
  float p, q, d, var, Ikf;
  float px, py, pz, rz;
};
template <class PF = NoPrefetch>
__device__ __forceinline__ FcafStage fcaf_stage_a(const LevelGeom& g, const TapRows& tr, const FcafConst& c, const FcaInF& in, PF pf = PF()) {
  FcafStage s;
  // (elements are copied to scalars first: __builtin_bit_cast applied to a vector element expression reads element 0)
  const uint32_t w0 = in.v.x, w1 = in.v.y, w2 = in.v.z;
  const float xf = (float)(w0 & 0xfffu), yf = (float)((w0 >> 12) & 0xfffu);
  s.Ikf = cvt_ubyte<3>(w0);
  s.var = __builtin_bit_cast(float, w1); s.d = __builtin_bit_cast(float, w2);
  s.p = __builtin_fmaf(xf, c.rfx, c.pc);
  s.q = __builtin_fmaf(yf, c.rfy, c.qc);   // p = u / fx, q = v / fy
  // K ((p, q, 1) + t d): the warped point divided by the keyframe depth Z, in pixel units times its own depth. The factors of Z
  // cancel in the projection and in the weight (1 / (pz^2 d) = Z rz^2 with the true pz; here pz is pz / Z).
  s.px = __builtin_fmaf(c.P[0], s.p, __builtin_fmaf(c.P[1], s.q, __builtin_fmaf(c.P[3], s.d, c.P[2])));
  s.py = __builtin_fmaf(c.P[4], s.p, __builtin_fmaf(c.P[5], s.q, __builtin_fmaf(c.P[7], s.d, c.P[6])));
  s.pz = __builtin_fmaf(c.P[8], s.p, __builtin_fmaf(c.P[9], s.q, __builtin_fmaf(c.P[11], s.d, c.P[10])));
  // no clamp of pz away from zero (ExternVariable.h:232): 1/0 = inf sends the point out of bounds, as the clamped value does
  s.rz = __builtin_amdgcn_rcpf(s.pz);
  s.x1 = s.px * s.rz; s.y1 = s.py * s.rz;
  s.tq = tap_request_f<true, PF>(tr, g.sw, g.cols, g.rows, s.x1, s.y1, pf);
  return s;
}
// SAVEW: 1 = store the weights (saved-weights call), 0 = do not, -1 = a.save_w decides at run time
template <bool DEBUG, int SAVEW = -1, bool LAT = false>
__device__ __forceinline__ FcaPix fcaf_stage_b(const GnArgs& a, const KfLevelDev& K, const LevelGeom& g, g_u8 cur, const FcafConst& c, unsigned i,
                                               const FcafStage& s) {
  const Taps t = tap_finish_f<true, LAT>(s.tq, cur, g.sw, g.cols, g.rows, s.x1, s.y1);
  const float p = s.p, q = s.q, d = s.d;
  FcaPix o;
  // 1x6 row (:296-320) with A = fx gradx, B = fy grady, T = A p + B q:
  //   J = [-(q T + B), p T + A, B p - A q, A d, B d, -d T]
  // (the taps return twice the gradients.) Entries 0 and 5 are carried with the opposite sign — the accumulators then hold
  // sign-flipped sums, exactly (rounding is symmetric), and fca_acc_unpack<true> flips them back
  const float A = c.hfx * t.gx, B = c.hfy * t.gy;
  const float T = __builtin_fmaf(A, p, B * q);
  o.J[0] = __builtin_fmaf(q, T, B);   // -J[0]
  o.J[1] = __builtin_fmaf(p, T, A);
  o.J[2] = __builtin_fmaf(B, p, -(A * q));
  o.J[3] = A * d;
  o.J[4] = B * d;
  o.J[5] = d * T;                     // -J[5]
  const float res = t.I - s.Ikf;
  // weight (:341-358): w_p = 1 / D, sqrt(w_p) = rsq(D);  Huber: w_p below the knee (|r| sqrt(w_p) < 1.5), 1.5 sqrt(w_p) / |r|
  // above it — the smaller of the two. With the point in pixel units, t' = K t:  fx (tx pz - tz px) = t'x pz - tz px', so
  // drpdd = (A n0 + B n1) rz^2 = (gx2 n0' + gy2 n1') rz^2 / 2  (gx2, gy2: twice the gradients)
  const float n0 = __builtin_fmaf(c.P[3], s.pz, -(c.P[11] * s.px)), n1 = __builtin_fmaf(c.P[7], s.pz, -(c.P[11] * s.py));
  const float drpdd = __builtin_fmaf(t.gx, n0, t.gy * n1) * (0.5f * (s.rz * s.rz));
  const float D = __builtin_fmaf(s.var * drpdd, drpdd, 16.0f);
  const float r = __builtin_amdgcn_rsqf(D);
  // r min(r, k) = min(r r, r k), the middle one of (0, r r, r k); a NaN (a point at pz = 0: it is out of bounds, its J is zero, and
  // NaN 0 would still poison the sums) comes out as 0: v_med3_f32 returns the minimum of the operands that are numbers
  const float wgt = __builtin_amdgcn_fmed3f(r * r, r * (1.5f * __builtin_amdgcn_rcpf(fabsf(res))), 0.0f);
  // out of bounds (I = -1: all four taps outside): the gradients, hence J, are 0 and the residual is finite, so every sum gets 0
  // whatever the weight; the weight itself must read 0 only where it is stored
  const bool oob = (t.I == -1.0f);
  o.residual = DEBUG ? (oob ? 0.0f : res) : res;
  if (DEBUG || SAVEW != 0) o.wgt = oob ? 0.0f : wgt;
  else o.wgt = wgt;
  if (SAVEW > 0 || (SAVEW < 0 && a.save_w)) *(ELLC_GLOBAL float*)((ELLC_GLOBAL char*)K.wlast + i * 4u) = o.wgt;
  if (DEBUG) {
    const int y = (int)rintf(__builtin_fmaf(q, g.fy, g.cy)), x = (int)rintf(__builtin_fmaf(p, g.fx, g.cx));
    const size_t n = (size_t)g.n, pp = (size_t)y * g.cols + x;
    a.planes[0 * n + pp] = o.residual;
    a.planes[1 * n + pp] = o.wgt;
    a.planes[2 * n + pp] = oob ? -1.0f : s.x1;
    a.planes[3 * n + pp] = oob ? -1.0f : s.y1;
#pragma unroll
    for (int k = 0; k < 6; k++) a.planes[(4 + k) * n + pp] = (k == 0 || k == 5) ? -o.J[k] : o.J[k];
  }
  return o;
}
template <bool DEBUG, int SAVEW = -1>
__device__ __forceinline__ FcaPix fcaf_pixel(const GnArgs& a, const KfLevelDev& K, const LevelGeom& g, g_u8 cur, const TapRows& tr, const FcafConst& c,
                                             unsigned i, const FcaInF& in) {
  return fcaf_stage_b<DEBUG, SAVEW>(a, K, g, cur, c, i, fcaf_stage_a(g, tr, c, in));
}

// H += (w J)^T J (upper triangle), b += J (r w)   (PixelWisePyramid.cpp:364-374)
// The accumulators are kept as register pairs so that every update is one packed fma (v_pk_fma_f32) of a J pair with a
// broadcast (w J[r]): rows 1, 3 and 5 carry one unused lane each (the element left of the diagonal) to keep the J pairs
// aligned. Each of the 27 used lanes still performs exactly acc = fma(J[r] * w, J[c], acc) per pixel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct FcaAcc {
  f32x2 h[12];   // (00,01)(02,03)(04,05) (10*,11)(12,13)(14,15) (22,23)(24,25) (32*,33)(34,35) (44,45) (54*,55)
  f32x2 b[3];
};
__device__ __forceinline__ void fca_acc_zero(FcaAcc& A) {
#pragma unroll
  for (int i = 0; i < 12; i++) A.h[i] = (f32x2)(0.0f);
#pragma unroll
  for (int i = 0; i < 3; i++) A.b[i] = (f32x2)(0.0f);
}
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void fca_accumulate_pixel(FcaAcc& A, const FcaPix& p) {
  const f32x2 J01 = {p.J[0], p.J[1]}, J23 = {p.J[2], p.J[3]}, J45 = {p.J[4], p.J[5]};
  const f32x2 w2 = (f32x2)(p.wgt);
  const f32x2 wJ01 = J01 * w2, wJ23 = J23 * w2, wJ45 = J45 * w2;
  A.h[0] = fma2((f32x2)(wJ01.x), J01, A.h[0]);
  A.h[1] = fma2((f32x2)(wJ01.x), J23, A.h[1]);
  A.h[2] = fma2((f32x2)(wJ01.x), J45, A.h[2]);
  A.h[3] = fma2((f32x2)(wJ01.y), J01, A.h[3]);
  A.h[4] = fma2((f32x2)(wJ01.y), J23, A.h[4]);
  A.h[5] = fma2((f32x2)(wJ01.y), J45, A.h[5]);
  A.h[6] = fma2((f32x2)(wJ23.x), J23, A.h[6]);
  A.h[7] = fma2((f32x2)(wJ23.x), J45, A.h[7]);
  A.h[8] = fma2((f32x2)(wJ23.y), J23, A.h[8]);
  A.h[9] = fma2((f32x2)(wJ23.y), J45, A.h[9]);
  A.h[10] = fma2((f32x2)(wJ45.x), J45, A.h[10]);
  A.h[11] = fma2((f32x2)(wJ45.y), J45, A.h[11]);
  const f32x2 rw = (f32x2)(p.residual * p.wgt);
  A.b[0] = fma2(J01, rw, A.b[0]);
  A.b[1] = fma2(J23, rw, A.b[1]);
  A.b[2] = fma2(J45, rw, A.b[2]);
}
// the 27 sums in the order of the partial record: upper triangle by rows, then b
// FLIP: the pixel pass carried J[0] and J[5] with the opposite sign (fcaf_pixel): H(0,1..4), H(1..4,5), b[0], b[5] change sign
template <bool FLIP = false>
__device__ __forceinline__ void fca_acc_unpack(const FcaAcc& A, float (&o)[27]) {
  o[0] = A.h[0].x; o[1] = A.h[0].y; o[2] = A.h[1].x; o[3] = A.h[1].y; o[4] = A.h[2].x; o[5] = A.h[2].y;
  o[6] = A.h[3].y; o[7] = A.h[4].x; o[8] = A.h[4].y; o[9] = A.h[5].x; o[10] = A.h[5].y;
  o[11] = A.h[6].x; o[12] = A.h[6].y; o[13] = A.h[7].x; o[14] = A.h[7].y;
  o[15] = A.h[8].y; o[16] = A.h[9].x; o[17] = A.h[9].y;
  o[18] = A.h[10].x; o[19] = A.h[10].y;
  o[20] = A.h[11].y;
  o[21] = A.b[0].x; o[22] = A.b[0].y; o[23] = A.b[1].x; o[24] = A.b[1].y; o[25] = A.b[2].x; o[26] = A.b[2].y;
  if (FLIP) {
    o[1] = -o[1]; o[2] = -o[2]; o[3] = -o[3]; o[4] = -o[4];
    o[10] = -o[10]; o[14] = -o[14]; o[17] = -o[17]; o[19] = -o[19];
    o[21] = -o[21]; o[26] = -o[26];
  }
}

// FCA accumulate without the folded solve (single-step API, debug planes, ELLC_NO_FUSE): grid (nblk, B). Each block
// owns a contiguous chunk of the alignment's compact pixel list and writes one 27-float partial record.
template <bool DEBUG, bool DIVC, bool FAST = false>
__global__ __launch_bounds__(ELLC_GN_THREADS) void gn_fca_accumulate(GnArgs a) {
  const int b = blockIdx.y;
  const AlignState& st = a.state[b];
  if (st.level_done == a.level) return;
  const LevelGeom g = a.geom[a.level];
  const KfLevelDev K = a.kf_tab[a.level * a.max_kf + a.kf_slot[b]];   // by value: uniform, lives in SGPRs
  const FrLevelDev& F = a.fr_tab[a.level * a.max_fr + a.fr_slot[b]];
  const int V = *K.count;
  const int chunk = (V + a.nblk - 1) / a.nblk;
  const int begin = blockIdx.x * chunk;
  const int end = min(V, begin + chunk);
  float S[12];
#pragma unroll
  for (int i = 0; i < 12; i++) S[i] = st.S[i];
  g_u8 cur = as_global(F.img);
  FcaAcc acc;
  fca_acc_zero(acc);
  const TapRows tr = tap_rows(cur, g.sw);
  const FcafConst fc = fcaf_const(g, S);
  for (int i = begin + (int)threadIdx.x; i < end; i += ELLC_GN_THREADS) {
    FcaPix p;
    if constexpr (FAST) p = fcaf_pixel<DEBUG>(a, K, g, cur, tr, fc, (unsigned)i, fcaf_load(K, (unsigned)i));
    else p = fca_pixel<DEBUG, DIVC>(a, K, g, cur, S, i);
    fca_accumulate_pixel(acc, p);
  }
  float sums[27];
  fca_acc_unpack<FAST>(acc, sums);
  block_reduce_store<27>(sums, a.partials + ((size_t)b * ELLC_NBLK_MAX + blockIdx.x) * ELLC_PART_STRIDE);
}

// ---------------------------------------------------------------------------------------------------
// ICA precompute (PixelWisePyramid.cpp:561-680 + H = WSD*SD^T :938): template-gradient Jacobian at the
// integer pixel, stored as 6 compact planes; H partials with the constant weights.
__global__ __launch_bounds__(ELLC_GN_THREADS) void gn_ica_precompute(GnArgs a, int cap) {
  const int b = blockIdx.y;
  const AlignState& st = a.state[b];
  if (st.level_done == a.level) return;
  const LevelGeom g = a.geom[a.level];
  const KfLevelDev K = a.kf_tab[a.level * a.max_kf + a.kf_slot[b]];   // by value: uniform, lives in SGPRs
  const int V = *K.count;
  // contiguous chunk per block (keeps a block's taps in a few image rows: 25 % less fetch traffic than a tile-cyclic
  // split, which was tried in r01 and did not change the run time — co-resident blocks finish staggered because the
  // SIMD arbiter serves the oldest wave first, not because their pixels differ)
  const int chunk = (V + a.nblk - 1) / a.nblk;
  const int begin = blockIdx.x * chunk;
  const int end = min(V, begin + chunk);
  const int stride = ELLC_GN_THREADS;
  g_u8 img = as_global(K.img);
  float acc[21];
#pragma unroll
  for (int i = 0; i < 21; i++) acc[i] = 0.0f;
  for (int i = begin + (int)threadIdx.x; i < end; i += stride) {
    const uint32_t xy = as_global(K.cxy)[i];
    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
    const float Z = as_global(K.cZ)[i];
    // frame::calculateGradient of the keyframe level image at (y,x)  (Frame.cpp:185-285)
    const int xm = clampi(x - 1, 0, g.cols - 1), xp = clampi(x + 1, 0, g.cols - 1);
    const int ym = clampi(y - 1, 0, g.rows - 1), yp = clampi(y + 1, 0, g.rows - 1);
    const float sx = (x == 0 || x == g.cols - 1) ? 1.0f : 0.5f;
    const float sy = (y == 0 || y == g.rows - 1) ? 1.0f : 0.5f;
    const float gradx = sx * ((float)img[(unsigned)(y * g.sw + xp)] - (float)img[(unsigned)(y * g.sw + xm)]);
    const float grady = sy * ((float)img[(unsigned)(yp * g.sw + x)] - (float)img[(unsigned)(ym * g.sw + x)]);
    float J[6];
    jacobian_row<false>(gradx, grady, x, y, 1.0 / (double)Z, g, J);
    const float wgt = as_global(K.cW)[i];
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      as_global_rw(K.sd)[(size_t)r * cap + i] = J[r];
      const float wJ = J[r] * wgt;   // weightedSteepestDescent (:664-669)
#pragma unroll
      for (int c = r; c < 6; c++) { acc[q] = __builtin_fmaf(wJ, J[c], acc[q]); q++; }
    }
  }
  float acc27[27];
#pragma unroll
  for (int i = 0; i < 21; i++) acc27[i] = acc[i];
#pragma unroll
  for (int i = 21; i < 27; i++) acc27[i] = 0.0f;
  block_reduce_store<27>(acc27, a.partials + ((size_t)b * ELLC_NBLK_MAX + blockIdx.x) * ELLC_PART_STRIDE);
}

// ICA iterate (PixelWisePyramid.cpp:687-913): warp, u8 tap, residual, b += SD * (r * w)
template <bool DEBUG>
__global__ __launch_bounds__(ELLC_GN_THREADS) void gn_ica_iterate(GnArgs a, int cap) {
  const int b = blockIdx.y;
  const AlignState& st = a.state[b];
  if (st.level_done == a.level) return;
  const LevelGeom g = a.geom[a.level];
  const KfLevelDev K = a.kf_tab[a.level * a.max_kf + a.kf_slot[b]];   // by value: uniform, lives in SGPRs
  const FrLevelDev& F = a.fr_tab[a.level * a.max_fr + a.fr_slot[b]];
  const int V = *K.count;
  // contiguous chunk per block (keeps a block's taps in a few image rows: 25 % less fetch traffic than a tile-cyclic
  // split, which was tried in r01 and did not change the run time — co-resident blocks finish staggered because the
  // SIMD arbiter serves the oldest wave first, not because their pixels differ)
  const int chunk = (V + a.nblk - 1) / a.nblk;
  const int begin = blockIdx.x * chunk;
  const int end = min(V, begin + chunk);
  const int stride = ELLC_GN_THREADS;
  float S[12];
#pragma unroll
  for (int i = 0; i < 12; i++) S[i] = st.S[i];
  g_u8 cur = as_global(F.img);
  g_f32 sd = as_global(K.sd);
  float acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0.0f;
  for (int i = begin + (int)threadIdx.x; i < end; i += stride) {
    const uint32_t xy = as_global(K.cxy)[i];
    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
    const float Z = as_global(K.cZ)[i];
    const Warp w = warp_pixel<false>(x, y, Z, g, S);
    const Taps t = tap_point<false>(cur, g.sw, g.cols, g.rows, w.wx, w.wy);
    const bool oob = (t.I == -1.0f);
    const float residual = oob ? 0.0f : (t.I - as_global(K.cI)[i]);
    const float wgt = as_global(K.cW)[i];
    const float rw = residual * wgt;
    if (DEBUG) {
      const size_t n = (size_t)g.n, p = (size_t)y * g.cols + x;
      a.planes[0 * n + p] = residual;
      a.planes[1 * n + p] = wgt;
      a.planes[2 * n + p] = oob ? -1.0f : w.wx;
      a.planes[3 * n + p] = oob ? -1.0f : w.wy;
#pragma unroll
      for (int k = 0; k < 6; k++) a.planes[(4 + k) * n + p] = sd[(size_t)k * cap + i];
    }
#pragma unroll
    for (int r = 0; r < 6; r++) acc[21 + r] = __builtin_fmaf(sd[(size_t)r * cap + i], rw, acc[21 + r]);
  }
  block_reduce_store<27>(acc, a.partials + ((size_t)b * ELLC_NBLK_MAX + blockIdx.x) * ELLC_PART_STRIDE);
}

// ---------------------------------------------------------------------------------------------------
// cv::Mat::inv(DECOMP_LU) on a 6x6 f32 matrix (OpenCV 3.0.0 LUImpl on (A | I), partial pivoting,
// |pivot| < FLT_EPSILON => singular => all-zero inverse; PixelWisePyramid.cpp:451).
// Lane j (0..5) of a wave carries right-hand-side column j; every lane eliminates its own copy of A, so
// the rounding sequence per entry is the scalar algorithm's. All indices are compile-time (registers only).
// On return x[i] = Hinv[i][lane].
__device__ __forceinline__ void lu_inverse6_lanes(float (&A)[36], int lane, float (&x)[6]) {   // A is destroyed
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] = (i == lane) ? 1.0f : 0.0f;
  bool singular = false;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    int k = i;
    float best = fabsf(A[i * 6 + i]);
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const float v = fabsf(A[j * 6 + i]);
      if (v > best) { best = v; k = j; }
    }
    if (best < 1.1920928955078125e-07f) singular = true;
    if (k != i) {   // every lane eliminates the same matrix, so the branch is wave-uniform; most pivots sit on the diagonal
#pragma unroll
      for (int j = i + 1; j < 6; j++) {
        const bool sw = (k == j);
#pragma unroll
        for (int q = i; q < 6; q++) {
          const float a = A[i * 6 + q], b = A[j * 6 + q];
          A[i * 6 + q] = sw ? b : a;
          A[j * 6 + q] = sw ? a : b;
        }
        const float a = x[i], b = x[j];
        x[i] = sw ? b : a;
        x[j] = sw ? a : b;
      }
    }
    const float d = -1.0f / A[i * 6 + i];
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const float alpha = A[j * 6 + i] * d;
#pragma unroll
      for (int q = i + 1; q < 6; q++) A[j * 6 + q] += alpha * A[i * 6 + q];
      x[j] += alpha * x[i];
    }
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float s = x[i];
#pragma unroll
    for (int q = i + 1; q < 6; q++) s -= A[i * 6 + q] * x[q];
    x[i] = s / A[i * 6 + i];
  }
  if (singular) {
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------------
// Tolerance-mode solve (cfg.arith = ELLC_ARITH_FAST). The normal equations H delta = -b are solved directly by an
// L D L^T factorisation in double (H is a sum of w J^T J, symmetric positive semi-definite) instead of forming
// cv::Mat::inv(DECOMP_LU) in f32 and multiplying: more accurate than the reference's own update and a much shorter
// dependent chain (every lane runs the whole 6x6 factorisation redundantly on registers; no cross-lane step). A pivot
// below FLT_EPSILON gives the zero update, as the reference's singular inverse does (PixelWisePyramid.cpp:451).
// sums: 21 upper-triangular entries by rows, then b.
__device__ __forceinline__ double rcp_f64(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ void ldlt_solve6(const double* sums, double (&x)[6]) {
  double A[6][6];   // lower triangle used
  {
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = r; c < 6; c++) A[c][r] = sums[q++];
  }
  double y[6], dinv[6];
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = -sums[21 + i];
  bool singular = false;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const double dj = A[j][j];
    if (!(dj >= 1.1920928955078125e-07)) singular = true;
    const double inv = rcp_f64(dj);
    dinv[j] = inv;
    double col[6];   // column j below the diagonal, still scaled: L_ij d_j
#pragma unroll
    for (int i = j + 1; i < 6; i++) col[i] = A[i][j];
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      const double l = col[i] * inv;
      A[i][j] = l;
#pragma unroll
      for (int k = j + 1; k <= i; k++) A[i][k] = __builtin_fma(-l, col[k], A[i][k]);
      y[i] = __builtin_fma(-l, y[j], y[i]);
    }
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double v = y[i] * dinv[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) v = __builtin_fma(-A[k][i], x[k], v);
    x[i] = v;
  }
  if (singular) {
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = 0.0;
  }
}

struct SolveArgs {
  AlignState* state;
  const float* partials;
  int level, nblk;
  int mode;         // 0 FCA: H and b from partials; 1 ICA-precompute: H only (stores Hinv); 2 ICA-iterate: b only
  int early_exit;
};

#define ELLC_SOLVE_THREADS 256

struct SolveShared {
  double part[ELLC_SOLVE_THREADS / 32][32];
  double sums[32];
  double prod[36];
  float Hinv[36];
  float delta[6];
  float newpose[6];
  float newS[12];
  float weighted;
  int level_done;
};

// The solve of one Gauss-Newton iteration for one alignment, executed by a whole 256-thread block
// (PixelWisePyramid.cpp:441-491): fixed-order f64 combine of the nblk block partials (all threads load in
// parallel), 6x6 LU inverse across six lanes, delta, weightedPose, pose <- log(exp(delta^) exp(pose^)) on one lane.
// Results are left in `sh` (new pose, exp(new pose), weightedPose, level_done) for every thread; when `dst` is
// non-null the state record is written too. src may alias dst. Ends with a block barrier.
// Thread t's share of the fixed-order combine: component t&31 over the block records k = t>>5, t>>5 + 8, ... < nblk,
// summed in double in ascending k. The loads depend on nothing but the kernel arguments, so they are issued eight at
// a time (missing records read as +0.0, which leaves the sum unchanged bit for bit) instead of one round trip each.
__device__ __forceinline__ double partial_group_sum(const float* __restrict__ partials, int nblk) {
  const int comp = threadIdx.x & 31, grp = threadIdx.x >> 5;
  g_f32 p = as_global(partials) + comp;
  double s = 0.0;
  for (int base = 0; base < nblk; base += 8 * (ELLC_SOLVE_THREADS / 32)) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {   // unconditional loads (index clamped to the last record), so that all eight are in flight
      const int k = base + grp + j * (ELLC_SOLVE_THREADS / 32);
      v[j] = p[(unsigned)min(k, nblk - 1) * ELLC_PART_STRIDE];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = base + grp + j * (ELLC_SOLVE_THREADS / 32);
      s += (k < nblk) ? (double)v[j] : 0.0;
    }
  }
  return s;
}

// partial_group_sum in two steps, for a launch that learns the number of pending records from the state record: the loads of
// the first 64 records (8 per thread, as above) are issued at once against an upper bound nload of the count, the sum is
// formed once the count nblk is known (records past it are ignored; past 64 they are loaded then). Same order, same bits.
__device__ __forceinline__ void partial_preload(const float* __restrict__ partials, int nload, float (&v)[8]) {
  const int comp = threadIdx.x & 31, grp = threadIdx.x >> 5;
  g_f32 p = as_global(partials) + comp;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = grp + j * (ELLC_SOLVE_THREADS / 32);
    v[j] = p[(unsigned)min(k, nload - 1) * ELLC_PART_STRIDE];
  }
}
__device__ __forceinline__ double partial_group_sum_from(const float* __restrict__ partials, const float (&v0)[8], int nblk) {
  const int comp = threadIdx.x & 31, grp = threadIdx.x >> 5;
  g_f32 p = as_global(partials) + comp;
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = grp + j * (ELLC_SOLVE_THREADS / 32);
    s += (k < nblk) ? (double)v0[j] : 0.0;
  }
  for (int base = 8 * (ELLC_SOLVE_THREADS / 32); base < nblk; base += 8 * (ELLC_SOLVE_THREADS / 32)) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = base + grp + j * (ELLC_SOLVE_THREADS / 32);
      v[j] = p[(unsigned)min(k, nblk - 1) * ELLC_PART_STRIDE];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = base + grp + j * (ELLC_SOLVE_THREADS / 32);
      s += (k < nblk) ? (double)v[j] : 0.0;
    }
  }
  return s;
}

// Second half of the solve: from the 27 combined sums in sh.sums to the new pose. S_cur / level_done_cur are the
// current exp(pose) and level_done (state record or LDS copy). Ends with a block barrier.
__device__ __forceinline__ void solve_finish(SolveShared& sh, int mode, int level, int early_exit, const AlignState& src,
                                             const float* S_cur, int level_done_cur, AlignState* dst) {
  const int t = threadIdx.x;
  if (t < 64) {   // wave 0 finishes the job; wave-level barriers only inside
    const int lane = t;
    const float S_lane = S_cur[min(lane, 11)];   // entry `lane` of the current exp(pose); consumed after the LU
    if (mode != 2) {
      float Hm[36];
      {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = r; c < 6; c++) {
            const float v = (float)sh.sums[q++];
            Hm[r * 6 + c] = v;
            Hm[c * 6 + r] = v;
          }
      }
      if (lane == 0 && dst) {
#pragma unroll
        for (int i = 0; i < 36; i++) dst->H[i] = Hm[i];
      }
      float x[6];
      lu_inverse6_lanes(Hm, lane < 6 ? lane : 0, x);
      if (lane < 6) {
#pragma unroll
        for (int i = 0; i < 6; i++) sh.Hinv[i * 6 + lane] = x[i];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    ELLC_STAMP(3);
    if (dst && lane < 36) dst->Hinv[lane] = sh.Hinv[lane];
    if (mode == 1) {
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) sh.newpose[i] = src.pose[i];
#pragma unroll
        for (int i = 0; i < 12; i++) sh.newS[i] = src.S[i];
        sh.weighted = src.weighted;
        sh.level_done = src.level_done;
      }
    } else {
      // delta = -(Hinv * b): cv::gemm f32 accumulates the products in double and rounds once
      if (lane < 36) {
        const int i = lane / 6, k = lane - 6 * i;
        sh.prod[lane] = (double)(float)sh.sums[21 + k] * (double)sh.Hinv[i * 6 + k];
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      if (lane < 6) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += sh.prod[lane * 6 + k];
        const float d = -(float)s;
        sh.delta[lane] = d;
        if (dst) {
          dst->delta[lane] = d;
          dst->b[lane] = (float)sh.sums[21 + lane];
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      ELLC_STAMP(4);
      {
        // pose <- log(exp(delta) * exp(pose)); exp(pose) is the f32 matrix the pixel pass used. The 64 lanes of the wave
        // share the work: lane 3r+k evaluates entry (r,k) of each exp (ellc_se3.hpp: exp_se3 is the same nine entry
        // evaluations in a loop), lane 4r+c entry (r,c) of the product; the log (series, a handful of terms) is evaluated
        // redundantly by every lane on broadcast inputs. Same operations per value as the scalar host functions.
        float delta[6];
#pragma unroll
        for (int i = 0; i < 6; i++) delta[i] = sh.delta[i];
        const float weighted = fabsf(delta[0] * 100000.0f) + fabsf(delta[1] * 100000.0f) + fabsf(delta[2] * 100000.0f) +
                               fabsf(delta[3] * 10000.0f) + fabsf(delta[4] * 10000.0f) + fabsf(delta[5] * 10000.0f);
        const int l9 = min(lane, 8);
        const int r3 = l9 / 3, k3 = l9 - 3 * r3;
        double Rrk, Vv;
        exp_se3_entry((double)delta[0], (double)delta[1], (double)delta[2], (double)delta[3], (double)delta[4], (double)delta[5], r3, k3, Rrk, Vv);
        const double trow = (Vv + __shfl_down(Vv, 1)) + __shfl_down(Vv, 2);   // t[r] in lanes 0, 3, 6
        const float Df = (float)Rrk, Dt = (float)trow;
        const int e = min(lane, 11), r = e >> 2, c = e & 3;
        const float d0 = __shfl(Df, 3 * r), d1 = __shfl(Df, 3 * r + 1), d2 = __shfl(Df, 3 * r + 2), d3 = __shfl(Dt, 3 * r);
        const float b0 = __shfl(S_lane, c), b1 = __shfl(S_lane, 4 + c), b2 = __shfl(S_lane, 8 + c);
        double cs = 0.0;   // compose_f32: products summed in double, rounded once
        cs += (double)d0 * (double)b0;
        cs += (double)d1 * (double)b1;
        cs += (double)d2 * (double)b2;
        if (c == 3) cs += (double)d3;
        const float Cf = (float)cs;
        float C[12], np[6];
#pragma unroll
        for (int i = 0; i < 12; i++) C[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Cf), i));
        log_se3_f32(C, np);
        double R2, V2;
        exp_se3_entry((double)np[0], (double)np[1], (double)np[2], (double)np[3], (double)np[4], (double)np[5], r3, k3, R2, V2);
        const double t2 = (V2 + __shfl_down(V2, 1)) + __shfl_down(V2, 2);
        if (lane < 9) {
          sh.newS[r3 * 4 + k3] = (float)R2;
          if (k3 == 0) sh.newS[r3 * 4 + 3] = (float)t2;
        }
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < 6; i++) sh.newpose[i] = np[i];
          sh.weighted = weighted;
          sh.level_done = (early_exit && weighted < 1.0f) ? level : level_done_cur;   // ImageFunc.cpp:251-252
        }
      }
      ELLC_STAMP(5);
    }
  }
  __syncthreads();
}

// Second half of the solve in tolerance mode: delta from the L D L^T solve (mode 0) or from the level's H^-1 (mode 2), then
// exp(pose) <- exp(delta) exp(pose) rounded to f32 — the same product the exact path forms — WITHOUT the round trip through
// log and exp that follows it there (PixelWisePyramid.cpp:483-489 keeps the pose as a twist and re-exponentiates it every
// iteration; mathematically the identity, a few 1e-8 per iteration in f32). The twist is recovered once, by the kernel that
// ends the schedule (log of the final matrix). Ends with a block barrier.
__device__ __forceinline__ void solve_finish_fast(SolveShared& sh, int mode, int level, int early_exit, const AlignState& src,
                                                  const float* S_cur, int level_done_cur, AlignState* dst) {
  const int t = threadIdx.x;
  if (t < 64) {
    const int lane = t;
    const float S_lane = S_cur[min(lane, 11)];
    double x[6];
    if (mode == 0) {
      double sums[27];
#pragma unroll
      for (int i = 0; i < 27; i++) sums[i] = sh.sums[i];
      ldlt_solve6(sums, x);
      if (lane == 0 && dst) {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = r; c < 6; c++) {
            const float v = (float)sums[q++];
            dst->H[r * 6 + c] = v;
            dst->H[c * 6 + r] = v;
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) acc += (double)(float)sh.sums[21 + k] * (double)sh.Hinv[i * 6 + k];
        x[i] = -acc;
      }
    }
    ELLC_STAMP(3);
    float delta[6];
#pragma unroll
    for (int i = 0; i < 6; i++) delta[i] = (float)x[i];
    if (lane == 0 && dst) {
#pragma unroll
      for (int i = 0; i < 6; i++) { dst->delta[i] = delta[i]; dst->b[i] = (float)sh.sums[21 + i]; }
    }
    ELLC_STAMP(4);
    const float weighted = fabsf(delta[0] * 100000.0f) + fabsf(delta[1] * 100000.0f) + fabsf(delta[2] * 100000.0f) +
                           fabsf(delta[3] * 10000.0f) + fabsf(delta[4] * 10000.0f) + fabsf(delta[5] * 10000.0f);
    const int l9 = min(lane, 8);
    const int r3 = l9 / 3, k3 = l9 - 3 * r3;
    double Rrk, Vv;
    exp_se3_entry<true>(x[0], x[1], x[2], x[3], x[4], x[5], r3, k3, Rrk, Vv);
    const double trow = (Vv + __shfl_down(Vv, 1)) + __shfl_down(Vv, 2);   // t[r] in lanes 0, 3, 6
    const float Df = (float)Rrk, Dt = (float)trow;
    const int e = min(lane, 11), r = e >> 2, c = e & 3;
    const float d0 = __shfl(Df, 3 * r), d1 = __shfl(Df, 3 * r + 1), d2 = __shfl(Df, 3 * r + 2), d3 = __shfl(Dt, 3 * r);
    const float b0 = __shfl(S_lane, c), b1 = __shfl(S_lane, 4 + c), b2 = __shfl(S_lane, 8 + c);
    double cs = 0.0;
    cs += (double)d0 * (double)b0;
    cs += (double)d1 * (double)b1;
    cs += (double)d2 * (double)b2;
    if (c == 3) cs += (double)d3;
    if (lane < 12) sh.newS[lane] = (float)cs;
    if (lane < 6) sh.newpose[lane] = src.pose[lane];   // not maintained per iteration in this mode
    if (lane == 0) {
      sh.weighted = weighted;
      sh.level_done = (early_exit && weighted < 1.0f) ? level : level_done_cur;   // ImageFunc.cpp:251-252
    }
    ELLC_STAMP(5);
  }
  __syncthreads();
}

template <bool FAST = false>
__device__ __forceinline__ void solve_step(SolveShared& sh, double group_sum, int mode, int level, int early_exit,
                                           const AlignState& src, AlignState* dst, const float* hinv_src = nullptr) {
  const int t = threadIdx.x;
  const int comp = t & 31, grp = t >> 5;
  sh.part[grp][comp] = group_sum;
  ELLC_STAMP(1);
  if (mode == 2 && t < 36) sh.Hinv[t] = hinv_src ? hinv_src[t] : src.Hinv[t];   // ICA iterate: the level's precomputed inverse
  __syncthreads();
  if (t < 27) {
    double s = sh.part[0][t];
#pragma unroll
    for (int g = 1; g < ELLC_SOLVE_THREADS / 32; g++) s += sh.part[g][t];
    sh.sums[t] = s;
  }
  __syncthreads();
  ELLC_STAMP(2);
  if (FAST && mode != 1) solve_finish_fast(sh, mode, level, early_exit, src, src.S, src.level_done, dst);
  else solve_finish(sh, mode, level, early_exit, src, src.S, src.level_done, dst);
}

// One block per alignment (used by the ICA path, the single-step API and as the final solve of a fused schedule).
template <bool FAST>
__global__ __launch_bounds__(ELLC_SOLVE_THREADS) void gn_solve(SolveArgs a) {
  const int b = blockIdx.x;
  AlignState& st = a.state[b];
  if (st.level_done == a.level) return;
  __shared__ SolveShared sh;
  solve_step<FAST>(sh, partial_group_sum(a.partials + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE, a.nblk), a.mode, a.level, a.early_exit, st, &st);
  if (a.mode == 1) return;
  const int t = threadIdx.x;
  if (FAST) {   // the twist of the updated matrix, for the callers of the single-step API
    if (t == 0) {
      float S[12], np[6];
      for (int i = 0; i < 12; i++) S[i] = sh.newS[i];
      log_se3_f32(S, np);
      for (int i = 0; i < 6; i++) sh.newpose[i] = np[i];
    }
    __syncthreads();
  }
  if (t < 6) st.pose[t] = sh.newpose[t];
  if (t < 12) st.S[t] = sh.newS[t];
  if (t == 0) {
    st.weighted = sh.weighted;
    st.iters[a.level] += 1;
    st.level_done = sh.level_done;
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused schedule for the FCA path: launch n first *consumes* the partial sums launch n-1 left behind (every
// block of the alignment redoes the tiny solve, so no extra launch and no cross-block synchronisation is needed),
// then runs its own pixel pass with the fresh pose. State and partials are double-buffered by launch parity:
// launch n reads state[n&1] / partials[(n+1)&1] and writes state[(n+1)&1] / partials[n&1].
struct ObsMats;
__device__ void track_setup_wave(const float* pose, const float* Kmat, ObsMats* mats);   // ellc_kernels_depth.hpp
struct FusedArgs {
  GnArgs g;
  int seq;          // launch index inside the schedule
  int prev_level;   // level / block count of the launch whose partials are pending
  int prev_nblk;
  int early_exit;
  int stride_state;     // elements between the two AlignState buffers
  size_t stride_part;   // floats between the two partial buffers
  // Age-balanced split (0 = off). When the grid is R full rounds of resident blocks, the blocks dispatched first share
  // their SIMDs with younger ones and are served first (oldest-wave-first arbitration), so with equal chunks the launch
  // waits for the youngest round (measured: pixel phases of 15 / 17 / 20 / 23 us for rounds 0..3 of an equal split).
  // With age_rounds = R every alignment gets nblk / R blocks in each round and round q is given the share
  // (age_cum[q+1] - age_cum[q]) / 65536 of the alignment's pixels. The mapping only assumes that blocks are dispatched
  // in linear order; if that were not so the result is unchanged (the split is static) and only the balance is lost.
  int age_rounds;
  int age_cum[5];
  int xcd_map;          // 1: blocks are renumbered so that all blocks of an alignment run on one XCD (see gn_fca_fused)
  AlignResult* res;     // gn_fused_finish: host-visible result records (null: none)
  int ica;              // 1: constant-weight schedule (gn_ica_fused): the pending sums are b only, H^-1 comes from the keyframe slot
  // state-driven schedule (gn_fca_adaptive): blocks and iteration caps per level, the grid's x extent
  int nblk_lv[ELLC_MAX_LEVELS];
  int max_it[ELLC_MAX_LEVELS];
  int nblk_grid;
  // tracked-frame call (ellc_track_frame): the finish kernel goes on to build the observation's matrices from the pose it has just
  // computed (track_setup_wave, ellc_kernels_depth.hpp) and opens the gate of the depth stages behind it; null: not such a call
  int host_polls;       // 1: the host may poll the result records' flag words (resolve_batch): the finish kernel orders its stores system-wide
  struct ObsMats* track_mats;
  int* track_gate;
  float track_K[9];
  unsigned* persist_bar = nullptr;   // gn_fca_persist's abort words of this batch set (ELLC_PERSIST_BAR_WORDS per alignment)
  int continuation;     // 1: this graph continues a state-driven schedule whose first graph has already run (and added the saved weights
                        // of the alignments that ended there): its first launch marks those records cur_level = -2
};

// The pixel pass of one block of a fused launch over its chunk [begin, end) of the compact list, thread t taking the
// entries begin + t, begin + t + 256, ...; the thread's first record (and, in the exact mode, its pose-independent products)
// was requested by the caller before the solve. newS: exp(pose) of this iteration (LDS). Leaves the thread's 27 sums.
template <bool DIVC, bool PIPE, bool FAST, int SAVEW, bool LAT = false>   // LAT: the latency regime (tap_general)
__device__ __forceinline__ void fca_chunk_pass(const GnArgs& a, const KfLevelDev& K, const LevelGeom& g, g_u8 cur, const float* newS, int begin,
                                               int end, const FcaIn& first, const FcaInF& firstf, const FcaPre& first_pre, float (&sums)[27]) {
  constexpr int stride = ELLC_GN_THREADS;
  const int t = threadIdx.x;
  float S[12];
#pragma unroll
  for (int i = 0; i < 12; i++) S[i] = newS[i];
  FcaAcc acc;
  fca_acc_zero(acc);
  int i = begin + t;
  if constexpr (FAST) {
    if (begin < end) {   // block-uniform
      const TapRows tr = tap_rows(cur, g.sw);
      const FcafConst fc = fcaf_const(g, S);
      ELLC_PTRACE(4, 0);
      // One pixel per step: the rows are requested and used in the same step; the next pixel's record is requested behind them. The
      // record stream is walked by byte offset, clamped to the chunk's last record, so that every request is unconditional, and the
      // trip count is block-uniform — every thread of the block has n_full pixels, the first `rem` threads one more, and a thread
      // without a pixel in the last step runs it on a copy of the chunk's last record without accumulating — so that the loop is a
      // plain scalar loop: a per-lane exit in the middle of the unrolled body makes the compiler merge the two record slots at the
      // back edge with register copies, and a copy of a slot waits for the load that fills it. The two slots alternate through the
      // explicitly unrolled body.
      unsigned off = (unsigned)i * ELLC_FREC, idx = (unsigned)i;
      const unsigned off_last = (unsigned)(end - 1) * ELLC_FREC;
      constexpr unsigned S16 = stride * ELLC_FREC;
      const int n_full = __builtin_amdgcn_readfirstlane((end - begin) / stride);
      const int rem = __builtin_amdgcn_readfirstlane((end - begin) - n_full * stride);
      const int n_steps = n_full + (rem > 0 ? 1 : 0);
      FcaInF r0 = firstf, r1 = firstf;
      auto step = [&](const FcaInF& cur_rec, FcaInF& next_rec, bool last) {
        const bool active = !last || rem == 0 || t < rem;
        auto refill = [&]() { next_rec = fcaf_load_off(K, min(off + S16, off_last)); };
        const FcafStage st = fcaf_stage_a(g, tr, fc, cur_rec, refill);
        ELLC_PTRACE(5, 0);
        if (active) fca_accumulate_pixel(acc, fcaf_stage_b<false, SAVEW, LAT>(a, K, g, cur, fc, idx, st));
        ELLC_PTRACE(6, 0);
        off += S16; idx += stride;
      };
      for (int k = 0; k < n_steps; k += 2) {
        step(r0, r1, k == n_steps - 1);
        if (k + 1 >= n_steps) break;   // block-uniform
        step(r1, r0, k + 1 == n_steps - 1);
      }
    }
  } else if (i < end) {
    if (PIPE) {   // exact mode: one record ahead (two slots; a third costs registers this kernel does not have)
      FcaIn r0 = first, r1 = first;
      {
        const int i1 = i + stride;
        auto prefetch = [&]() { r1 = fca_load<DIVC>(K, g, (unsigned)min(i1, end - 1)); };
        if constexpr (LAT) fca_accumulate_pixel(acc, fca_pixel_pre<false>(a, K, g, cur, S, (unsigned)i, first, first_pre, lat_pf(prefetch)));
        else fca_accumulate_pixel(acc, fca_pixel_pre<false>(a, K, g, cur, S, (unsigned)i, first, first_pre, prefetch));
        i += stride;
      }
      auto step = [&](const FcaIn& in, FcaIn& fill) {
        const int i1 = i + stride;
        auto prefetch = [&]() { fill = fca_load<DIVC>(K, g, (unsigned)min(i1, end - 1)); };
        if constexpr (LAT) fca_accumulate_pixel(acc, fca_pixel_in<false, DIVC>(a, K, g, cur, S, (unsigned)i, in, lat_pf(prefetch)));
        else fca_accumulate_pixel(acc, fca_pixel_in<false, DIVC>(a, K, g, cur, S, (unsigned)i, in, prefetch));
        i += stride;
      };
      while (i < end) {
        step(r1, r0);
        if (i >= end) break;
        step(r0, r1);
      }
    } else {
      fca_accumulate_pixel(acc, fca_pixel_pre<false>(a, K, g, cur, S, (unsigned)i, first, first_pre));
      for (i += stride; i < end; i += stride) {
        const FcaPix q = fca_pixel<false, DIVC>(a, K, g, cur, S, (unsigned)i);
        fca_accumulate_pixel(acc, q);
      }
    }
  }
  fca_acc_unpack<FAST>(acc, sums);
}

// The leading scalar parameters repeat what the prologue's first loads need (addresses of the state record and of the
// pending partial sums, block counts): the library is built with kernel-argument preloading, so they arrive in SGPRs with
// the wave instead of through a scalar load from the argument buffer — one memory round trip less at the head of a
// latency-bound kernel. Everything else stays in the by-value struct.
template <bool DIVC, bool PIPE, bool FAST = false, int SAVEW = -1>   // SAVEW: see fcaf_pixel (tolerance mode only)
__global__ __launch_bounds__(ELLC_GN_THREADS, 4) void gn_fca_fused(const AlignState* src_state, const float* prev_part, int prev_nblk,
                                                                   int nblk, int age_rounds, FusedArgs fa) {   // 4 waves per SIMD: at most 128 VGPRs
  const GnArgs& a = fa.g;
  int b = blockIdx.y, sub = blockIdx.x, age = 0, per_age = nblk;
  if (age_rounds > 1) {
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int per_round = (int)(gridDim.x * gridDim.y) / age_rounds;
    per_age = nblk / age_rounds;          // blocks of one alignment in each round
    age = lin / per_round;
    const int j = lin - age * per_round;
    b = j / per_age;
    sub = age * per_age + (j - b * per_age);
  } else if (fa.xcd_map) {
    // Workgroups are dealt round-robin over the 8 XCDs in dispatch order, each XCD with an L2 of its own. Renumbered so that
    // alignment b's blocks are the linear ids congruent to b mod 8, all of an alignment's taps (and its record list) go
    // through ONE L2: with a frame per alignment (1280x960: 1.2 MB each) every XCD otherwise pulls every image. Pure
    // relabelling of (alignment, chunk): results unchanged; a different dispatch order would only lose the locality.
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int w = lin >> 3, bl = w / nblk;
    sub = w - bl * nblk;
    b = bl * 8 + (lin & 7);
  }
  const AlignState& src = src_state[b];
  AlignState* dst = a.state + (size_t)((fa.seq + 1) & 1) * fa.stride_state + b;
  __shared__ SolveShared sh;
  const int t = threadIdx.x;
  const bool writer = (sub == 0);
  ELLC_STAMP(0);
  ELLC_BSTAMP(0);
  ELLC_SEQSTAMP(0, fa.seq);
  // Load order of the prologue (everything below depends on the kernel arguments only, or on the uniform table
  // entries): the scalar chain slot -> table entry -> count is started first, the pending partial sums are read
  // unconditionally next (prev_nblk is 0 on the first launch of a schedule) so that they share one memory round trip
  // with the chain and the state record, and this thread's first compact pixel is requested last, to arrive while
  // the solve runs. None of it depends on the pose.
  const LevelGeom g = a.geom[a.level];
  const KfLevelDev K = a.kf_tab[a.level * a.max_kf + a.kf_slot[b]];   // by value: uniform, lives in SGPRs
  const FrLevelDev& F = a.fr_tab[a.level * a.max_fr + a.fr_slot[b]];
  const int pending = src.pending;
  const int V = *as_global(K.count);
  const double group_sum = partial_group_sum(prev_part + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE, prev_nblk);
  // contiguous chunk per block (keeps a block's taps in a few image rows: 25 % less fetch traffic than a tile-cyclic
  // split, which was tried in r01 and did not change the run time — co-resident blocks finish staggered because the
  // SIMD arbiter serves the oldest wave first, not because their pixels differ)
  int begin, end;
  if (age_rounds > 1) {
    const int gb = (int)(((long long)V * fa.age_cum[age]) >> 16), ge = (int)(((long long)V * fa.age_cum[age + 1]) >> 16);
    const int chunk = (ge - gb + per_age - 1) / per_age;
    begin = gb + (sub - age * per_age) * chunk;
    end = min(ge, begin + chunk);
  } else {
    const int chunk = (V + nblk - 1) / nblk;
    begin = sub * chunk;
    end = min(V, begin + chunk);
  }
  g_u8 cur = as_global(F.img);
  // this thread's first compact pixel, requested before the solve (exact mode: together with its pose-independent products)
  FcaIn first = fca_in_empty();
  FcaInF firstf = fcaf_empty();
  FcaPre first_pre;
  if constexpr (FAST) {
    if (begin < end) firstf = fcaf_load(K, (unsigned)min(begin + t, end - 1));   // (a thread past the chunk's end starts on a copy of its last record: the pixel loop is block-uniform)
  } else {
    if (begin + t < end) {
      first = fca_load<DIVC>(K, g, (unsigned)(begin + t));
    }
    first_pre = fca_prepare<DIVC>(g, first);
    // pin the arithmetic here (the compiler would otherwise sink it below the solve, onto the critical path)
    asm volatile("" ::"v"(first_pre.c_t0), "v"(first_pre.c_b1), "v"(first_pre.d), "v"(first_pre.fxz), "v"(first_pre.fyz),
                 "v"(first_pre.nvz), "v"(first_pre.nuz));
  }
  if (pending) {
    // (H, b, delta and H^-1 of the state record serve the single-step API only, which runs gn_solve: not stored here)
    solve_step<FAST>(sh, group_sum, 0, fa.prev_level, fa.early_exit, src, nullptr);
  } else {
    if (t < 6) sh.newpose[t] = src.pose[t];
    if (t < 12) sh.newS[t] = src.S[t];
    if (t == 0) { sh.weighted = src.weighted; sh.level_done = src.level_done; }
    __syncthreads();
  }
  ELLC_STAMP(6);
  ELLC_BSTAMP(1);
  const int level_done = sh.level_done;
  const bool skip = (level_done == a.level);
  if (writer) {
    if (t < 6) dst->pose[t] = sh.newpose[t];
    if (t < 12) dst->S[t] = sh.newS[t];
    if (t < ELLC_MAX_LEVELS) dst->iters[t] = src.iters[t] + ((pending && t == fa.prev_level) ? 1 : 0);
    if (t == 0) {
      dst->weighted = sh.weighted;
      dst->level_done = level_done;
      dst->pending = skip ? 0 : 1;
    }
  }
  if (skip) return;
  float sums[27];
  fca_chunk_pass<DIVC, PIPE, FAST, SAVEW>(a, K, g, cur, sh.newS, begin, end, first, firstf, first_pre, sums);
  ELLC_STAMP(7);
  ELLC_BSTAMP(2);
  float* out = a.partials + (size_t)(fa.seq & 1) * fa.stride_part + ((size_t)b * ELLC_NBLK_MAX + sub) * ELLC_PART_STRIDE;
  block_reduce_store<27>(sums, out);
  ELLC_STAMP(8);
  ELLC_BSTAMP(3);
  ELLC_SEQSTAMP(32, fa.seq);
}

// ---------------------------------------------------------------------------------------------------
// The list-free form of gn_fca_fused for DENSE maps (r05; BASELINE configs[4]: 1280x960, every pixel holds a depth). When at least
// nine tenths of a keyframe's pixels are valid a compact list is the plane itself, only larger: the compaction moved 25 bytes per
// pixel to turn 9 bytes of planes (depth 4, variance 4, intensity 1) into a record that every later launch read back — 13 %
// of a dense launch group's kernel time went into compacting a mask that is all ones. Here thread <-> pixel by index: a block owns
// the contiguous chunk of the PLANE that it owned of the list (the same split, age-balanced where the list's was), thread t takes
// pixels begin + t, begin + t + 256, ..., reads the three planes with coalesced loads one step ahead (requested behind the tap
// loads, where the record prefetch sat), forms the record's four words in registers — the expressions prep_scatter stores, so the
// per-pixel values are those of the list path bit for bit — and runs the same pixel step. A pixel without depth is skipped where it
// occurs (it takes the step on a stand-in and accumulates nothing). No compaction launch, no record traffic: a full-schedule
// alignment reads 9 bytes per pixel and iteration where the list path moved 16 + the compaction's 25 once per level.
// Tolerance mode, no saved weights (the list path keeps both); schedules as gn_fca_fused (launch n solves launch n - 1's sums).
struct DensePix { float Z, var; uint32_t I; };
__device__ __forceinline__ DensePix dense_request(const KfLevelDev& K, unsigned i, unsigned img_off) {
  DensePix p;
  p.Z = as_global(K.depth)[i];
  p.var = as_global(K.var)[i];
  p.I = (uint32_t)as_global(K.img)[img_off];
  return p;
}
__device__ __forceinline__ FcaInF dense_form(const LevelGeom& g, const DensePix& p, int x, int y, bool& valid) {
  valid = p.Z > 0.0f;
  FcaInF in;
  const float dd = __builtin_amdgcn_rcpf(valid ? p.Z : 1.0f);
  in.v = (u32x4_t){(uint32_t)x | ((uint32_t)y << 12) | (p.I << 24), __builtin_bit_cast(uint32_t, p.var), __builtin_bit_cast(uint32_t, dd), 0u};
  return in;
}

__global__ __launch_bounds__(ELLC_GN_THREADS, 4) void gn_fca_dense(const AlignState* src_state, const float* prev_part, int prev_nblk, int nblk, int age_rounds,
                                                                   FusedArgs fa) {
  const GnArgs& a = fa.g;
  int b = blockIdx.y, sub = blockIdx.x, age = 0, per_age = nblk;
  if (age_rounds > 1) {   // see gn_fca_fused
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int per_round = (int)(gridDim.x * gridDim.y) / age_rounds;
    per_age = nblk / age_rounds;
    age = lin / per_round;
    const int j = lin - age * per_round;
    b = j / per_age;
    sub = age * per_age + (j - b * per_age);
  } else if (fa.xcd_map) {
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int w = lin >> 3, bl = w / nblk;
    sub = w - bl * nblk;
    b = bl * 8 + (lin & 7);
  }
  const AlignState& src = src_state[b];
  AlignState* dst = a.state + (size_t)((fa.seq + 1) & 1) * fa.stride_state + b;
  __shared__ SolveShared sh;
  const int t = threadIdx.x;
  const bool writer = (sub == 0);
  const LevelGeom g = a.geom[a.level];
  const KfLevelDev K = a.kf_tab[a.level * a.max_kf + a.kf_slot[b]];   // by value: uniform, lives in SGPRs
  const FrLevelDev& F = a.fr_tab[a.level * a.max_fr + a.fr_slot[b]];
  const int pending = src.pending;
  const int V = g.n;   // the "list" is the plane
  const double group_sum = partial_group_sum(prev_part + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE, prev_nblk);
  int begin, end;
  if (age_rounds > 1) {
    const int gb = (int)(((long long)V * fa.age_cum[age]) >> 16), ge = (int)(((long long)V * fa.age_cum[age + 1]) >> 16);
    const int chunk = (ge - gb + per_age - 1) / per_age;
    begin = gb + (sub - age * per_age) * chunk;
    end = min(ge, begin + chunk);
  } else {
    const int chunk = (V + nblk - 1) / nblk;
    begin = sub * chunk;
    end = min(V, begin + chunk);
  }
  g_u8 cur = as_global(F.img);
  // this thread's first pixel: position and planes, requested before the solve
  const int cols = g.cols, sw = g.sw;
  const int qstep = ELLC_GN_THREADS / cols, rstep = ELLC_GN_THREADS - qstep * cols;   // a step advances a thread by 256 pixels: qstep rows and rstep columns
  int x = 0, y = 0;
  DensePix pix;
  pix.Z = 0.0f; pix.var = 0.0f; pix.I = 0u;
  if (begin < end) {   // block-uniform (a thread past the chunk's end starts on a copy of its last pixel: the pixel loop is block-uniform)
    const int i0 = min(begin + t, end - 1);
    y = (int)(((float)i0 + 0.5f) * (1.0f / (float)cols));   // i < 2^24: exact conversion; corrected to the exact quotient
    if (y * cols > i0) y--;
    if ((y + 1) * cols <= i0) y++;
    x = i0 - y * cols;
    pix = dense_request(K, (unsigned)i0, (unsigned)(y * sw + x));
  }
  if (pending) {
    solve_step<true>(sh, group_sum, 0, fa.prev_level, fa.early_exit, src, nullptr);
  } else {
    if (t < 6) sh.newpose[t] = src.pose[t];
    if (t < 12) sh.newS[t] = src.S[t];
    if (t == 0) { sh.weighted = src.weighted; sh.level_done = src.level_done; }
    __syncthreads();
  }
  const int level_done = sh.level_done;
  const bool skip = (level_done == a.level);
  if (writer) {
    if (t < 6) dst->pose[t] = sh.newpose[t];
    if (t < 12) dst->S[t] = sh.newS[t];
    if (t < ELLC_MAX_LEVELS) dst->iters[t] = src.iters[t] + ((pending && t == fa.prev_level) ? 1 : 0);
    if (t == 0) {
      dst->weighted = sh.weighted;
      dst->level_done = level_done;
      dst->pending = skip ? 0 : 1;
    }
  }
  if (skip) return;
  float sums[27];
  {
    float S[12];
#pragma unroll
    for (int i = 0; i < 12; i++) S[i] = sh.newS[i];
    FcaAcc acc;
    fca_acc_zero(acc);
    if (begin < end) {   // block-uniform
      const TapRows tr = tap_rows(cur, sw);
      const FcafConst fc = fcaf_const(g, S);
      const int n_full = __builtin_amdgcn_readfirstlane((end - begin) / ELLC_GN_THREADS);
      const int rem = __builtin_amdgcn_readfirstlane((end - begin) - n_full * ELLC_GN_THREADS);
      const int n_steps = n_full + (rem > 0 ? 1 : 0);
      int i = begin + t;
      for (int k = 0; k < n_steps; k++) {
        const bool last = (k == n_steps - 1);
        const bool active = !last || rem == 0 || t < rem;
        bool valid;
        const FcaInF rec = dense_form(g, pix, x, y, valid);
        // the next pixel of this thread: 256 further on (clamped to the chunk's last pixel: every request is unconditional)
        int xn = x + rstep, yn = y + qstep;
        if (xn >= cols) { xn -= cols; yn++; }
        const int in_ = i + ELLC_GN_THREADS;
        if (in_ > end - 1) {   // past the end: the chunk's last pixel
          const int il = end - 1;
          yn = (int)(((float)il + 0.5f) * (1.0f / (float)cols));
          if (yn * cols > il) yn--;
          if ((yn + 1) * cols <= il) yn++;
          xn = il - yn * cols;
        }
        const unsigned inext = (unsigned)min(in_, end - 1);
        auto refill = [&]() { pix = dense_request(K, inext, (unsigned)(yn * sw + xn)); };
        const FcafStage st = fcaf_stage_a(g, tr, fc, rec, refill);
        if (active && valid) fca_accumulate_pixel(acc, fcaf_stage_b<false, 0>(a, K, g, cur, fc, (unsigned)i, st));
        x = xn; y = yn; i = in_;
      }
    }
    fca_acc_unpack<true>(acc, sums);
  }
  float* out = a.partials + (size_t)(fa.seq & 1) * fa.stride_part + ((size_t)b * ELLC_NBLK_MAX + sub) * ELLC_PART_STRIDE;
  block_reduce_store<27>(sums, out);
}

// ---------------------------------------------------------------------------------------------------
// gn_fca_dense in the EXACT arithmetic (r06; r05's verdict: the exact mode's C4 launch moved 1.28 x its algorithmic bytes as 20-byte
// records, and its batch spent 12 % of its time compacting a mask that is all ones). The same thread <-> pixel walk over the block's
// chunk of the plane; the record's four values come from the planes — depth, variance, intensity and the slot's 1 / Z plane in
// double (KfLevelDev::invz, written once per upload: the f64 division is the compaction's expression, per pixel and ITERATION it
// would be a seventh of this instruction-bound pass) — and go through fca_load's and fca_pixel_in's expressions: per-pixel values
// bit for bit those of the list path, sums in another order. No saved weights (the list path keeps them), level-bound schedule.
struct DensePixX { float Z, var; uint32_t I; double invZ; };
__device__ __forceinline__ DensePixX dense_request_x(const KfLevelDev& K, unsigned i, unsigned img_off) {
  DensePixX p;
  p.Z = as_global(K.depth)[i];
  p.var = as_global(K.var)[i];
  p.I = (uint32_t)as_global(K.img)[img_off];
  p.invZ = ((const ELLC_GLOBAL double*)K.invz)[i];
  return p;
}
template <bool DIVC>
__global__ __launch_bounds__(ELLC_GN_THREADS, 4) void gn_fca_dense_x(const AlignState* src_state, const float* prev_part, int prev_nblk, int nblk, int age_rounds,
                                                                     FusedArgs fa) {
  const GnArgs& a = fa.g;
  int b = blockIdx.y, sub = blockIdx.x, age = 0, per_age = nblk;
  if (age_rounds > 1) {   // see gn_fca_fused
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int per_round = (int)(gridDim.x * gridDim.y) / age_rounds;
    per_age = nblk / age_rounds;
    age = lin / per_round;
    const int j = lin - age * per_round;
    b = j / per_age;
    sub = age * per_age + (j - b * per_age);
  } else if (fa.xcd_map) {
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int w = lin >> 3, bl = w / nblk;
    sub = w - bl * nblk;
    b = bl * 8 + (lin & 7);
  }
  const AlignState& src = src_state[b];
  AlignState* dst = a.state + (size_t)((fa.seq + 1) & 1) * fa.stride_state + b;
  __shared__ SolveShared sh;
  const int t = threadIdx.x;
  const bool writer = (sub == 0);
  const LevelGeom g = a.geom[a.level];
  const KfLevelDev K = a.kf_tab[a.level * a.max_kf + a.kf_slot[b]];
  const FrLevelDev& F = a.fr_tab[a.level * a.max_fr + a.fr_slot[b]];
  const int pending = src.pending;
  const int V = g.n;   // the "list" is the plane
  const double group_sum = partial_group_sum(prev_part + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE, prev_nblk);
  int begin, end;
  if (age_rounds > 1) {
    const int gb = (int)(((long long)V * fa.age_cum[age]) >> 16), ge = (int)(((long long)V * fa.age_cum[age + 1]) >> 16);
    const int chunk = (ge - gb + per_age - 1) / per_age;
    begin = gb + (sub - age * per_age) * chunk;
    end = min(ge, begin + chunk);
  } else {
    const int chunk = (V + nblk - 1) / nblk;
    begin = sub * chunk;
    end = min(V, begin + chunk);
  }
  g_u8 cur = as_global(F.img);
  const int cols = g.cols, sw = g.sw;
  const int qstep = ELLC_GN_THREADS / cols, rstep = ELLC_GN_THREADS - qstep * cols;
  int x = 0, y = 0;
  DensePixX pix;
  pix.Z = 0.0f; pix.var = 0.0f; pix.I = 0u; pix.invZ = 1.0;
  if (begin < end) {   // block-uniform (a thread past the chunk's end starts on a copy of its last pixel: the pixel loop is block-uniform)
    const int i0 = min(begin + t, end - 1);
    y = (int)(((float)i0 + 0.5f) * (1.0f / (float)cols));
    if (y * cols > i0) y--;
    if ((y + 1) * cols <= i0) y++;
    x = i0 - y * cols;
    pix = dense_request_x(K, (unsigned)i0, (unsigned)(y * sw + x));
  }
  if (pending) {
    solve_step<false>(sh, group_sum, 0, fa.prev_level, fa.early_exit, src, nullptr);
  } else {
    if (t < 6) sh.newpose[t] = src.pose[t];
    if (t < 12) sh.newS[t] = src.S[t];
    if (t == 0) { sh.weighted = src.weighted; sh.level_done = src.level_done; }
    __syncthreads();
  }
  const int level_done = sh.level_done;
  const bool skip = (level_done == a.level);
  if (writer) {
    if (t < 6) dst->pose[t] = sh.newpose[t];
    if (t < 12) dst->S[t] = sh.newS[t];
    if (t < ELLC_MAX_LEVELS) dst->iters[t] = src.iters[t] + ((pending && t == fa.prev_level) ? 1 : 0);
    if (t == 0) {
      dst->weighted = sh.weighted;
      dst->level_done = level_done;
      dst->pending = skip ? 0 : 1;
    }
  }
  if (skip) return;
  float sums[27];
  {
    float S[12];
#pragma unroll
    for (int i = 0; i < 12; i++) S[i] = sh.newS[i];
    FcaAcc acc;
    fca_acc_zero(acc);
    if (begin < end) {   // block-uniform
      const int n_full = __builtin_amdgcn_readfirstlane((end - begin) / ELLC_GN_THREADS);
      const int rem = __builtin_amdgcn_readfirstlane((end - begin) - n_full * ELLC_GN_THREADS);
      const int n_steps = n_full + (rem > 0 ? 1 : 0);
      int i = begin + t;
      for (int k = 0; k < n_steps; k++) {
        const bool last = (k == n_steps - 1);
        const bool active = !last || rem == 0 || t < rem;
        const bool valid = pix.Z > 0.0f;
        FcaIn in;   // fca_load's values, from the planes
        in.xy = ((uint32_t)y << 16) | (uint32_t)x;
        in.Ikf = (float)pix.I;
        in.Z = valid ? pix.Z : 1.0f;
        in.var = pix.var;
        in.invZ = valid ? pix.invZ : 1.0;
        const float aX = ((float)x - g.cx) * in.Z, aY = ((float)y - g.cy) * in.Z;
        in.X = DIVC ? div_const(aX, g.fx, g.rfx) : aX / g.fx;
        in.Y = DIVC ? div_const(aY, g.fy, g.rfy) : aY / g.fy;
        // the next pixel of this thread: 256 further on (clamped to the chunk's last pixel: every request is unconditional)
        int xn = x + rstep, yn = y + qstep;
        if (xn >= cols) { xn -= cols; yn++; }
        const int in_ = i + ELLC_GN_THREADS;
        if (in_ > end - 1) {
          const int il = end - 1;
          yn = (int)(((float)il + 0.5f) * (1.0f / (float)cols));
          if (yn * cols > il) yn--;
          if ((yn + 1) * cols <= il) yn++;
          xn = il - yn * cols;
        }
        const unsigned inext = (unsigned)min(in_, end - 1);
        auto refill = [&]() { pix = dense_request_x(K, inext, (unsigned)(yn * sw + xn)); };
        const FcaPix q = fca_pixel_in<false, DIVC>(a, K, g, cur, S, (unsigned)i, in, refill);
        if (active && valid) fca_accumulate_pixel(acc, q);
        x = xn; y = yn; i = in_;
      }
    }
    fca_acc_unpack<false>(acc, sums);
  }
  float* out = a.partials + (size_t)(fa.seq & 1) * fa.stride_part + ((size_t)b * ELLC_NBLK_MAX + sub) * ELLC_PART_STRIDE;
  block_reduce_store<27>(sums, out);
}

// ---------------------------------------------------------------------------------------------------
// gn_fca_dense with FOUR ADJACENT PIXELS PER THREAD (r06), for levels whose width is a multiple of four. What the pixel pass of a
// dense level pays for is the number of wave-level load instructions it hands the CU's vector cache — about 16.7 cycles each whatever
// the lanes ask for, twice that when the lanes of a tap request sit on two image rows (tools/micro/quad_window.hip: the taps of 256
// neighbouring pixels cost 253-520 cycles of the cache as 4 x 4 unaligned dwords, 139 as 5 rows of ONE 8-byte window per four
// pixels) — against 424 cycles of the four SIMDs' arithmetic for the same 256 pixels. Here a thread owns pixels x .. x + 3 of one
// keyframe row: depth and variance arrive as one dwordx4 each, the four intensities as one dword; the four points are warped
// (the row's share of K exp(pose) (p, q, 1) is formed once per thread); and when their neighbourhoods "fit" — all interior, the four
// floor(x) within 4 columns, the four floor(y) within 2 rows — FIVE 8-byte requests at (min floor(y) - 1 .. + 3, min floor(x) - 1)
// serve all four: a pixel's row words are cut out with v_perm_b32 (its column offset as the byte selector) and chosen by its row
// offset. 8 load instructions per four pixels instead of 28. From there on the pixel step is fcaf_stage_b's, unchanged.
//   A thread whose four points do not fit (image border, a depth edge, a map with holes next to it) does not branch into the
// per-tap path — one such lane would hold up its wave, and with 256 pixels of a row per wave two waves in five touch the border —
// it appends its (valid) pixels to a queue of the WAVE in LDS, and whenever 64 are queued the wave runs them through
// gn_fca_dense's step (dense_form + fcaf_stage_a / _b: the per-pixel values of that kernel), all lanes busy; the rest at the end of
// the chunk. The queue's order is fixed by ballots: same inputs, same bits. Chunks are split in units of four pixels.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef u32x2_t u32x2_a1 __attribute__((aligned(1)));
struct QuadPlanes { f32x4_t Z, var; uint32_t I; };
__device__ __forceinline__ QuadPlanes quad_request(const KfLevelDev& K, unsigned i4, unsigned img_off) {
  QuadPlanes p;
  const unsigned boff = i4 << 2;   // 32-bit byte offsets: uniform base + lane offset, no 64-bit lane arithmetic (planes < 4 GiB)
  p.Z = *(const ELLC_GLOBAL f32x4_t*)((const ELLC_GLOBAL char*)K.idepth + boff);   // 1 / depth (0: none): KfLevelDev::idepth
  p.var = *(const ELLC_GLOBAL f32x4_t*)((const ELLC_GLOBAL char*)K.var + boff);
  p.I = *(const ELLC_GLOBAL uint32_t*)((const ELLC_GLOBAL char*)K.img + img_off);
  return p;
}
// Three blocks per CU (168 registers a thread): with four pixels in flight per thread the step holds the planes of the NEXT quad (nine
// registers, requested a whole step ahead: they come from HBM) beside everything of this one; at four per CU (128) that spills, and
// with the variances and intensities requested in the step that uses them the tap rows wait behind them (vector loads return in
// order): 412 us against the r05 kernel's 369. The host sizes the grids of list-free launches for three per CU (choose_nblk).
#define ELLC_QUAD_BLOCKS_PER_CU 3
#define ELLC_QUAD_QCAP (64 + 4 * 64)   // entries of a wave's queue: drained below 64 after every step, a step appends at most 256
#ifdef ELLC_QUAD_STATS
__device__ unsigned long long g_quad_stats[4];   // diagnostic builds: quads seen / quads that did not fit / pixels queued / drain rounds
#endif

__global__ __launch_bounds__(ELLC_GN_THREADS, ELLC_QUAD_BLOCKS_PER_CU) void gn_fca_dense4(const AlignState* src_state, const float* prev_part, int prev_nblk, int nblk, int age_rounds,
                                                                    FusedArgs fa) {
  ELLC_BSTAMP(0);
  const GnArgs& a = fa.g;
  int b = blockIdx.y, sub = blockIdx.x, age = 0, per_age = nblk;
  if (age_rounds > 1) {   // see gn_fca_fused
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int per_round = (int)(gridDim.x * gridDim.y) / age_rounds;
    per_age = nblk / age_rounds;
    age = lin / per_round;
    const int j = lin - age * per_round;
    b = j / per_age;
    sub = age * per_age + (j - b * per_age);
  } else if (fa.xcd_map) {
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int w = lin >> 3, bl = w / nblk;
    sub = w - bl * nblk;
    b = bl * 8 + (lin & 7);
  }
  const AlignState& src = src_state[b];
  AlignState* dst = a.state + (size_t)((fa.seq + 1) & 1) * fa.stride_state + b;
  __shared__ SolveShared sh;
  __shared__ uint32_t qbuf[ELLC_GN_THREADS / 64][ELLC_QUAD_QCAP];
  const int t = threadIdx.x;
  const bool writer = (sub == 0);
  const LevelGeom g = a.geom[a.level];
  const KfLevelDev K = a.kf_tab[a.level * a.max_kf + a.kf_slot[b]];   // by value: uniform, lives in SGPRs
  const FrLevelDev& F = a.fr_tab[a.level * a.max_fr + a.fr_slot[b]];
  const int pending = src.pending;
  const int cols = g.cols, sw = g.sw;
  const int cols4 = cols >> 2;      // quads per row (the host launches this kernel for cols % 4 == 0 only)
  const int V = g.n >> 2;           // the "list" is the plane, in quads
  const double group_sum = partial_group_sum(prev_part + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE, prev_nblk);
  int begin, end;
  if (age_rounds > 1) {
    const int gb = (int)(((long long)V * fa.age_cum[age]) >> 16), ge = (int)(((long long)V * fa.age_cum[age + 1]) >> 16);
    const int chunk = (ge - gb + per_age - 1) / per_age;
    begin = gb + (sub - age * per_age) * chunk;
    end = min(ge, begin + chunk);
  } else {
    const int chunk = (V + nblk - 1) / nblk;
    begin = sub * chunk;
    end = min(V, begin + chunk);
  }
  g_u8 cur = as_global(F.img);
  // this thread's first quad: position and planes, requested before the solve
  const int qstep = ELLC_GN_THREADS / cols4, rstep = ELLC_GN_THREADS - qstep * cols4;   // a step advances a thread by 256 quads: qstep rows and rstep quad columns
  int xq = 0, y = 0;   // quad column, row
  QuadPlanes pl;
  pl.Z = (f32x4_t)(0.0f); pl.var = (f32x4_t)(0.0f); pl.I = 0u;
  if (begin < end) {   // block-uniform (a thread past the chunk's end starts on a copy of its last quad: the loop is block-uniform)
    const int i0 = min(begin + t, end - 1);
    y = (int)(((float)i0 + 0.5f) * (1.0f / (float)cols4));   // i < 2^24: exact conversion; corrected to the exact quotient
    if (y * cols4 > i0) y--;
    if ((y + 1) * cols4 <= i0) y++;
    xq = i0 - y * cols4;
    pl = quad_request(K, 4u * (unsigned)i0, __umul24((unsigned)y, (unsigned)sw) + 4u * (unsigned)xq);
  }
  if (pending) {
    solve_step<true>(sh, group_sum, 0, fa.prev_level, fa.early_exit, src, nullptr);
  } else {
    if (t < 6) sh.newpose[t] = src.pose[t];
    if (t < 12) sh.newS[t] = src.S[t];
    if (t == 0) { sh.weighted = src.weighted; sh.level_done = src.level_done; }
    __syncthreads();
  }
  const int level_done = sh.level_done;
  const bool skip = (level_done == a.level);
  if (writer) {
    if (t < 6) dst->pose[t] = sh.newpose[t];
    if (t < 12) dst->S[t] = sh.newS[t];
    if (t < ELLC_MAX_LEVELS) dst->iters[t] = src.iters[t] + ((pending && t == fa.prev_level) ? 1 : 0);
    if (t == 0) {
      dst->weighted = sh.weighted;
      dst->level_done = level_done;
      dst->pending = skip ? 0 : 1;
    }
  }
  ELLC_BSTAMP(1);
  if (skip) return;
  float sums[27];
  {
    // exp(pose) of this iteration: uniform, so into scalar registers (the per-pixel operands get vector copies below: an SGPR
    // operand halves the issue rate of the f32 multiply-adds; what is used once per quad or per drain stays scalar)
    float S[12];
#pragma unroll
    for (int i = 0; i < 12; i++) S[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sh.newS[i])));
    FcaAcc acc;
    fca_acc_zero(acc);
    if (begin < end) {   // block-uniform
      const TapRows tr = tap_rows(cur, sw);
      const g_u8 row_e = tr.rd + sw;   // the fifth row of a window: y0 + 3
      // P = K exp(pose) (fcaf_const), in scalar registers
      float P[12];
      {
        const FcafConst f0 = fcaf_const(g, S);
#pragma unroll
        for (int i = 0; i < 12; i++) P[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f0.P[i])));
      }
      const float s_rfy = g.rfy, s_qc = -(g.cy * g.rfy), s_pc = -(g.cx * g.rfx);
      float vP0 = P[0], vP3 = P[3], vP4 = P[4], vP7 = P[7], vP8 = P[8], vP11 = P[11], v_rfx = g.rfx, v_hfx = 0.5f * g.fx, v_hfy = 0.5f * g.fy;
      asm volatile("" : "+v"(vP0), "+v"(vP3), "+v"(vP4), "+v"(vP7), "+v"(vP8), "+v"(vP11), "+v"(v_rfx), "+v"(v_hfx), "+v"(v_hfy));
      const int n_full = __builtin_amdgcn_readfirstlane((end - begin) / ELLC_GN_THREADS);
      const int rem = __builtin_amdgcn_readfirstlane((end - begin) - n_full * ELLC_GN_THREADS);
      const int n_steps = n_full + (rem > 0 ? 1 : 0);
      const int lane = t & 63;
      uint32_t* const q = qbuf[t >> 6];
      int qn = 0;   // wave-uniform: entries queued
      int i = begin + t;
      for (int k = 0;; k++) {
        const bool more = (k < n_steps);   // block-uniform
        if (more) {
          const bool last = (k == n_steps - 1);
          const bool active = !last || rem == 0 || t < rem;
          const f32x4_t cZ = pl.Z, cvar = pl.var;
          const uint32_t cI = pl.I;
          const unsigned i4 = 4u * (unsigned)min(i, end - 1);
          const int x = 4 * xq, yrow = y;
          // the next quad of this thread: 256 further on (clamped to the chunk's last quad: every request is unconditional)
          int xn = xq + rstep, yn = y + qstep;
          if (xn >= cols4) { xn -= cols4; yn++; }
          const int in_ = i + ELLC_GN_THREADS;
          if (in_ > end - 1) {
            const int il = end - 1;
            yn = (int)(((float)il + 0.5f) * (1.0f / (float)cols4));
            if (yn * cols4 > il) yn--;
            if ((yn + 1) * cols4 <= il) yn++;
            xn = il - yn * cols4;
          }
          const unsigned inext = 4u * (unsigned)min(in_, end - 1);
          // inverse depths (the slot's reciprocal planes: the v_rcp_f32 the compaction would store; 0 = no depth); a pixel without
          // depth takes a neighbour's (its point then lands beside theirs and does not spoil the fit)
          const bool v0 = cZ.x > 0.0f, v1 = cZ.y > 0.0f, v2 = cZ.z > 0.0f, v3 = cZ.w > 0.0f;
          float d0 = cZ.x, d1 = cZ.y, d2 = cZ.z, d3 = cZ.w;
          if (__builtin_amdgcn_ballot_w64(!(v0 && v1 && v2 && v3)) != 0ull) {
            const float dref = v0 ? d0 : (v1 ? d1 : (v2 ? d2 : (v3 ? d3 : 1.0f)));
            d0 = v0 ? d0 : dref; d1 = v1 ? d1 : dref; d2 = v2 ? d2 : dref; d3 = v3 ? d3 : dref;
          }
          const float dd[4] = {d0, d1, d2, d3};
          const bool vv[4] = {v0, v1, v2, v3};
          // the row's and the thread's share of K exp(pose) (p, q, 1): once per quad (scalar operands)
          const float p0 = __builtin_fmaf((float)x, v_rfx, s_pc), qq = __builtin_fmaf((float)yrow, s_rfy, s_qc);
          const float bqx = __builtin_fmaf(P[1], qq, P[2]), bqy = __builtin_fmaf(P[5], qq, P[6]), bqz = __builtin_fmaf(P[9], qq, P[10]);
          // per pixel, what the second half needs of the warp: the position and half the reciprocal depth of the warped point
          float x1[4], y1[4], hrz[4];
          int x0[4], y0[4];
          int xs, xm, ys, ym;
          {
            float pj = p0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float px = __builtin_fmaf(vP0, pj, __builtin_fmaf(vP3, dd[j], bqx));
              const float py = __builtin_fmaf(vP4, pj, __builtin_fmaf(vP7, dd[j], bqy));
              const float pz = __builtin_fmaf(vP8, pj, __builtin_fmaf(vP11, dd[j], bqz));
              const float rz = __builtin_amdgcn_rcpf(pz);   // no clamp of pz (fcaf_stage_a)
              x1[j] = px * rz; y1[j] = py * rz; hrz[j] = 0.5f * rz;
              x0[j] = cvt_floor_i32(x1[j]); y0[j] = cvt_floor_i32(y1[j]);
              pj += v_rfx;
            }
            xs = min(min(x0[0], x0[1]), min(x0[2], x0[3])); xm = max(max(x0[0], x0[1]), max(x0[2], x0[3]));
            ys = min(min(y0[0], y0[1]), min(y0[2], y0[3])); ym = max(max(y0[0], y0[1]), max(y0[2], y0[3]));
          }
          // all sixteen-neighbourhoods interior (x0 in [1, cols - 3], y0 in [1, rows - 3]; a NaN converts to 0, an infinity saturates:
          // neither passes), the columns within one 8-byte window, the rows within five
          const bool fit = ((unsigned)(xs - 1) <= (unsigned)(cols - 4)) & ((unsigned)(xm - 1) <= (unsigned)(cols - 4)) &
                           ((unsigned)(ys - 1) <= (unsigned)(g.rows - 4)) & ((unsigned)(ym - 1) <= (unsigned)(g.rows - 4)) &
                           (xm - xs <= 4) & (ym - ys <= 1);
          const unsigned off = fit ? __umul24((unsigned)ys, (unsigned)sw) + (unsigned)xs : (unsigned)sw + 1u;
          const u32x2_t wA = *(const ELLC_GLOBAL u32x2_a1*)(tr.ra + off);
          const u32x2_t wB = *(const ELLC_GLOBAL u32x2_a1*)(tr.rb + off);
          const u32x2_t wC = *(const ELLC_GLOBAL u32x2_a1*)(tr.rc + off);
          const u32x2_t wD = *(const ELLC_GLOBAL u32x2_a1*)(tr.rd + off);
          const u32x2_t wE = *(const ELLC_GLOBAL u32x2_a1*)(row_e + off);
          __builtin_amdgcn_sched_barrier(0);
          // the next quad's planes, behind the row requests (vector loads return in issue order)
          pl = quad_request(K, inext, __umul24((unsigned)yn, (unsigned)sw) + 4u * (unsigned)xn);
          __builtin_amdgcn_sched_barrier(0);
          const float var[4] = {cvar.x, cvar.y, cvar.z, cvar.w};
          const uint32_t wAx = wA.x, wAy = wA.y, wBx = wB.x, wBy = wB.y, wCx = wC.x, wCy = wC.y, wDx = wD.x, wDy = wD.y, wEx = wE.x, wEy = wE.y;
          float pj = p0;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            // bytes s .. s + 3 of a row's window are columns x0 - 1 .. x0 + 2 of this pixel; its rows start at the window's first or second
            const float fx_ = __builtin_amdgcn_fractf(x1[j]), fy_ = __builtin_amdgcn_fractf(y1[j]);   // x - floor(x), exact for x >= 1
            const uint32_t s1 = (uint32_t)(x0[j] - xs);            // 0 .. 4 where the quad fits
            const uint32_t sel = __builtin_amdgcn_perm(s1, s1, 0u) + 0x03020100u;   // (byte 0 of s1 four times: s .. s + 3)
            const bool dy = (y0[j] != ys);
            const uint32_t r0 = __builtin_amdgcn_perm(wAy, wAx, sel), r1 = __builtin_amdgcn_perm(wBy, wBx, sel), r2 = __builtin_amdgcn_perm(wCy, wCx, sel);
            const uint32_t r3 = __builtin_amdgcn_perm(wDy, wDx, sel), r4 = __builtin_amdgcn_perm(wEy, wEx, sel);
            const uint32_t qa = dy ? r1 : r0, qb = dy ? r2 : r1, qc = dy ? r3 : r2, qd = dy ? r4 : r3;
            // the taps (tap_finish_f's interior branch)
            const float Pbb = cvt_ubyte<1>(qb), Pbc = cvt_ubyte<2>(qb), Pcb = cvt_ubyte<1>(qc), Pcc = cvt_ubyte<2>(qc);
            const float top = __builtin_fmaf(fx_, Pbc - Pbb, Pbb);
            const float btm = __builtin_fmaf(fx_, Pcc - Pcb, Pcb);
            const float I = __builtin_fmaf(fy_, btm - top, top);
            const float Pba = cvt_ubyte<0>(qb), Pbd = cvt_ubyte<3>(qb), Pca = cvt_ubyte<0>(qc), Pcd = cvt_ubyte<3>(qc);
            const float Pab = cvt_ubyte<1>(qa), Pac = cvt_ubyte<2>(qa), Pdb = cvt_ubyte<1>(qd), Pdc = cvt_ubyte<2>(qd);
            const float g00 = Pbc - Pba, g01 = Pbd - Pbb, g10 = Pcc - Pca, g11 = Pcd - Pcb;   // twice the central differences
            float t2 = __builtin_fmaf(fx_, g01 - g00, g00);
            float b2 = __builtin_fmaf(fx_, g11 - g10, g10);
            const float gx = __builtin_fmaf(fy_, b2 - t2, t2);   // TWICE the gradient
            const float h00 = Pcb - Pab, h01 = Pcc - Pac, h10 = Pdb - Pbb, h11 = Pdc - Pbc;
            t2 = __builtin_fmaf(fx_, h01 - h00, h00);
            b2 = __builtin_fmaf(fx_, h11 - h10, h10);
            const float gy = __builtin_fmaf(fy_, b2 - t2, t2);
            // Jacobian row, residual, weight (fcaf_stage_b; interior, so no out-of-bounds case)
            FcaPix o;
            const float A = v_hfx * gx, B = v_hfy * gy;
            const float T = __builtin_fmaf(A, pj, B * qq);
            const float d = dd[j];
            o.J[0] = __builtin_fmaf(qq, T, B);   // -J[0]
            o.J[1] = __builtin_fmaf(pj, T, A);
            o.J[2] = __builtin_fmaf(B, pj, -(A * qq));
            o.J[3] = A * d;
            o.J[4] = B * d;
            o.J[5] = d * T;                      // -J[5]
            const float Ikf = (j == 0) ? cvt_ubyte<0>(cI) : (j == 1) ? cvt_ubyte<1>(cI) : (j == 2) ? cvt_ubyte<2>(cI) : cvt_ubyte<3>(cI);
            const float res = I - Ikf;
            // d(residual)/d(inverse depth) = (gx2 n0 + gy2 n1) rz^2 / 2 with n0 = P3 pz - P11 px = pz (P3 - P11 x1), n1 likewise, pz rz = 1:
            const float drpdd = hrz[j] * __builtin_fmaf(gx, __builtin_fmaf(-vP11, x1[j], vP3), gy * __builtin_fmaf(-vP11, y1[j], vP7));
            const float D = __builtin_fmaf(var[j] * drpdd, drpdd, 16.0f);
            const float r = __builtin_amdgcn_rsqf(D);
            o.wgt = __builtin_amdgcn_fmed3f(r * r, r * (1.5f * __builtin_amdgcn_rcpf(fabsf(res))), 0.0f);
            o.residual = res;
            if (active && fit && vv[j]) fca_accumulate_pixel(acc, o);
            pj += v_rfx;
            // two pixels at a time: the compiler may interleave the second halves of a pair (dependent arithmetic of one fills the
            // latencies of the other: 339.5 against 344.0 us at C4, three interleaved runs), not all four (registers)
            if (j == 1) __builtin_amdgcn_sched_barrier(0);
          }
          // a quad that does not fit: its pixels join the wave's queue
          const bool nofit = active && !fit;
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(nofit) != 0ull, 0)) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              // (tap_general: a point with x < 0, x >= cols, y < 0 or y >= rows — or a NaN — has none of its four taps in bounds and adds
              // nothing to any sum: not queued)
              const bool inside = (x1[j] >= 0.0f) & (x1[j] < (float)cols) & (y1[j] >= 0.0f) & (y1[j] < (float)g.rows);
              const bool push = nofit && vv[j] && inside;
              const unsigned long long m = __builtin_amdgcn_ballot_w64(push);
              if (push) q[qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = i4 + (unsigned)j;
              qn += __builtin_popcountll(m);
            }
#ifdef ELLC_QUAD_STATS
            { const unsigned long long mn = __builtin_amdgcn_ballot_w64(nofit); if (lane == 0) atomicAdd(&g_quad_stats[1], (unsigned long long)__builtin_popcountll(mn)); }
#endif
          }
#ifdef ELLC_QUAD_STATS
          { const unsigned long long ma = __builtin_amdgcn_ballot_w64(active); if (lane == 0) atomicAdd(&g_quad_stats[0], (unsigned long long)__builtin_popcountll(ma)); }
#endif
          xq = xn; y = yn; i = in_;
        }
        // the queue: 64 at a time through the per-pixel step while the chunk lasts, then what is left
        while (__builtin_expect(qn >= (more ? 64 : 1), 0)) {   // (cold: what the register allocator must spill, it spills here)
          const int n = min(qn, 64);
          __builtin_amdgcn_wave_barrier();
          const unsigned idx = q[qn - n + min(lane, n - 1)];   // (a lane without an entry runs on a copy of the last one and accumulates nothing)
          __builtin_amdgcn_wave_barrier();
          qn -= n;
          int yy = (int)(((float)idx + 0.5f) * (1.0f / (float)cols));
          if (yy * cols > (int)idx) yy--;
          if ((yy + 1) * cols <= (int)idx) yy++;
          const int xx = (int)idx - yy * cols;
          const DensePix dp = dense_request(K, idx, (unsigned)(yy * sw + xx));
          bool valid;
          const FcaInF rec = dense_form(g, dp, xx, yy, valid);
          const FcafConst fc = fcaf_const(g, S);   // (vector copies of the constants, made here: the main loop keeps them scalar)
          const FcafStage st = fcaf_stage_a(g, tr, fc, rec);
          if (lane < n && valid) fca_accumulate_pixel(acc, fcaf_stage_b<false, 0>(a, K, g, cur, fc, idx, st));
#ifdef ELLC_QUAD_STATS
          if (lane == 0) { atomicAdd(&g_quad_stats[2], (unsigned long long)n); atomicAdd(&g_quad_stats[3], 1ull); }
#endif
        }
        if (!more) break;
      }
    }
    fca_acc_unpack<true>(acc, sums);
  }
  float* out = a.partials + (size_t)(fa.seq & 1) * fa.stride_part + ((size_t)b * ELLC_NBLK_MAX + sub) * ELLC_PART_STRIDE;
  ELLC_BSTAMP(2);
  block_reduce_store<27>(sums, out);
  ELLC_BSTAMP(3);
}

// ---------------------------------------------------------------------------------------------------
// State-driven form of the fused FCA schedule, for contexts with the reference's early exit on (ImageFunc.cpp:251-252). A
// launch of gn_fca_fused is bound to a level when the schedule is captured: with early exit most of the 32 launches find
// their level already ended and return at once, but each still costs a dependent launch (about 4.5 us; 17 of 32 for a
// tracked frame). Here the LEVEL comes from the state record: launch n solves the pending sums of the record's level,
// decides — level ended by weightedPose < 1 or by its iteration cap — whether the pixel pass that follows belongs to the same
// level or the next finer one, and runs it; once the last level has ended the remaining launches only carry the record
// forward. A captured graph is therefore as long as alignments usually need, not as long as the caps allow; the host looks
// at the exported records and, if an alignment has not ended, replays a continuation graph (ellc_align_fetch). Per
// alignment the sequence of pixel passes and solves — and so every bit of the result — is that of the level-bound schedule.
//   The grid is (max over levels of the level's block count, B); a block stays if the record's level or the next finer one
// needs it (the blocks of the finer level take part in the solve: only its result says whether they are needed). The
// pending partial sums are requested before the level is known (partial_preload against the grid's block count); the level's
// table entries start with the state record's arrival and are read again only at a level change.
// (exact mode: three waves per SIMD — with the level change in the kernel the exact pixel loop needs a few registers more than
// the 128 that four waves leave, and spills cost more than the fourth wave gives)
template <bool DIVC, bool FAST, int SAVEW>
__global__ __launch_bounds__(ELLC_GN_THREADS, FAST ? 4 : 3) void gn_fca_adaptive(const AlignState* src_state, const float* prev_part, int nblk_grid,
                                                                      FusedArgs fa) {
  const GnArgs& a = fa.g;
  int b = blockIdx.y, sub = blockIdx.x;
  if (fa.xcd_map) {   // see gn_fca_fused
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int w = lin >> 3, bl = w / nblk_grid;
    sub = w - bl * nblk_grid;
    b = bl * 8 + (lin & 7);
  }
  const AlignState& src = src_state[b];
  AlignState* dst = a.state + (size_t)((fa.seq + 1) & 1) * fa.stride_state + b;
  __shared__ SolveShared sh;
  const int t = threadIdx.x;
  const bool writer = (sub == 0);
  const float* pend = prev_part + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE;
  float pv[8];
  partial_preload(pend, min(nblk_grid, 8 * (ELLC_SOLVE_THREADS / 32)), pv);
  const int slot = a.kf_slot[b], frs = a.fr_slot[b];
  const int lvl = src.cur_level;
  const int pending = src.pending;
  if (lvl < 0) {   // the schedule of this alignment has ended: the record moves to the other buffer unchanged
    if (writer) {
      if (t < 6) dst->pose[t] = src.pose[t];
      if (t < 12) dst->S[t] = src.S[t];
      if (t < ELLC_MAX_LEVELS) dst->iters[t] = src.iters[t];
      // -1: ended, saved weights still to be added by the gn_add_saved_weights_all behind this graph; -2: ended in an earlier
      // graph of the schedule, which added them (the first launch of a continuation marks the records it finds ended)
      if (t == 0) { dst->weighted = src.weighted; dst->level_done = src.level_done; dst->pending = 0;
                    dst->cur_level = (fa.continuation && fa.seq == 0) ? -2 : lvl; dst->it_in_level = 0; }
    }
    return;
  }
  const int nb_l = fa.nblk_lv[lvl];
  if (sub >= max(nb_l, lvl > 0 ? fa.nblk_lv[lvl - 1] : 0)) return;   // needed neither at this level nor at the next (block 0 always is)
  // speculatively the tables of the record's level (the pixel pass stays there unless the solve ends the level)
  LevelGeom g = a.geom[lvl];
  KfLevelDev K = a.kf_tab[lvl * a.max_kf + slot];
  const FrLevelDev* F = &a.fr_tab[lvl * a.max_fr + frs];
  int V = *as_global(K.count);
  const double group_sum = partial_group_sum_from(pend, pv, nb_l);
  int begin, end;
  {
    const int chunk = (V + nb_l - 1) / nb_l;
    begin = sub * chunk;
    end = min(V, begin + chunk);
  }
  // this thread's first record (exact mode: and its pose-independent products), requested before the solve
  FcaIn first = fca_in_empty();
  FcaInF firstf = fcaf_empty();
  FcaPre first_pre;
  if constexpr (FAST) {
    if (sub < nb_l && begin < end) firstf = fcaf_load(K, (unsigned)min(begin + t, end - 1));
  } else {
    if (sub < nb_l && begin + t < end) first = fca_load<DIVC>(K, g, (unsigned)(begin + t));
    first_pre = fca_prepare<DIVC>(g, first);
    asm volatile("" ::"v"(first_pre.c_t0), "v"(first_pre.c_b1), "v"(first_pre.d), "v"(first_pre.fxz), "v"(first_pre.fyz),
                 "v"(first_pre.nvz), "v"(first_pre.nuz));   // pinned above the solve, see gn_fca_fused
  }
  if (pending) {
    solve_step<FAST>(sh, group_sum, 0, lvl, fa.early_exit, src, nullptr);
  } else {
    if (t < 6) sh.newpose[t] = src.pose[t];
    if (t < 12) sh.newS[t] = src.S[t];
    if (t == 0) { sh.weighted = src.weighted; sh.level_done = src.level_done; }
    __syncthreads();
  }
  const int it = src.it_in_level + (pending ? 1 : 0);
  const bool over = pending && (sh.level_done == lvl || it >= fa.max_it[lvl]);   // the level has ended: early exit, or its cap
  const int nl = over ? lvl - 1 : lvl;
  if (writer) {
    if (t < 6) dst->pose[t] = sh.newpose[t];
    if (t < 12) dst->S[t] = sh.newS[t];
    if (t < ELLC_MAX_LEVELS) dst->iters[t] = src.iters[t] + ((pending && t == lvl) ? 1 : 0);
    if (t == 0) {
      dst->weighted = sh.weighted;
      dst->level_done = sh.level_done;
      dst->pending = nl >= 0 ? 1 : 0;
      dst->cur_level = nl;
      dst->it_in_level = over ? 0 : it;
    }
  }
  if (nl < 0) return;
  if (over) {   // a level change (at most L - 1 per alignment): tables, chunk and first record of the finer level
    const int nb_n = fa.nblk_lv[nl];
    if (sub >= nb_n) return;
    g = a.geom[nl];
    K = a.kf_tab[nl * a.max_kf + slot];
    F = &a.fr_tab[nl * a.max_fr + frs];
    V = *as_global(K.count);
    const int chunk = (V + nb_n - 1) / nb_n;
    begin = sub * chunk;
    end = min(V, begin + chunk);
    if constexpr (FAST) {
      if (begin < end) firstf = fcaf_load(K, (unsigned)min(begin + t, end - 1));   // (a thread past the chunk's end starts on a copy of its last record: the pixel loop is block-uniform)
    } else {
      if (begin + t < end) first = fca_load<DIVC>(K, g, (unsigned)(begin + t));
      first_pre = fca_prepare<DIVC>(g, first);
    }
  } else if (sub >= nb_l) {
    return;
  }
  g_u8 cur = as_global(F->img);
  float sums[27];
  fca_chunk_pass<DIVC, true, FAST, SAVEW>(a, K, g, cur, sh.newS, begin, end, first, firstf, first_pre, sums);
  float* out = a.partials + (size_t)(fa.seq & 1) * fa.stride_part + ((size_t)b * ELLC_NBLK_MAX + sub) * ELLC_PART_STRIDE;
  block_reduce_store<27>(sums, out);
}

// ---------------------------------------------------------------------------------------------------
// The state-driven schedule as ONE launch (r05): the tracking call's alignment is ~15 dependent iterations of 1-3 us of work each,
// and as launches every one of them pays the kernel boundary and a first memory round trip into caches the boundary emptied
// (tools/dbg/seq_stamps.py, tools/stamps.py). Here G blocks per alignment stay resident for the whole schedule; an iteration is
// gn_fca_adaptive's body — combine the pending partial sums, solve, decide the level, pixel pass over the block's chunk, 27 sums
// per block. There is NO barrier: the only thing that crosses blocks is the partial records, and a record carries the round it
// belongs to. A block stores its record as ONE 128-byte line — 27 sums and, in the last word of each of the line's four 32-byte
// sectors, the tag (call epoch << 8 | round) — with agent-scope write-through stores from 32 consecutive lanes; a reader loads
// whole records with agent-scope loads (past its XCD's L2: the XCDs' L2s are not coherent with each other, and an agent-scope
// FENCE would write the L2 back — tools/micro/grid_barrier.hip) and takes a record when all four tags say the round it waits
// for, else asks again. Two memory hops per iteration (store becomes visible, load returns) where a counter barrier needs four
// (stores acknowledged, arrival, poll, loads): 0.131 -> 0.122 ms per early-exit alignment with the barrier form, -> 0.114-0.118 with
// this one. The two record buffers alternate by round: a block writes round r + 2 into the buffer of round r only after it has
// read every record of round r + 1, which exist only once every block has read round r. The alignment's state record is not
// shared at all: every block keeps its own copy in LDS and advances it by the same solve on the same sums, so all blocks take
// the same level changes and leave the loop in the same iteration; block 0 then does what gn_fused_finish does (the final
// record, the result for the host, the observation's matrices and the depth stages' gate): no second launch. A wave that has asked ELLC_PERSIST_SPIN_LIMIT times for a record that does not come (a launch whose blocks
// are not all resident: the device shared with more such launches than it holds) raises the abort word (it names the call: nothing
// has to clear it): every block leaves, the
// record says "not ended, nothing pending", and the host finishes the schedule with ordinary launches (the continuation of the
// state-driven schedule). Block counts per level, chunks and the order of the combine are the launch-per-iteration schedule's
// (gn_fca_adaptive): the same bits, also for a schedule that is abandoned here and finished there. Three blocks per CU in
// the tolerance mode, two in the exact mode (launch bounds; the exact pixel loop inside this loop wants 226 registers and spills
// cost it more than the launches it saves): 768 / 512 resident blocks hold three / two such launches of one alignment each
// (r06: an alignment takes at most 128 blocks — every block reads every block's record, choose_nblk — so six / four).
#define ELLC_PERSIST_BAR_WORDS 64   // per alignment: the abort word in a 128-byte line of its own, then the state line (below)
#define ELLC_PERSIST_STATE_WORD 32  //   first word of the state line
#define ELLC_PERSIST_SPIN_LIMIT (1u << 15)   // polls (~1 us each with their s_sleep): a record normally arrives within tens
// word of a tagged record that holds sum s (the last word of every 32-byte sector is the tag)
__device__ __forceinline__ int persist_word_of(int s) { return s + s / 7; }
// block reduction of the 27 per-thread sums (as block_reduce_store) and the block's tagged record, stored by lanes 0..31 of wave 0
__device__ __forceinline__ void persist_store_record(float (&acc)[27], unsigned* out, unsigned tag) {
  __shared__ float red[ELLC_GN_THREADS / 64][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float rows[WaveRows<27>::N2];
  wave_sum_rows<27>(acc, rows);
  ELLC_PTRACE(8, 0);
  if ((lane & 15) == 0) {
    const int q = lane >> 4, col = 2 * (q & 1) + (q >> 1);
#pragma unroll
    for (int j = 0; j < WaveRows<27>::N2; j++) red[wave][4 * j + col] = rows[j];
  }
  __syncthreads();
  ELLC_PTRACE(9, 0);
  if (threadIdx.x < 32) {
    const int w = (int)threadIdx.x, sidx = w - (w >> 3);
    const bool tagl = (w & 7) == 7;
    float v = 0.0f;
    if (!tagl && sidx < 27) {
      v = red[0][sidx];
#pragma unroll
      for (int k = 1; k < ELLC_GN_THREADS / 64; k++) v += red[k][sidx];
    }
    __hip_atomic_store(out + w, tagl ? tag : __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// partial_group_sum over tagged records of the round `tag` names: the same fixed-order combine, every record taken only once
// its four tags match. Returns PERSIST_OK, PERSIST_ABANDON when the launch is being abandoned, or PERSIST_LAPPED when a record's slot
// already holds a LATER round of this call (same buffer, so two rounds on): the writers have gone on without this block, which can
// only happen to a block that writes nothing at the level (nobody waits for its records) — see the state line in gn_fca_persist.
enum { PERSIST_OK = 0, PERSIST_ABANDON = 1, PERSIST_LAPPED = 2 };
#ifdef ELLC_DIAG_ABI
__device__ unsigned long long g_persist_adoptions;   // blocks that were lapped and re-joined through the state line (ellc_debug_persist_counters)
#endif
__device__ __forceinline__ int persist_group_sum(const unsigned* recs, int nblk, unsigned tag, unsigned* abortw, unsigned abort_tag, unsigned spin_limit, double& out) {
  const int lane = threadIdx.x & 63, comp = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const bool is_tag = (comp & 7) == 7;
  const unsigned long long half = (lane < 32) ? 0x00000000ffffffffull : 0xffffffff00000000ull;
  const int src_lane = (lane & 32) | (comp < 27 ? persist_word_of(comp) : 31);
  double s = 0.0;
  unsigned spins = 0;
  if (spin_limit == 0u) {   // test hook (ellc_debug_persist_spin_limit): abandon at the first record
    if (lane == 0) __hip_atomic_store(abortw, abort_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out = 0.0;
    return PERSIST_ABANDON;
  }
  for (int base = 0; base < nblk; base += 8 * (ELLC_SOLVE_THREADS / 32)) {
    unsigned w[8];
    unsigned okm = 0;   // bit j: record j of this thread's half-wave has been taken (or is not needed)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      w[j] = 0u;
      if (base + grp + j * (ELLC_SOLVE_THREADS / 32) >= nblk) okm |= 1u << j;
    }
    for (;;) {
#pragma unroll
      for (int j = 0; j < 8; j++) {   // every record still missing is asked for again, all requests in flight together
        const int k = base + grp + j * (ELLC_SOLVE_THREADS / 32);
        if (!((okm >> j) & 1u)) w[j] = __hip_atomic_load(recs + (size_t)k * ELLC_PART_STRIDE + comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool miss = !((okm >> j) & 1u);
        const unsigned long long badm = __ballot(miss && is_tag && w[j] != tag);
        if (miss && (badm & half) == 0ull) okm |= 1u << j;
      }
      if (__ballot(okm != 0xffu) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      spins++;
      if ((spins & 7u) == 0u) {   // (wave-uniform; off the path of a record that arrives in time) a slot that holds a later round of this call?
        bool later = false;
#pragma unroll
        for (int j = 0; j < 8; j++) later |= !((okm >> j) & 1u) && is_tag && (w[j] >> 8) == (tag >> 8) && (w[j] & 0xffu) > (tag & 0xffu);
        if (__ballot(later) != 0ull) { out = 0.0; return PERSIST_LAPPED; }
      }
      if ((spins & 63u) == 0u) {   // (wave-uniform)
        if (spins > spin_limit && lane == 0) __hip_atomic_store(abortw, abort_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == abort_tag) { out = 0.0; return PERSIST_ABANDON; }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = base + grp + j * (ELLC_SOLVE_THREADS / 32);
      const unsigned v = (unsigned)__shfl((int)w[j], src_lane, 64);
      s += (k < nblk && comp < 27) ? (double)__builtin_bit_cast(float, v) : 0.0;
    }
  }
  out = s;
  return PERSIST_OK;
}

// The state line of a resident launch: whenever a level ends, block 0 of the alignment publishes what every block's copy of the state
// record holds at that point — pose, exp(pose), weightedPose, level_done, the round and the level that begins (-1: the schedule has
// ended) — as one tagged 128-byte record (tag = call epoch << 8 | round, as the partial records). Nobody reads it in the normal
// course of a launch. It is what a block that was LAPPED re-joins by: a block that writes no records at the current level (its
// index is beyond the level's block count) is waited for by nobody, so if it is dispatched late or held up for two rounds the
// records it wants are overwritten (r05 shipped that hole: such a block spun to the poll limit and abandoned the launch). It now
// waits here for the next level's beginning instead, adopts the state and carries on — as a writer if the new level has work for
// it: the writers of that level cannot pass its first round without its record, so the line it needs cannot be overwritten before
// it has read it. Returns false when the launch is being abandoned.
__device__ __forceinline__ void persist_publish_state(unsigned* line, unsigned tag, const SolveShared& sh, int seq, int next_level) {
  if (threadIdx.x < 32) {
    const int w = (int)threadIdx.x, sidx = w - (w >> 3);   // payload index of word w (the last word of every 32-byte sector is the tag)
    unsigned v = 0u;
    if ((w & 7) == 7) v = tag;
    else if (sidx < 6) v = __builtin_bit_cast(unsigned, sh.newpose[sidx]);
    else if (sidx < 18) v = __builtin_bit_cast(unsigned, sh.newS[sidx - 6]);
    else if (sidx == 18) v = __builtin_bit_cast(unsigned, sh.weighted);
    else if (sidx == 19) v = (unsigned)sh.level_done;
    else if (sidx == 20) v = (unsigned)seq;
    else if (sidx == 21) v = (unsigned)next_level;
    __hip_atomic_store(line + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// wave 0 of a lapped block: waits for a state line of this call whose round is at least `seq`, leaves it in sh / s_seq / s_level
__device__ __forceinline__ bool persist_adopt_state(const unsigned* line, unsigned epoch, int seq, unsigned* abortw, unsigned abort_tag, unsigned spin_limit,
                                                    SolveShared& sh, int& s_seq, int& s_level) {
  const int lane = threadIdx.x & 63, w = lane & 31, sidx = w - (w >> 3);
  for (unsigned spins = 0;; spins++) {
    const unsigned v = __hip_atomic_load(line + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool is_tag = (w & 7) == 7;
    const unsigned t0 = (unsigned)__shfl((int)v, 7, 64);
    const bool fresh = (t0 >> 8) == epoch && (int)(t0 & 0xffu) >= seq;
    if (fresh && __ballot(is_tag && v != t0) == 0ull) {   // four equal tags of this call, not older than the round this block stands in
      if (lane < 32 && !is_tag) {
        if (sidx < 6) sh.newpose[sidx] = __builtin_bit_cast(float, v);
        else if (sidx < 18) sh.newS[sidx - 6] = __builtin_bit_cast(float, v);
        else if (sidx == 18) sh.weighted = __builtin_bit_cast(float, v);
        else if (sidx == 19) sh.level_done = (int)v;
        else if (sidx == 20) s_seq = (int)v;
        else if (sidx == 21) s_level = (int)v;
      }
      return true;
    }
    __builtin_amdgcn_s_sleep(2);
    if ((spins & 63u) == 63u) {
      if (spins > spin_limit && lane == 0) __hip_atomic_store(abortw, abort_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == abort_tag) return false;
    }
  }
}

template <bool FAST, bool ADAPT>
__device__ __forceinline__ void fused_finish_body(const FusedArgs& fa, int b, const AlignState& src, AlignState* dst, SolveShared& sh);

// The staging of a tracking call folded into its resident launch (r06): when the call's lists are already there (built behind the
// depth map's export) the staging kernel's only work is the batch description (two slots, the initial pose) and the state records
// — a 5 us launch and a kernel boundary in front of the alignment of every tracked frame. With `on` every block builds its copy of
// the state record itself from the kernel arguments (exp(pose) by nine lanes, exactly stage_in_args's operations), block 0 of each
// alignment writes the device-side batch description for the kernels behind the launch, and ellc_track_frame's count of the valid
// hypotheses rides in `count_blocks` extra blocks of this launch (blockIdx.x >= the persist blocks, first alignment) instead.
struct StageSmall {
  int kf[2], fr[2], uniq[2];
  float pose[12];
};
struct PersistStage {
  int on;
  StageSmall s;
  int* dst;            // the device-side batch description (kf slots | frame slots | unique slots | initial poses), `cap` apart
  int cap, top_level;
  int persist_blocks;  // blocks per alignment of the schedule itself (the grid's x extent may be larger: count blocks)
  const uint8_t* count_valid;
  int count_n, count_blocks;
  int* count_acc;
  int* count_host;
};
__device__ void dm_count_valid_body(const uint8_t* valid, int n, int* acc, int* host_visible, int block, int nblocks);

template <bool DIVC, bool FAST, int SAVEW>
__global__ __launch_bounds__(ELLC_GN_THREADS, FAST ? 3 : 2) void gn_fca_persist(FusedArgs fa, int max_rounds, unsigned epoch, unsigned spin_limit,
                                                                              int delay_from, int delay_polls, PersistStage ps) {
  const GnArgs& a = fa.g;
  const int b = blockIdx.y, sub = blockIdx.x, t = threadIdx.x;
  if (ps.on && sub >= ps.persist_blocks) {   // block-uniform: a count block (see PersistStage)
    if (b == 0 && sub - ps.persist_blocks < ps.count_blocks)
      dm_count_valid_body(ps.count_valid, ps.count_n, ps.count_acc, ps.count_host, sub - ps.persist_blocks, ps.count_blocks);
    return;
  }
  __shared__ SolveShared sh;
  __shared__ AlignState st;   // this block's copy of the alignment's record (see above)
  __shared__ int s_flag, s_seq, s_level;
  unsigned* abortw = fa.persist_bar + (size_t)b * ELLC_PERSIST_BAR_WORDS;
  unsigned* state_line = abortw + ELLC_PERSIST_STATE_WORD;
  if (delay_polls > 0 && sub >= delay_from)   // test hook (ellc_debug_persist_delay): these blocks start late, as if dispatched late
    for (int i = 0; i < delay_polls; i++) __builtin_amdgcn_s_sleep(32);
  AlignState* rec = a.state + b;   // buffer 0: initialised by the staging kernel; the final record for gn_fused_finish
  const bool writer = (sub == 0);
  if (ps.on) {   // (block-uniform) the record the staging kernel would have left, built here
    uint32_t* dp = (uint32_t*)&st;
    for (int i = t; i < (int)(sizeof(AlignState) / 4); i += ELLC_GN_THREADS) dp[i] = 0u;   // delta, b, H, Hinv, iters, pending, it_in_level
    __syncthreads();
    if (t < 16) {   // stage_in_args's lanes of one alignment
      const int l = t, l9 = min(l, 8), r3 = l9 / 3, k3 = l9 - 3 * r3;
      const float* p = ps.s.pose + min(b, 1) * 6;
      double Rrk, Vv;
      exp_se3_entry((double)p[0], (double)p[1], (double)p[2], (double)p[3], (double)p[4], (double)p[5], r3, k3, Rrk, Vv);
      const double trow = (Vv + __shfl_down(Vv, 1)) + __shfl_down(Vv, 2);
      if (l < 9) {
        st.S[r3 * 4 + k3] = (float)Rrk;
        if (k3 == 0) st.S[r3 * 4 + 3] = (float)trow;
      }
      if (l < 6) st.pose[l] = p[l];
      if (l == 9) { st.weighted = 0.0f; st.level_done = -1; st.pending = 0; st.cur_level = ps.top_level; st.it_in_level = 0; }
    }
    if (writer && t >= 64 && t < 64 + 16) {   // the batch description for the kernels behind this launch (saved weights, ...)
      const int k = t - 64;
      if (k == 0) { ps.dst[b] = ps.s.kf[min(b, 1)]; ps.dst[ps.cap + b] = ps.s.fr[min(b, 1)]; }
      if (k < 6) ((float*)(ps.dst + 3 * ps.cap))[b * 6 + k] = ps.s.pose[min(b, 1) * 6 + k];
    }
  } else {
    const uint32_t* sp = (const uint32_t*)rec;
    uint32_t* dp = (uint32_t*)&st;
    for (int i = t; i < (int)(sizeof(AlignState) / 4); i += ELLC_GN_THREADS) dp[i] = sp[i];
  }
  __syncthreads();
  // (block-uniform, said so: a slot in a vector register would drag every table entry behind it into vector registers — NOTEBOOK 6.4)
  const int slot = __builtin_amdgcn_readfirstlane(ps.on ? ps.s.kf[min(b, 1)] : a.kf_slot[b]);
  const int frs = __builtin_amdgcn_readfirstlane(ps.on ? ps.s.fr[min(b, 1)] : a.fr_slot[b]);
  if (t == 0) s_flag = 0;
  __syncthreads();
  for (int seq = 0; seq < max_rounds; seq++) {
    ELLC_PTRACE_ROUND(sub, seq);
    ELLC_PTRACE(0, 0);
    if (delay_polls < 0 && seq >= -delay_polls && st.cur_level >= 0) {   // test hook (ellc_debug_persist_delay, polls < 0): the launch is
      if (t == 0 && sub == 0) __hip_atomic_store(abortw, epoch | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // abandoned in round -polls:
      break;   // every block leaves where a block that had read the abort word in this round's gather would (block-uniform)
    }
    const int lvl = st.cur_level, pending = st.pending, it_in = st.it_in_level;
    if (lvl < 0) break;
    const int nb_l = fa.nblk_lv[lvl];
    LevelGeom g = a.geom[lvl];
    KfLevelDev K = a.kf_tab[lvl * a.max_kf + slot];
    const FrLevelDev* F = &a.fr_tab[lvl * a.max_fr + frs];
    int V = *as_global(K.count);
    const unsigned* pend = (const unsigned*)(a.partials + (size_t)((seq + 1) & 1) * fa.stride_part + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE);
    int begin, end;
    {
      const int chunk = (V + nb_l - 1) / nb_l;
      begin = sub * chunk;
      end = min(V, begin + chunk);
    }
    // this thread's first record of the level the record names, requested before the solve (as gn_fca_adaptive)
    FcaIn first = fca_in_empty();
    FcaInF firstf = fcaf_empty();
    FcaPre first_pre;
    if constexpr (FAST) {
      if (sub < nb_l && begin < end) firstf = fcaf_load(K, (unsigned)min(begin + t, end - 1));
    } else {
      if (sub < nb_l && begin + t < end) first = fca_load<DIVC>(K, g, (unsigned)(begin + t));
      first_pre = fca_prepare<DIVC>(g, first);
    }
    bool adopted = false;
    ELLC_PTRACE(1, 0);
    ELLC_PTRACE(11, (lvl << 8) | (sub < nb_l ? 1 : 0));
    if (pending) {
      double group_sum;
      const int got = persist_group_sum(pend, nb_l, (epoch << 8) | (unsigned)seq, abortw, epoch | 0x80000000u, spin_limit, group_sum);   // the records of round seq (the previous iteration's)
      if (got != PERSIST_OK && (t & 63) == 0) atomicMax(&s_flag, got == PERSIST_LAPPED && sub >= nb_l ? 2 : 1);   // (a writer cannot be lapped: treated as a reason to abandon)
      ELLC_PTRACE(2, 0);
      __syncthreads();
      if (s_flag == 1) break;   // abandoned (block-uniform); the record still names this iteration's level with its sums unsolved
      if (s_flag == 2) {        // lapped (block-uniform; this block writes nothing at this level): re-join at the next level's beginning
        if (t < 64 && !persist_adopt_state(state_line, epoch, seq, abortw, epoch | 0x80000000u, spin_limit, sh, s_seq, s_level)) s_flag = 1;
        __syncthreads();
        if (s_flag == 1) break;
        adopted = true;
#ifdef ELLC_DIAG_ABI
        if (t == 0) atomicAdd(&g_persist_adoptions, 1ull);   // (diagnostic library: tests assert that the path was taken)
#endif
        seq = s_seq;
        __syncthreads();
        if (t == 0) s_flag = 0;
      } else {
        solve_step<FAST>(sh, group_sum, 0, lvl, fa.early_exit, st, writer ? rec : nullptr);
      }
    } else {
      if (t < 6) sh.newpose[t] = st.pose[t];
      if (t < 12) sh.newS[t] = st.S[t];
      if (t == 0) { sh.weighted = st.weighted; sh.level_done = st.level_done; }
      __syncthreads();
    }
    ELLC_PTRACE(3, 0);
    const int it = it_in + (pending ? 1 : 0);
    const bool over = adopted || (pending && (sh.level_done == lvl || it >= fa.max_it[lvl]));   // the level has ended: early exit, or its cap
    const int nl = adopted ? s_level : (over ? lvl - 1 : lvl);
    if (writer && over) persist_publish_state(state_line, (epoch << 8) | (unsigned)seq, sh, seq, nl);   // (see persist_publish_state)
    // (every read of `st` of this iteration lies in front of the barrier that ends the solve)
    if (t < 6) st.pose[t] = sh.newpose[t];
    if (t < 12) st.S[t] = sh.newS[t];
    if (t == lvl && pending) st.iters[t] += 1;
    if (t == 0) {
      st.weighted = sh.weighted;
      st.level_done = sh.level_done;
      st.pending = nl >= 0 ? 1 : 0;
      st.cur_level = nl;
      st.it_in_level = over ? 0 : it;
    }
    if (nl < 0) break;
    bool work = sub < nb_l;
    if (over) {   // a level change (at most L - 1 per alignment): tables, chunk and first record of the finer level
      const int nb_n = fa.nblk_lv[nl];
      work = sub < nb_n;
      g = a.geom[nl];
      K = a.kf_tab[nl * a.max_kf + slot];
      F = &a.fr_tab[nl * a.max_fr + frs];
      V = *as_global(K.count);
      const int chunk = (V + nb_n - 1) / nb_n;
      begin = sub * chunk;
      end = min(V, begin + chunk);
      if (work) {
        if constexpr (FAST) {
          if (begin < end) firstf = fcaf_load(K, (unsigned)min(begin + t, end - 1));
        } else {
          if (begin + t < end) first = fca_load<DIVC>(K, g, (unsigned)(begin + t));
          first_pre = fca_prepare<DIVC>(g, first);
        }
      }
    }
    if (work) {   // block-uniform
      g_u8 cur = as_global(F->img);
      float sums[27];
      fca_chunk_pass<DIVC, true, FAST, SAVEW, true>(a, K, g, cur, sh.newS, begin, end, first, firstf, first_pre, sums);
      ELLC_PTRACE(7, 0);
      unsigned* out = (unsigned*)(a.partials + (size_t)(seq & 1) * fa.stride_part + ((size_t)b * ELLC_NBLK_MAX + sub) * ELLC_PART_STRIDE);
      persist_store_record(sums, out, (epoch << 8) | (unsigned)(seq + 1));
      ELLC_PTRACE(10, 0);
    }
    __syncthreads();   // (`st` is read again at the top)
  }
  __syncthreads();
  if (writer) {   // block-uniform: what gn_fused_finish does behind the launch-per-iteration schedule, on this block's copy of the record
    // abandoned (or out of rounds, which max_rounds excludes): the sums of the round that was under way are lost — nothing
    // pending, the continuation repeats that pixel pass
    if (t == 0) st.pending = 0;
    __syncthreads();
    fused_finish_body<FAST, true>(fa, b, st, rec, sh);
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused schedule of the constant-weight path (PixelWisePyramid.cpp:687-913 + :941-974): the same launch structure as
// gn_fca_fused — launch n first solves the sums launch n-1 left behind, then runs its own pixel pass — with the light
// ICA pixel pass: warp, one u8 tap, residual, b += SD (r w). H^-1 of the level was computed once per keyframe by the
// compaction (IcaRec, ica_hinv), so the pending sums are b only and no launch is spent on the precompute.
struct IcaIn { float X, Y, Z, Ikf, W, sd[6]; };
__device__ __forceinline__ IcaIn ica_load(const IcaRec* irec, unsigned i) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const ELLC_GLOBAL f32x4* r = (const ELLC_GLOBAL f32x4*)((const ELLC_GLOBAL char*)irec + i * (unsigned)sizeof(IcaRec));
  const f32x4 a = r[0], b = r[1], c = r[2];
  IcaIn in;
  in.X = a.x; in.Y = a.y; in.Z = a.z; in.Ikf = a.w;
  in.W = b.x; in.sd[0] = b.y; in.sd[1] = b.z; in.sd[2] = b.w;
  in.sd[3] = c.x; in.sd[4] = c.y; in.sd[5] = c.z;
  return in;
}
// Tolerance mode: the pixel as ONE 16-byte record {x | y << 12 | I << 24, d = 1 / Z, saved weight, 2 gradx | 2 grady << 16} in the
// slot's crec list instead of the 48-byte IcaRec: the pass is bound by streaming its records (r02: 203 us for a level-0 launch over
// 128 alignments), and the row of the template Jacobian is a handful of multiply-adds of (A, B, p, q, d):  J = [-(q T + B),
// p T + A, B p - A q, A d, B d, -d T], T = A p + B q, A = fx gradx, B = fy grady (PixelWisePyramid.cpp:561-680 in the form
// fcaf_pixel uses). The central differences of a u8 image are multiples of one half below 256 in size: twice each is a 16-bit
// integer, exactly (r04; r03 carried A in the record and B in a plane of its own, 20 bytes and two loads per pixel — same values:
// fx (0.5 g2) is the product the compaction formed). H^-1 still comes from the compaction's exact row (ica_hinv).
struct IcaInF { uint32_t xyI; float d, W; uint32_t gxy; };
__device__ __forceinline__ IcaInF ica_load_fast(const KfLevelDev& K, unsigned i) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = *(const ELLC_GLOBAL u32x4*)((const ELLC_GLOBAL char*)K.crec + i * 16u);
  IcaInF in;
  const uint32_t w1 = v.y, w2 = v.z, w3 = v.w;   // (copied first: bit_cast of a vector element expression reads element 0)
  in.xyI = v.x; in.d = __builtin_bit_cast(float, w1); in.W = __builtin_bit_cast(float, w2); in.gxy = w3;
  return in;
}
template <bool FAST> struct IcaInOf { typedef IcaIn type; };
template <> struct IcaInOf<true> { typedef IcaInF type; };
template <bool FAST>
__device__ __forceinline__ typename IcaInOf<FAST>::type ica_load_any(const KfLevelDev& K, unsigned i) {
  if constexpr (FAST) return ica_load_fast(K, i);
  else return ica_load(K.irec, i);
}
template <bool FAST>
__device__ __forceinline__ typename IcaInOf<FAST>::type ica_in_empty() {
  typename IcaInOf<FAST>::type in;
  if constexpr (FAST) {
    in.xyI = 0; in.d = 1.0f; in.W = 0.0f; in.gxy = 0u;
  } else {
    in.X = 0.0f; in.Y = 0.0f; in.Z = 1.0f; in.Ikf = 0.0f; in.W = 0.0f;
#pragma unroll
    for (int r = 0; r < 6; r++) in.sd[r] = 0.0f;
  }
  return in;
}

template <bool FAST>
__device__ __forceinline__ void ica_accumulate_pixel(float (&acc)[6], const typename IcaInOf<FAST>::type& in, const LevelGeom& g, g_u8 cur,
                                                     const float* S) {
  if constexpr (FAST) {
    // the point divided by Z, (p, q, 1) + t d, projects to the same pixel (see fcaf_pixel): fused multiply-adds, hardware reciprocal
    const float p = __builtin_fmaf((float)(in.xyI & 0xfffu), g.rfx, -(g.cx * g.rfx));
    const float q = __builtin_fmaf((float)((in.xyI >> 12) & 0xfffu), g.rfy, -(g.cy * g.rfy));
    const float d = in.d;
    const float px = __builtin_fmaf(S[0], p, __builtin_fmaf(S[1], q, __builtin_fmaf(S[3], d, S[2])));
    const float py = __builtin_fmaf(S[4], p, __builtin_fmaf(S[5], q, __builtin_fmaf(S[7], d, S[6])));
    const float pz = __builtin_fmaf(S[8], p, __builtin_fmaf(S[9], q, __builtin_fmaf(S[11], d, S[10])));
    const float rz = __builtin_amdgcn_rcpf(pz);
    const float wx = __builtin_fmaf(px * rz, g.fx, g.cx);
    const float wy = __builtin_fmaf(py * rz, g.fy, g.cy);
    const Taps t = tap_point<false, true>(cur, g.sw, g.cols, g.rows, wx, wy);
    const bool oob = (t.I == -1.0f);
    const float residual = oob ? 0.0f : (t.I - byte_f32<3>(in.xyI));
    const float rw = residual * in.W;
    const float A = g.fx * ((float)(int)(short)(in.gxy & 0xffffu) * 0.5f), B = g.fy * ((float)((int)in.gxy >> 16) * 0.5f);   // fx gradx, fy grady
    const float T = __builtin_fmaf(A, p, B * q);
    acc[0] = __builtin_fmaf(-__builtin_fmaf(q, T, B), rw, acc[0]);
    acc[1] = __builtin_fmaf(__builtin_fmaf(p, T, A), rw, acc[1]);
    acc[2] = __builtin_fmaf(__builtin_fmaf(B, p, -(A * q)), rw, acc[2]);
    acc[3] = __builtin_fmaf(A * d, rw, acc[3]);
    acc[4] = __builtin_fmaf(B * d, rw, acc[4]);
    acc[5] = __builtin_fmaf(-(d * T), rw, acc[5]);
  } else {
    const Warp w = warp_point<true>(in.X, in.Y, in.Z, g, S);
    const Taps t = tap_point<false, false>(cur, g.sw, g.cols, g.rows, w.wx, w.wy);
    const bool oob = (t.I == -1.0f);
    const float residual = oob ? 0.0f : (t.I - in.Ikf);
    const float rw = residual * in.W;
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] = __builtin_fmaf(in.sd[r], rw, acc[r]);
  }
}

template <bool FAST>
__global__ __launch_bounds__(ELLC_GN_THREADS) void gn_ica_fused(const AlignState* src_state, const float* prev_part, int prev_nblk, FusedArgs fa) {
  const GnArgs& a = fa.g;   // leading scalars: preloaded kernel arguments, see gn_fca_fused
  const int b = blockIdx.y, sub = blockIdx.x;
  const AlignState& src = src_state[b];
  AlignState* dst = a.state + (size_t)((fa.seq + 1) & 1) * fa.stride_state + b;
  __shared__ SolveShared sh;
  const int t = threadIdx.x;
  const bool writer = (sub == 0);
  const LevelGeom g = a.geom[a.level];
  const int slot = a.kf_slot[b];
  const KfLevelDev K = a.kf_tab[a.level * a.max_kf + slot];   // by value: uniform, lives in SGPRs
  const FrLevelDev& F = a.fr_tab[a.level * a.max_fr + a.fr_slot[b]];
  const int pending = src.pending;
  const int V = *as_global(K.count);
  const double group_sum = partial_group_sum(prev_part + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE, prev_nblk);
  const int chunk = (V + a.nblk - 1) / a.nblk;
  const int begin = sub * chunk;
  const int end = min(V, begin + chunk);
  g_u8 cur = as_global(F.img);
  typename IcaInOf<FAST>::type first = ica_in_empty<FAST>();
  if (begin + t < end) first = ica_load_any<FAST>(K, (unsigned)(begin + t));
  if (pending) {
    const float* hinv = a.kf_tab[fa.prev_level * a.max_kf + slot].hinv;
    solve_step<FAST>(sh, group_sum, 2, fa.prev_level, fa.early_exit, src, nullptr, hinv);
  } else {
    if (t < 6) sh.newpose[t] = src.pose[t];
    if (t < 12) sh.newS[t] = src.S[t];
    if (t == 0) { sh.weighted = src.weighted; sh.level_done = src.level_done; }
    __syncthreads();
  }
  const int level_done = sh.level_done;
  const bool skip = (level_done == a.level);
  if (writer) {
    if (t < 6) dst->pose[t] = sh.newpose[t];
    if (t < 12) dst->S[t] = sh.newS[t];
    if (t < ELLC_MAX_LEVELS) dst->iters[t] = src.iters[t] + ((pending && t == fa.prev_level) ? 1 : 0);
    if (t == 0) {
      dst->weighted = sh.weighted;
      dst->level_done = level_done;
      dst->pending = skip ? 0 : 1;
    }
  }
  if (skip) return;
  float S[12];
#pragma unroll
  for (int i = 0; i < 12; i++) S[i] = sh.newS[i];
  float acc[6];
#pragma unroll
  for (int i = 0; i < 6; i++) acc[i] = 0.0f;
  int i = begin + t;
  if (i < end) {
    ica_accumulate_pixel<FAST>(acc, first, g, cur, S);
    for (i += ELLC_GN_THREADS; i < end; i += ELLC_GN_THREADS) ica_accumulate_pixel<FAST>(acc, ica_load_any<FAST>(K, (unsigned)i), g, cur, S);
  }
  float* out = a.partials + (size_t)(fa.seq & 1) * fa.stride_part + ((size_t)b * ELLC_NBLK_MAX + sub) * ELLC_PART_STRIDE;
  block_reduce_store<6>(acc, out + 21);   // the b slots of the partial record; the H slots are not read by a mode-2 solve
}

// Final solve of a fused schedule: consumes the last pending partials; result always lands in state buffer 0. (The body of
// gn_fused_finish; gn_fca_persist's first block runs it itself at the end of its launch, on its own copy of the record.)
template <bool FAST, bool ADAPT>
__device__ __forceinline__ void fused_finish_body(const FusedArgs& fa, int b, const AlignState& src, AlignState* dst, SolveShared& sh) {
  const GnArgs& a = fa.g;
  const int t = threadIdx.x;
  // state-driven schedule (ADAPT, gn_fca_adaptive): the level of the pending sums comes from the record, and the schedule
  // may not have ended yet — the result record then says so (pad = 1) and the host replays a continuation
  const int lvl = ADAPT ? src.cur_level : fa.prev_level;
  const int pending = ADAPT ? (src.pending && lvl >= 0) : src.pending;
  const int it_in = ADAPT ? src.it_in_level : 0;
  __shared__ int it_copy[ELLC_MAX_LEVELS];
  if (t < ELLC_MAX_LEVELS) it_copy[t] = src.iters[t];
  if (pending) {
    const float* prev = a.partials + (size_t)((fa.seq + 1) & 1) * fa.stride_part + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE;
    const float* hinv = fa.ica ? a.kf_tab[lvl * a.max_kf + a.kf_slot[b]].hinv : nullptr;
    solve_step<FAST>(sh, partial_group_sum(prev, ADAPT ? fa.nblk_lv[lvl] : fa.prev_nblk), fa.ica ? 2 : 0, lvl, fa.early_exit, src, dst, hinv);
  } else {
    if (t < 6) sh.newpose[t] = src.pose[t];
    if (t < 12) sh.newS[t] = src.S[t];
    if (t == 0) { sh.weighted = src.weighted; sh.level_done = src.level_done; }
    __syncthreads();
  }
  int nl = -1, nit = 0;
  if (ADAPT && lvl >= 0) {
    const int it = it_in + (pending ? 1 : 0);
    const bool over = pending && (sh.level_done == lvl || it >= fa.max_it[lvl]);
    nl = over ? lvl - 1 : lvl;
    nit = over ? 0 : it;
  }
  const bool ended = nl < 0;
  if (FAST && ended) {   // tolerance mode carries exp(pose) through the schedule; the twist is its log, taken once here
    if (t == 0) {
      float S[12], np[6];
      for (int i = 0; i < 12; i++) S[i] = sh.newS[i];
      log_se3_f32(S, np);
      for (int i = 0; i < 6; i++) sh.newpose[i] = np[i];
    }
    __syncthreads();
  }
  if (t < 6) dst->pose[t] = sh.newpose[t];
  if (t < 12) dst->S[t] = sh.newS[t];
  if (t < ELLC_MAX_LEVELS) dst->iters[t] = it_copy[t] + ((pending && t == lvl) ? 1 : 0);
  if (t == 0) {
    dst->weighted = sh.weighted;
    dst->level_done = sh.level_done;
    dst->pending = 0;
    dst->cur_level = (ADAPT && lvl < 0) ? lvl : nl;   // -2 stays -2: the weights of this alignment were added by an earlier graph
    dst->it_in_level = nit;
  }
  if (fa.res) {
    // the record lives in pinned host memory and a host thread may be polling its pad word (resolve_batch): the fields first,
    // made visible system-wide, then the word that says they are there
    AlignResult* r = fa.res + b;
    if (ended) {
      if (t < 6) r->pose[t] = sh.newpose[t];
      if (t < ELLC_MAX_LEVELS) r->iters[t] = it_copy[t] + ((pending && t == lvl) ? 1 : 0);
      if (t == 0) r->weighted = sh.weighted;
      if (fa.host_polls) __threadfence_system();
    }
    __syncthreads();
    if (t == 0) {
      // (only where the host may poll: a system-scope release writes the L2 back, which the other groups of a pipelined context
      // would pay for — measured 3.5 % of the pipeline's rate with the fence in every finish kernel)
      if (fa.host_polls) __threadfence_system();
      *(volatile int*)&r->pad = ended ? 0 : 1;
    }
  }
  if (fa.track_mats && b == 0 && t < 64) {   // (sh.newpose is final: every path above ends in a block barrier before the stores)
    float p[6];
#pragma unroll
    for (int i = 0; i < 6; i++) p[i] = sh.newpose[i];
    track_setup_wave(p, fa.track_K, fa.track_mats);
    if (t == 0) *fa.track_gate = ended ? 1 : 0;   // closed: the schedule needs a continuation only the host can start
  }
}
template <bool FAST, bool ADAPT = false>
__global__ __launch_bounds__(ELLC_SOLVE_THREADS) void gn_fused_finish(FusedArgs fa) {
  const int b = blockIdx.x;
  __shared__ SolveShared sh;
  fused_finish_body<FAST, ADAPT>(fa, b, fa.g.state[(size_t)(fa.seq & 1) * fa.stride_state + b], fa.g.state + b, sh);
}

// PixelWisePyramid's display planes of one pass (PixelWisePyramid.cpp:209-225 and :275-284): for every pixel of the level where
// the keyframe has a depth — display_templateimg = the current image, display_2bewarpedimg = the keyframe image,
// display_origres = their difference before warping (uchar - uchar, as an int), display_warpedimg = the current image
// interpolated at the warped point (0 when it falls outside) — and 0 where the pixel is masked. One thread per pixel of the
// plane, the reference's expressions (GUI inputs in the reference: DisplayWarpedImgPxelWise, ImageFunc.cpp:277).
__global__ void gn_display_planes(GnArgs a, uint8_t* __restrict__ templateimg, uint8_t* __restrict__ tobewarpedimg, float* __restrict__ warpedimg,
                                  float* __restrict__ origres) {
  const LevelGeom g = a.geom[a.level];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= g.cols || y >= g.rows) return;
  const KfLevelDev& K = a.kf_tab[a.level * a.max_kf + a.kf_slot[0]];
  const FrLevelDev& F = a.fr_tab[a.level * a.max_fr + a.fr_slot[0]];
  const int p = y * g.cols + x;
  const float Z = K.depth[p];
  uint8_t t = 0, k = 0;
  float w = 0.0f, o = 0.0f;
  if (Z > 0.0f) {   // mask = depth_pyramid > 0 (Frame.cpp:295-301)
    float S[12];
    for (int i = 0; i < 12; i++) S[i] = a.state[0].S[i];
    t = F.img[y * g.sw + x];
    k = K.img[y * g.sw + x];
    o = (float)((int)t - (int)k);
    const Warp wp = warp_pixel<false>(x, y, Z, g, S);
    const Taps tp = tap_point<false>(as_global(F.img), g.sw, g.cols, g.rows, wp.wx, wp.wy);
    w = (tp.I == -1.0f) ? 0.0f : tp.I;
  }
  templateimg[p] = t;
  tobewarpedimg[p] = k;
  warpedimg[p] = w;
  origres[p] = o;
}

// result export for schedules that do not end in gn_fused_finish
__global__ void gn_export_results(const AlignState* state, AlignResult* res, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const AlignState& st = state[b];
  AlignResult& r = res[b];
  for (int i = 0; i < 6; i++) r.pose[i] = st.pose[i];
  r.weighted = st.weighted;
  r.pad = 0;
  for (int l = 0; l < ELLC_MAX_LEVELS; l++) r.iters[l] = st.iters[l];
}

// initial state of alignment b from the caller's initial relative pose
__device__ inline void init_state_record(AlignState& st, const float* init_pose, int b, int top_level);

// first kernel of a schedule: the staged batch description (slots, unique slots, initial poses: 9 * max_batch words) is
// copied from pinned host memory by the first copy_blocks blocks; the remaining blocks initialise the alignment states
// straight from the staged initial poses (same launch: one dependent kernel boundary less at the head of every batch)
__global__ void stage_in(int* __restrict__ dst, const int* __restrict__ src_host, int n, int copy_blocks, AlignState* state, int B, int max_batch,
                         int top_level) {
  if ((int)blockIdx.x < copy_blocks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src_host[i];
    return;
  }
  const int b = ((int)blockIdx.x - copy_blocks) * blockDim.x + threadIdx.x;
  if (b < B) init_state_record(state[b], (const float*)(src_host + 3 * max_batch), b, top_level);
}

// The same for a schedule that is launched kernel by kernel (the tracking call, one or two alignments): the staged record
// arrives in the kernel arguments instead of being read from pinned host memory (a PCIe round trip at the head of the chain)
// (StageSmall: defined in front of gn_fca_persist, which can take the staging along)
// count_*: ellc_track_frame's seeds figure rides along — blocks 1.. count the depth map's valid hypotheses (dm_count_valid_body,
// ellc_kernels_depth.hpp) while block 0 stages; count_valid == nullptr: a launch of one block
__global__ __launch_bounds__(1024) void stage_in_args(int* __restrict__ dst, StageSmall s, int B, int n_unique, int cap, AlignState* state, int top_level,
                                                      const uint8_t* count_valid, int count_n, int* count_acc, int* count_host) {
  if (blockIdx.x > 0) {
    dm_count_valid_body(count_valid, count_n, count_acc, count_host, (int)blockIdx.x - 1, (int)gridDim.x - 1);
    return;
  }
  const int t = threadIdx.x;
  if (t >= 64) return;
  if (t < B) { dst[t] = s.kf[t]; dst[cap + t] = s.fr[t]; }
  if (t < n_unique) dst[2 * cap + t] = s.uniq[t];
  if (t < 6 * B) ((float*)(dst + 3 * cap))[t] = s.pose[t];
  // init_state_record with exp(pose) spread over nine lanes per alignment (lane 3 r + k of a group of 16 evaluates entry (r, k):
  // the operations exp_se3 performs for it, see exp_se3_entry), instead of ~5 us of dependent f64 arithmetic on one lane
  const int b = t >> 4, l = t & 15;
  const int bb = min(b, 1), l9 = min(l, 8), r3 = l9 / 3, k3 = l9 - 3 * r3;
  const float* p = s.pose + bb * 6;
  double Rrk, Vv;
  exp_se3_entry((double)p[0], (double)p[1], (double)p[2], (double)p[3], (double)p[4], (double)p[5], r3, k3, Rrk, Vv);
  const double trow = (Vv + __shfl_down(Vv, 1)) + __shfl_down(Vv, 2);   // t[r] in lanes 0, 3, 6 of the group
  if (b < B) {
    AlignState& st = state[b];
    if (l < 9) {
      st.S[r3 * 4 + k3] = (float)Rrk;
      if (k3 == 0) st.S[r3 * 4 + 3] = (float)trow;
    }
    if (l < 6) { st.pose[l] = p[l]; st.delta[l] = 0.0f; st.b[l] = 0.0f; }
    if (l < ELLC_MAX_LEVELS) st.iters[l] = 0;
    if (l == 9) { st.weighted = 0.0f; st.level_done = -1; st.pending = 0; st.cur_level = top_level; st.it_in_level = 0; }
    for (int i = l; i < 36; i += 16) { st.H[i] = 0.0f; st.Hinv[i] = 0.0f; }
  }
}

#ifdef ELLC_DIAG_ABI   // (the measurement hooks' staging: ellc_profile_gn_kernel)
__global__ void gn_init_state(AlignState* state, const float* init_pose, int B, int top_level) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  init_state_record(state[b], init_pose, b, top_level);
}
#endif

__device__ inline void init_state_record(AlignState& st, const float* init_pose, int b, int top_level) {
  float p[6];
  for (int i = 0; i < 6; i++) { p[i] = init_pose[b * 6 + i]; st.pose[i] = p[i]; st.delta[i] = 0.0f; }
  float S[12];
  exp_se3_f32(p, S);
  for (int i = 0; i < 12; i++) st.S[i] = S[i];
  st.weighted = 0.0f;
  st.level_done = -1;
  st.pending = 0;
  st.cur_level = top_level;
  st.it_in_level = 0;
  for (int l = 0; l < ELLC_MAX_LEVELS; l++) st.iters[l] = 0;
  for (int i = 0; i < 36; i++) { st.H[i] = 0.0f; st.Hinv[i] = 0.0f; }
  for (int i = 0; i < 6; i++) st.b[i] = 0.0f;
}

// PixelWisePyramid::saveWeights(true) (:544-549): weight_pyramid[l] += display_weightimg (masked pixels add 0)
__global__ void gn_add_saved_weights(const KfLevelDev* kf_tab, const int* kf_slot, const LevelGeom* geom, int level, int max_kf, int fast_records) {
  const int b = blockIdx.y;
  const KfLevelDev& K = kf_tab[level * max_kf + kf_slot[b]];
  const int V = *K.count;
  const int cols = geom[level].cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
    // saved weights exist in the FCA schedule only: its records carry the pixel position
    size_t p;
    if (fast_records) {
      int x, y;
      fcaf_position_at(K, geom[level], i, x, y);
      p = (size_t)y * cols + x;
    } else {
      const uint32_t xyI = K.crec[i].xyI;
      p = (size_t)((xyI >> 12) & 0xfffu) * cols + (xyI & 0xfffu);
    }
    K.weight[p] = K.weight[p] + K.wlast[i];
  }
}

// The same for every level in ONE launch at the end of a fused schedule (grid (x, B, L)): each level's wlast holds the
// weights of that level's last executed pixel pass. Only once the alignment's schedule has ended (the state-driven schedule
// may stop short of it and be continued: cur_level of the record the finish kernel wrote is -1 at the end) and only ONCE per
// alignment: a continuation graph carries the alignments its first graph already ended as cur_level = -2 (gn_fca_adaptive).
__device__ __forceinline__ void saved_weights_all_body(const KfLevelDev* kf_tab, const int* kf_slot, const LevelGeom* geom, const AlignState* state, int max_kf,
                                                       int fast_records, int b, int level, int bx, int nbx, int tid, int nthreads) {
  if (state[b].cur_level != -1) return;
  const KfLevelDev& K = kf_tab[level * max_kf + kf_slot[b]];
  const int V = *K.count;
  const int cols = geom[level].cols;
  for (int i = bx * nthreads + tid; i < V; i += nbx * nthreads) {
    size_t p;
    if (fast_records) {
      int x, y;
      fcaf_position_at(K, geom[level], i, x, y);
      p = (size_t)y * cols + x;
    } else {
      const uint32_t xyI = K.crec[i].xyI;
      p = (size_t)((xyI >> 12) & 0xfffu) * cols + (xyI & 0xfffu);
    }
    K.weight[p] = K.weight[p] + K.wlast[i];
  }
}
__global__ void gn_add_saved_weights_all(const KfLevelDev* kf_tab, const int* kf_slot, const LevelGeom* geom, const AlignState* state, int max_kf,
                                         int fast_records) {
  saved_weights_all_body(kf_tab, kf_slot, geom, state, max_kf, fast_records, (int)blockIdx.y, (int)blockIdx.z, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x,
                         (int)blockDim.x);
}
// ellc_track_frame (r06): the tracking call's saved weights ride in its observation's selection launch (dm_observe_select<true>) —
// further blocks of that launch, which is the next one behind the resident launch either way — instead of a launch of their own
// between the alignment and the depth stages (5 us and a kernel boundary of every tracked frame).
struct RideWeights {
  int n = 0, per_level = 0;   // n further blocks (0: none), per_level of them for each pyramid level
  const KfLevelDev* kf_tab = nullptr;
  const int* kf_slot = nullptr;
  const LevelGeom* geom = nullptr;
  const AlignState* state = nullptr;
  int max_kf = 0, fast_records = 0;
};

}  // namespace ellc
