// Compaction of a keyframe's valid pixels into the per-level lists the Gauss-Newton kernels iterate (reference: the
// mask / count of frame::calculateNonZeroDepthPts + updationOnPyrChange, Frame.cpp:295-327), plus everything about a
// pixel that does not depend on the pose: the FCA records (FcaRec) and, for the constant-weight path, the ICA records
// with the template-gradient Jacobian and the per-tile sums of H (PixelWisePyramid.cpp:561-680, :938).
#pragma once
#include "ellc_kernels_image.hpp"
#include "ellc_kernels_gn.hpp"

namespace ellc {

// ---------------------------------------------------------------------------------------------------
// Compaction of the keyframe's valid pixels (mask = depth_pyramid[l] > 0, Frame.cpp:295-301), raster
// order preserved. Two launches (count per tile, then scatter) cover all levels of all listed keyframe slots.
#define ELLC_TILE 2048          // pixels per block: 256 threads x 8 consecutive pixels (two float4 loads)

struct PrepArgs {
  const LevelGeom* geom;
  const KfLevelDev* kf_tab;
  const int* slots;            // unique keyframe slots (null: slot_inline — a launch outside an alignment, see enqueue_eager_lists)
  int levels, max_kf;
  int need;                    // bit 0: planes Z / I / saved weight (unfused ICA kernels); bit 1: FcaRec records (FCA);
                               // bit 2: IcaRec records + per-tile sums of H (fused ICA schedule); bit 3: FcaRecF records
                               // (FCA in tolerance mode, cfg.arith = ELLC_ARITH_FAST); bit 4 (with bit 2): the ICA records in
                               // the tolerance mode's 16-byte form (IcaInF) instead of IcaRec
  int tile_begin[ELLC_MAX_LEVELS + 1];   // prefix of tiles per level
  int tile0, level0;           // this launch covers tiles tile0 + blockIdx.x (count / scatter), levels level0 + blockIdx.x (scan)
  int slot_inline;             // the slot itself when `slots` is null (one keyframe; a scalar: a dynamically indexed member would send the whole argument struct through private memory)
  unsigned lb_tag;             // 0: prep_count has left the tiles' counts. Else (r05, the tracking call: one or two keyframes, launched
                               // kernel by kernel) there is NO count launch: a scatter block publishes `lb_tag << 12 | its tile's count`
                               // as soon as it has it and sums the tagged counts of the tiles of its level in front of it as they
                               // appear (agent-scope stores / loads, no fence; blocks are dispatched in tile order, so the tiles a block
                               // waits for are resident or done). For launch groups the same was measured 17 % SLOWER (NOTEBOOK 5.1):
                               // 25 000 waiting blocks; here they are 201.
};
// (the value is block-uniform; said explicitly, because behind the null test the compiler loads a.slots[k] with a VECTOR load and then
// carries the slot's table entry — every plane pointer of the kernel — in vector registers: prep_count 45.7 -> 58.3 us, prep_scatter
// 122.8 -> 134.5 us per launch group, the batch pipeline 0.1245 -> 0.1317 ms per step, found by tools/ab_trace.sh in r06)
__device__ __forceinline__ int prep_slot(const PrepArgs& a, unsigned k) { return __builtin_amdgcn_readfirstlane(a.slots ? a.slots[k] : a.slot_inline); }


__device__ __forceinline__ int prep_level_of(const PrepArgs& a, int tile, int& local) {
  int l = 0;
  while (l + 1 < a.levels && tile >= a.tile_begin[l + 1]) l++;
  local = tile - a.tile_begin[l];
  return l;
}

// eight consecutive depths of this thread (zeros past the end of the plane)
__device__ __forceinline__ void prep_load8(const float* __restrict__ depth, int i0, int n, float (&d)[8]) {
  if (i0 + 7 < n) {
    const float4 a = *reinterpret_cast<const float4*>(depth + i0);
    const float4 b = *reinterpret_cast<const float4*>(depth + i0 + 4);
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = (i0 + j < n) ? depth[i0 + j] : 0.0f;
  }
}

// inclusive scan inside a wave; returns the wave total through `total`
__device__ __forceinline__ int wave_inclusive_scan(int v, int& total) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  total = __shfl(v, 63, 64);
  return v;
}

__global__ __launch_bounds__(256) void prep_count(PrepArgs a) {
  int local;
  const int level = prep_level_of(a, a.tile0 + (int)blockIdx.x, local);
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + prep_slot(a, blockIdx.y)];
  const int n = a.geom[level].n;
  const int i0 = local * ELLC_TILE + threadIdx.x * 8;
  float d[8];
  prep_load8(K.depth, i0, n, d);
  int c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) c += (d[j] > 0.0f) ? 1 : 0;
  __shared__ int ws[4];
  int tot;
  wave_inclusive_scan(c, tot);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    K.tile_count[local] = ws[0] + ws[1] + ws[2] + ws[3];
  }
}

// Scatter, two phases per tile of ELLC_TILE pixels. Phase 1: thread t owns pixels base + j*256 + t (j = 0..7), so the
// depth loads of a wave are contiguous; ballot ranks give every valid pixel its raster-order rank inside the tile
// (order = (j, wave, lane)), and (pixel index, depth) are parked in LDS at that rank. Phase 2 runs densely over the
// parked entries — every lane has a valid pixel — computes the record (three IEEE divisions) and stores it; consecutive
// lanes write consecutive records. Without the LDS step the divisions would run for every wave that holds at least one
// valid pixel, i.e. about four times as often on a semi-dense map.
template <int NEED>   // compile-time copy of PrepArgs::need: the FCA variant carries no Jacobian / H-sum code (and registers)
__global__ __launch_bounds__(256) void prep_scatter(PrepArgs a) {
  int local;
  const int level = prep_level_of(a, a.tile0 + (int)blockIdx.x, local);
  const KfLevelDev K = a.kf_tab[level * a.max_kf + prep_slot(a, blockIdx.y)];
  const LevelGeom& g = a.geom[level];
  const int n = g.n;
  const int base = local * ELLC_TILE + (int)threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int cnt[33];   // [j][wave] exclusive offsets, [32] = tile total
  __shared__ int before[4]; // per wave: valid pixels in the tiles of this level that precede this one
  __shared__ uint32_t s_idx[ELLC_TILE];
  __shared__ float s_Z[ELLC_TILE];
  // the eight depth loads first, then the loads of the tile counts: both sets are in flight together (r03: a block's life is a
  // chain of memory round trips of 2-3 us each under load — table entry, counts, depths, gathers, stores: 12 us for 2048 pixels)
  float d[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = base + j * 256;
    d[j] = (i < n) ? gptr(K.depth)[(unsigned)i] : 0.0f;
  }
  if (!a.lb_tag) {   // this tile's offset in the level's list = sum of the counts prep_count left for the tiles before it (at most a few hundred)
    int part = 0, tot;
    for (int i = (int)threadIdx.x; i < local; i += 256) part += gptr(K.tile_count)[(unsigned)i];
    wave_inclusive_scan(part, tot);
    if (lane == 0) before[wave] = tot;
  }
  unsigned long long m[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    m[j] = __ballot(d[j] > 0.0f);
    if (lane == 0) cnt[j * 4 + wave] = __popcll(m[j]);
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // exclusive scan of the 32 (j, wave) counts
    int v = (lane < 32) ? cnt[lane] : 0, tot;
    const int inc = wave_inclusive_scan(v, tot);
    if (lane < 32) cnt[lane] = inc - v;
    if (lane == 0) cnt[32] = tot;
  }
  __syncthreads();
  if (a.lb_tag) {   // block-uniform: see PrepArgs::lb_tag
    const unsigned tag = a.lb_tag << 12;
    unsigned* tc = (unsigned*)K.tile_count;
    if (threadIdx.x == 0) __hip_atomic_store(tc + local, tag | (unsigned)cnt[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int part = 0, tot;
    for (int i = (int)threadIdx.x; i < local; i += 256) {
      unsigned w = 0u, polls = 0u;
      do { w = __hip_atomic_load(tc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w & 0xfffff000u) != tag && ++polls < 4096u);
      if ((w & 0xfffff000u) == tag) {
        part += (int)(w & 0xfffu);
      } else {   // (does not happen with blocks dispatched in tile order; a count is a function of the plane: no wait can be endless)
        const int e0 = i * ELLC_TILE, e1 = min(n, e0 + ELLC_TILE);
        int cnt_i = 0;
        for (int e = e0; e < e1; e++) cnt_i += (gptr(K.depth)[(unsigned)e] > 0.0f) ? 1 : 0;
        part += cnt_i;
      }
    }
    wave_inclusive_scan(part, tot);
    if (lane == 0) before[wave] = tot;
    __syncthreads();
  }
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if (d[j] > 0.0f) {
      const int r = cnt[j * 4 + wave] + __popcll(m[j] & lt);
      s_idx[r] = (uint32_t)(base + j * 256);
      s_Z[r] = d[j];
    }
  }
  __syncthreads();
  const int nvalid = cnt[32];
  const unsigned tile_off = (unsigned)(before[0] + before[1] + before[2] + before[3]);
  if (threadIdx.x == 0 && local == a.tile_begin[level + 1] - a.tile_begin[level] - 1) *K.count = (int)tile_off + nvalid;   // last tile: the level's total
  const float inv_cols = 1.0f / (float)g.cols;
  const ELLC_GLOBAL float* var = gptr(K.var);
  const ELLC_GLOBAL float* wgt = gptr(K.weight);
  const ELLC_GLOBAL uint8_t* img = gptr(K.img);
  ELLC_GLOBAL uint32_t* cxy = gptr_rw(K.cxy);
  ELLC_GLOBAL float* cZ = gptr_rw(K.cZ);
  ELLC_GLOBAL float* cI = gptr_rw(K.cI);
  ELLC_GLOBAL float* cW = gptr_rw(K.cW);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  ELLC_GLOBAL u32x4* crec = (ELLC_GLOBAL u32x4*)K.crec;
  const int cols = g.cols, sw = g.sw;
  constexpr int need = NEED;
  const float fx = g.fx, fy = g.fy, cx = g.cx, cy = g.cy;
  float hacc[27];
#pragma unroll
  for (int q = 0; q < 27; q++) hacc[q] = 0.0f;
  // The 48-byte ICA records leave through an LDS staging block of 256 records so that consecutive lanes store consecutive
  // 16-byte words (lane-per-record, every store instruction would touch a third of each line): -5 % on the kernel. The
  // FCA records (20 bytes exact, 16 tolerance mode) are stored directly (r01 A/B: the two extra barriers per 256 records cost more than the partial-line
  // stores).
  constexpr int CH = 3;
  __shared__ u32x4 s_rec[((NEED & 4) && !(NEED & 16)) ? 256 * CH : 1];
  ELLC_GLOBAL u32x4* rec_out = (ELLC_GLOBAL u32x4*)K.irec;
  if constexpr (NEED == 8 || NEED == 2) {
    // FCA records: up to four records per thread and trip, all their gathers (the image byte and the variance of each) issued
    // before the first is used — one memory round trip per 1024 records instead of one per 256
    constexpr int U = 4;
    for (int r0 = 0; r0 < nvalid; r0 += U * 256) {   // block-uniform trip count
      int ii[U], xx[U], yy[U];
      float ZZ[U], vv[U];
      uint8_t Ib[U];
      bool act[U];
#pragma unroll
      for (int k = 0; k < U; k++) {
        const int r = r0 + k * 256 + (int)threadIdx.x;
        act[k] = r < nvalid;
        const int rr = act[k] ? r : 0;   // (an idle lane reads entry 0: a valid address, nothing is stored)
        ii[k] = (int)s_idx[rr];
        ZZ[k] = s_Z[rr];
        int y = (int)(((float)ii[k] + 0.5f) * inv_cols);   // i < 2^24: exact conversion; corrected below
        if (y * cols > ii[k]) y--;
        if ((y + 1) * cols <= ii[k]) y++;
        yy[k] = y;
        xx[k] = ii[k] - y * cols;
        Ib[k] = img[(unsigned)(y * sw + xx[k])];
        vv[k] = var[(unsigned)ii[k]];
      }
#pragma unroll
      for (int k = 0; k < U; k++) {
        if (!act[k]) continue;
        const unsigned pos = tile_off + (unsigned)(r0 + k * 256) + threadIdx.x;
        const int x = xx[k], y = yy[k];
        const float Z = ZZ[k];
        if constexpr (NEED == 8) {   // tolerance mode: one 12-byte record per pixel (FcaRecF)
          const float dd = __builtin_amdgcn_rcpf(Z);
          *(ELLC_GLOBAL Rec12*)((ELLC_GLOBAL char*)K.crec + pos * 12u) =
              (Rec12){(uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)Ib[k] << 24), __builtin_bit_cast(uint32_t, vv[k]), __builtin_bit_cast(uint32_t, dd)};
        } else {   // one 20-byte record per pixel (FcaRec): a 16-byte word and a 4-byte word
          const uint32_t xyI = (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)Ib[k] << 24);
          const double invZ = 1.0 / (double)Z;
          const unsigned long long zb = __builtin_bit_cast(unsigned long long, invZ);
          ELLC_GLOBAL char* r = (ELLC_GLOBAL char*)K.crec + pos * (unsigned)sizeof(FcaRec);
          typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
          *(ELLC_GLOBAL u32x4a*)r = (u32x4a){xyI, __builtin_bit_cast(uint32_t, Z), __builtin_bit_cast(uint32_t, vv[k]), (uint32_t)zb};
          *(ELLC_GLOBAL uint32_t*)(r + 16) = (uint32_t)(zb >> 32);
        }
      }
    }
  } else
  for (int r0 = 0; r0 < nvalid; r0 += 256) {   // block-uniform trip count
    const int r = r0 + (int)threadIdx.x;
    if (r < nvalid) {
    const int i = (int)s_idx[r];
    const float Z = s_Z[r];
    const unsigned pos = tile_off + (unsigned)r;
    int y = (int)(((float)i + 0.5f) * inv_cols);   // i < 2^24: exact conversion; corrected below
    if (y * cols > i) y--;
    if ((y + 1) * cols <= i) y++;
    const int x = i - y * cols;
    const uint32_t xy = ((uint32_t)y << 16) | (uint32_t)x;
    // the pixel and its two row neighbours (clamped at the image's edges, Frame.cpp:185-285) from ONE unaligned dword starting at
    // max(x - 1, 0) — three byte gathers cost the vector cache three times what the dword costs (the row's stored width and the
    // slack behind the last level's image cover the read past x + 1)
    typedef uint32_t u32a1 __attribute__((aligned(1)));
    const uint32_t rowq = (need & (4 | 16)) ? *(const ELLC_GLOBAL u32a1*)(img + (unsigned)(y * sw + max(x - 1, 0))) : 0u;
    const uint32_t b0 = rowq & 0xffu, b1 = (rowq >> 8) & 0xffu, b2 = (rowq >> 16) & 0xffu;
    const uint32_t pc = (x == 0) ? b0 : b1;                                  // I(x, y)
    const uint32_t pxm = b0;                                                 // I(max(x - 1, 0), y)
    const uint32_t pxp = (x == 0) ? b1 : ((x == cols - 1) ? b1 : b2);        // I(min(x + 1, cols - 1), y)
    const float Ikf = (need & (4 | 16)) ? (float)pc : (float)img[(unsigned)(y * sw + x)];
    if (need & 1) {   // unfused ICA kernels read planes
      cxy[pos] = xy;
      cZ[pos] = Z;
      cI[pos] = Ikf;
      cW[pos] = wgt[(unsigned)i];
    }
    if (need & (4 | 16)) {   // ICA record: template-gradient Jacobian at the integer pixel (PixelWisePyramid.cpp:561-680)
      // frame::calculateGradient of the keyframe level image at (y,x)  (Frame.cpp:185-285)
      const int xm = max(x - 1, 0), xp = min(x + 1, cols - 1), ym = max(y - 1, 0), yp = min(y + 1, g.rows - 1);
      const float sx = (x == 0 || x == cols - 1) ? 1.0f : 0.5f;
      const float sy = (y == 0 || y == g.rows - 1) ? 1.0f : 0.5f;
      const float gradx = sx * ((float)pxp - (float)pxm);
      (void)xm; (void)xp;
      const float grady = sy * ((float)img[(unsigned)(yp * sw + x)] - (float)img[(unsigned)(ym * sw + x)]);
      const float wsave = wgt[(unsigned)i];
      if (need & 16) {   // tolerance mode: one 16-byte word (ica_load_fast); twice a central difference of bytes is an integer below 2^15
        const uint32_t xyI = (uint32_t)x | ((uint32_t)y << 12) | (pc << 24);
        const uint32_t gxy = ((uint32_t)(int)(2.0f * gradx) & 0xffffu) | ((uint32_t)(int)(2.0f * grady) << 16);
        crec[pos] = (u32x4){xyI, __builtin_bit_cast(uint32_t, __builtin_amdgcn_rcpf(Z)), __builtin_bit_cast(uint32_t, wsave), gxy};
      }
      // NEED = 16 alone (r06): the records only — the slot's H^-1 of this level is still the one an earlier call left (it is a
      // function of the keyframe's planes alone, and every writer of those clears kf_hinv_ok): no exact template row, no f64 and
      // IEEE f32 divisions, no 21 sums per pixel
      if constexpr ((NEED & 4) != 0) {
      float J[6];
      jacobian_row<false>(gradx, grady, x, y, 1.0 / (double)Z, g, J);
      const float X = (((float)x - cx) * Z) / fx;
      const float Y = (((float)y - cy) * Z) / fy;
      if (!(need & 16)) {
        const unsigned t3 = 3u * threadIdx.x;
        s_rec[t3] = (u32x4){__builtin_bit_cast(uint32_t, X), __builtin_bit_cast(uint32_t, Y), __builtin_bit_cast(uint32_t, Z), __builtin_bit_cast(uint32_t, Ikf)};
        s_rec[t3 + 1] = (u32x4){__builtin_bit_cast(uint32_t, wsave), __builtin_bit_cast(uint32_t, J[0]), __builtin_bit_cast(uint32_t, J[1]), __builtin_bit_cast(uint32_t, J[2])};
        s_rec[t3 + 2] = (u32x4){__builtin_bit_cast(uint32_t, J[3]), __builtin_bit_cast(uint32_t, J[4]), __builtin_bit_cast(uint32_t, J[5]), 0u};
      }
      int q = 0;
#pragma unroll
      for (int rr = 0; rr < 6; rr++) {
        const float wJ = J[rr] * wsave;   // weightedSteepestDescent (:664-669); H = WSD * SD^T (:938)
#pragma unroll
        for (int cc = rr; cc < 6; cc++) { hacc[q] = __builtin_fmaf(wJ, J[cc], hacc[q]); q++; }
      }
      }
    }
    if (need & 8) {   // FCA in tolerance mode: one 12-byte record per pixel (FcaRecF)
      const float d = __builtin_amdgcn_rcpf(Z);
      *(ELLC_GLOBAL Rec12*)((ELLC_GLOBAL char*)K.crec + pos * 12u) = (Rec12){(uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)img[(unsigned)(y * sw + x)] << 24),
                                                                         __builtin_bit_cast(uint32_t, var[(unsigned)i]), __builtin_bit_cast(uint32_t, d)};
    }
    if (need & 2) {   // FCA reads one 20-byte record per pixel (FcaRec)
      const uint32_t xyI = (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)img[(unsigned)(y * sw + x)] << 24);
      const double invZ = 1.0 / (double)Z;
      const unsigned long long zb = __builtin_bit_cast(unsigned long long, invZ);
      ELLC_GLOBAL char* r = (ELLC_GLOBAL char*)K.crec + pos * (unsigned)sizeof(FcaRec);
      typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
      *(ELLC_GLOBAL u32x4a*)r = (u32x4a){xyI, __builtin_bit_cast(uint32_t, Z), __builtin_bit_cast(uint32_t, var[(unsigned)i]), (uint32_t)zb};
      *(ELLC_GLOBAL uint32_t*)(r + 16) = (uint32_t)(zb >> 32);
    }
    }
    if ((NEED & 4) && !(NEED & 16)) {
      __syncthreads();
      const int chunks = min(256, nvalid - r0) * CH;
      const unsigned obase = (tile_off + (unsigned)r0) * CH;
      for (int cidx = (int)threadIdx.x; cidx < chunks; cidx += 256) rec_out[obase + (unsigned)cidx] = s_rec[cidx];
      __syncthreads();
    }
  }
  if (need & 4) block_reduce_store<27>(hacc, K.hpart + (size_t)local * ELLC_PART_STRIDE);   // block-uniform condition
}

// ICA: H of one (keyframe slot, level) from the per-tile sums (fixed-order f64 combine), then cv::Mat::inv(DECOMP_LU)
// (PixelWisePyramid.cpp:938-939). One block per (level, unique slot); the level's inverse is kept with the slot.
__global__ __launch_bounds__(ELLC_SOLVE_THREADS) void ica_hinv(PrepArgs a) {
  const int level = a.level0 + (int)blockIdx.x;
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + prep_slot(a, blockIdx.y)];
  const int T = a.tile_begin[level + 1] - a.tile_begin[level];
  __shared__ SolveShared sh;
  const int t = threadIdx.x;
  sh.part[t >> 5][t & 31] = partial_group_sum(K.hpart, T);
  __syncthreads();
  if (t < 27) {
    double s = sh.part[0][t];
#pragma unroll
    for (int g = 1; g < ELLC_SOLVE_THREADS / 32; g++) s += sh.part[g][t];
    sh.sums[t] = s;
  }
  __syncthreads();
  if (t < 64) {
    float Hm[36];
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = r; c < 6; c++) {
        const float v = (float)sh.sums[q++];
        Hm[r * 6 + c] = v;
        Hm[c * 6 + r] = v;
      }
    float x[6];
    lu_inverse6_lanes(Hm, t < 6 ? t : 0, x);
    if (t < 6) {
#pragma unroll
      for (int i = 0; i < 6; i++) K.hinv[i * 6 + t] = x[i];
    }
  }
}

}  // namespace ellc
