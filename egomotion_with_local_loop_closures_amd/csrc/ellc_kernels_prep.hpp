// Compaction of a keyframe's valid pixels into the per-level lists the Gauss-Newton kernels iterate (reference: the
// mask / count of frame::calculateNonZeroDepthPts + updationOnPyrChange, Frame.cpp:295-327), plus everything about a
// pixel that does not depend on the pose: the FCA records (FcaRec) and, for the constant-weight path, the ICA records
// with the template-gradient Jacobian and the per-tile sums of H (PixelWisePyramid.cpp:561-680, :938).
#pragma once
#include "ellc_kernels_image.hpp"
#include "ellc_kernels_gn.hpp"

namespace ellc {

// ---------------------------------------------------------------------------------------------------
// Compaction of the keyframe's valid pixels (mask = depth_pyramid[l] > 0, Frame.cpp:295-301) into the block-owned regions of
// the level's layout (LevelLayout, ellc_device.hpp). r05: ONE launch, no count pass, no prefix over tiles — the block that builds a
// region is the only one that needs its count (r01-r04: a count launch per tile of 2048 pixels, then a scatter launch whose blocks
// summed the counts of the tiles before theirs; 18 % of a launch group's kernel time and the depth planes read twice). The
// production schedules do not even launch this kernel: the first Gauss-Newton launch of a level builds the regions it then walks
// (fca_build_pass / ica_build_pass, ellc_kernels_gn.hpp). It remains for the single-step API, for batches in which alignments share
// a keyframe slot (their blocks would write the same regions side by side), and for the state-driven tracking schedule.

struct PrepArgs {
  const LevelGeom* geom;
  const KfLevelDev* kf_tab;
  const LevelLayout* lay;      // [levels] the layout the consumers of these lists use
  const int* slots;            // unique keyframe slots
  int levels, max_kf;
  int need;                    // bit 0: planes Z / I / saved weight (unfused ICA kernels); bit 1: FcaRec records (FCA);
                               // bit 2: IcaRec records + per-block sums of H (fused ICA schedule); bit 3: FcaRecF records
                               // (FCA in tolerance mode, cfg.arith = ELLC_ARITH_FAST); bit 4 (with bit 2): the ICA records in
                               // the tolerance mode's 16-byte form (IcaInF) instead of IcaRec
  int blk_prefix[ELLC_MAX_LEVELS + 1];   // prefix of the blocks per level (prep_build: blockIdx.x -> level, block)
  int level0;                  // ica_hinv: levels level0 + blockIdx.x
};

// Second half, dense over the parked entries — every lane has a valid pixel — computes the record (three IEEE divisions in the
// exact forms) and stores it; consecutive lanes write consecutive records. Without the LDS step the divisions would run for every
// wave that holds at least one valid pixel, i.e. about four times as often on a semi-dense map.
template <int NEED>   // compile-time copy of PrepArgs::need: the FCA variant carries no Jacobian / H-sum code (and registers)
__global__ __launch_bounds__(256) void prep_build(PrepArgs a) {
  int level = 0;
  while (level + 1 < a.levels && (int)blockIdx.x >= a.blk_prefix[level + 1]) level++;
  const int sub = (int)blockIdx.x - a.blk_prefix[level];
  const KfLevelDev K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const LevelLayout Lay = a.lay[level];
  const LevelGeom& g = a.geom[level];
  const int n = g.n;
  const int tb = Lay.blk_begin[sub], te = Lay.blk_begin[sub + 1];
  const int ppt = Lay.ppt, T = ppt << 8;
  const unsigned region = (unsigned)tb * (unsigned)T;
  constexpr int QCAP = ELLC_TILE_MAX;   // (a tile's entries are consumed before the next tile is parked: the ring never wraps here)
  __shared__ int cnt[33];
  __shared__ uint2 ring[QCAP];
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
  int running = 0;   // records of this block's region so far
  float d[8];
  if (tb < te) tile_load(gptr(K.depth), n, ppt, (unsigned)Lay.tiles[tb] * (unsigned)T + threadIdx.x, d);
  for (int jt = tb; jt < te; jt++) {   // block-uniform
    const unsigned pix0 = (unsigned)Lay.tiles[jt] * (unsigned)T + threadIdx.x;
    const int nvalid = tile_park<QCAP>(d, ppt, pix0, 0, cnt, ring);
    // the next tile's depths are requested before this tile's records are formed (a block's life is a chain of memory round trips)
    if (jt + 1 < te) tile_load(gptr(K.depth), n, ppt, (unsigned)Lay.tiles[jt + 1] * (unsigned)T + threadIdx.x, d);
    const unsigned tile_off = region + (unsigned)running;
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
        const uint2 e = ring[act[k] ? r : 0];   // (an idle lane reads entry 0: a valid address, nothing is stored)
        ii[k] = (int)e.x;
        ZZ[k] = __builtin_bit_cast(float, e.y);
        pix_xy(ii[k], cols, inv_cols, xx[k], yy[k]);
        Ib[k] = img[(unsigned)(yy[k] * sw + xx[k])];
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
    const uint2 e = ring[r];
    const int i = (int)e.x;
    const float Z = __builtin_bit_cast(float, e.y);
    const unsigned pos = tile_off + (unsigned)r;
    int x, y;
    pix_xy(i, cols, inv_cols, x, y);
    const uint32_t xy = ((uint32_t)y << 16) | (uint32_t)x;
    // the pixel and its two row neighbours (clamped at the image's edges, Frame.cpp:185-285) from ONE unaligned dword starting at
    // max(x - 1, 0) — three byte gathers cost the vector cache three times what the dword costs (the row's stored width and the
    // slack behind the last level's image cover the read past x + 1)
    typedef uint32_t u32a1 __attribute__((aligned(1)));
    const uint32_t rowq = (need & 4) ? *(const ELLC_GLOBAL u32a1*)(img + (unsigned)(y * sw + max(x - 1, 0))) : 0u;
    const uint32_t b0 = rowq & 0xffu, b1 = (rowq >> 8) & 0xffu, b2 = (rowq >> 16) & 0xffu;
    const uint32_t pc = (x == 0) ? b0 : b1;                                  // I(x, y)
    const uint32_t pxm = b0;                                                 // I(max(x - 1, 0), y)
    const uint32_t pxp = (x == 0) ? b1 : ((x == cols - 1) ? b1 : b2);        // I(min(x + 1, cols - 1), y)
    const float Ikf = (need & 4) ? (float)pc : (float)img[(unsigned)(y * sw + x)];
    if (need & 1) {   // unfused ICA kernels read planes
      cxy[pos] = xy;
      cZ[pos] = Z;
      cI[pos] = Ikf;
      cW[pos] = wgt[(unsigned)i];
    }
    if (need & 4) {   // ICA record: template-gradient Jacobian at the integer pixel (PixelWisePyramid.cpp:561-680)
      // frame::calculateGradient of the keyframe level image at (y,x)  (Frame.cpp:185-285)
      const int ym = max(y - 1, 0), yp = min(y + 1, g.rows - 1);
      const float sx = (x == 0 || x == cols - 1) ? 1.0f : 0.5f;
      const float sy = (y == 0 || y == g.rows - 1) ? 1.0f : 0.5f;
      const float gradx = sx * ((float)pxp - (float)pxm);
      const float grady = sy * ((float)img[(unsigned)(yp * sw + x)] - (float)img[(unsigned)(ym * sw + x)]);
      float J[6];
      jacobian_row<false>(gradx, grady, x, y, 1.0 / (double)Z, g, J);
      const float wsave = wgt[(unsigned)i];
      const float X = (((float)x - cx) * Z) / fx;
      const float Y = (((float)y - cy) * Z) / fy;
      if (need & 16) {   // tolerance mode: one 16-byte word (ica_load_fast); twice a central difference of bytes is an integer below 2^15
        const uint32_t xyI = (uint32_t)x | ((uint32_t)y << 12) | (pc << 24);
        const uint32_t gxy = ((uint32_t)(int)(2.0f * gradx) & 0xffffu) | ((uint32_t)(int)(2.0f * grady) << 16);
        crec[pos] = (u32x4){xyI, __builtin_bit_cast(uint32_t, __builtin_amdgcn_rcpf(Z)), __builtin_bit_cast(uint32_t, wsave), gxy};
      } else {
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
    if ((NEED & 4) && !(NEED & 16)) {
      __syncthreads();
      const int chunks = min(256, nvalid - r0) * CH;
      const unsigned obase = (tile_off + (unsigned)r0) * CH;
      for (int cidx = (int)threadIdx.x; cidx < chunks; cidx += 256) rec_out[obase + (unsigned)cidx] = s_rec[cidx];
      __syncthreads();
    }
  }
    running += nvalid;
    __syncthreads();   // the ring and the counts are reused by the next tile
  }
  if (threadIdx.x == 0) K.blk_count[sub] = running;
  if (need & 4) block_reduce_store<27>(hacc, K.hpart + (size_t)sub * ELLC_PART_STRIDE);   // block-uniform condition
}

// ICA: H of one (keyframe slot, level) from the per-block sums (fixed-order f64 combine), then cv::Mat::inv(DECOMP_LU)
// (PixelWisePyramid.cpp:938-939). One block per (level, unique slot); the level's inverse is kept with the slot.
__global__ __launch_bounds__(ELLC_SOLVE_THREADS) void ica_hinv(PrepArgs a) {
  const int level = a.level0 + (int)blockIdx.x;
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const int T = a.lay[level].nblk;
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
