// Compaction of a keyframe's valid pixels into the per-level lists the Gauss-Newton kernels iterate (reference: the
// mask / count of frame::calculateNonZeroDepthPts + updationOnPyrChange, Frame.cpp:295-327), plus everything about a
// pixel that does not depend on the pose: the FCA records (FcaRec) and, for the constant-weight path, the ICA records
// with the template-gradient Jacobian and the per-tile sums of H (PixelWisePyramid.cpp:561-680, :938).
#pragma once
#include "ellc_kernels_image.hpp"
#include "ellc_kernels_gn.hpp"

namespace ellc {

// ---------------------------------------------------------------------------------------------------
// Compaction of the keyframe's valid pixels (mask = depth_pyramid[l] > 0, Frame.cpp:295-301) into the wave-owned regions of the
// level's layout (LevelLayout, ellc_device.hpp) as launches of its own: a count launch and a scatter launch, a WAVE per tile in both,
// every tile of every level of every listed keyframe at once — the form for schedules that cannot build their lists themselves
// (launches over fewer than eight alignments, where a wave's handful of tiles walked one after the other costs more than two
// launches: a single 640x480 alignment 0.223 ms against 0.234; alignments that share a keyframe slot, whose blocks would write the
// same regions side by side; the state-driven tracking schedule; the single-step API) and for the modes whose pixel pass does not
// hide a build (exact arithmetic, constant weights: measured). The tolerance-mode FCA schedule of a batch builds its
// lists in the first launch of every level instead (fcaf_build_pass, ellc_kernels_gn.hpp) and launches neither kernel.
// Tiles are addressed by their position p in the layout's tile table: tiles[p] is the tile, owner[p] the wave region it belongs
// to, and the tiles of one region are consecutive in p — the offset of a tile inside its region is the sum of the counts of the
// region's earlier tiles (at most a few dozen words, read by the scatter wave itself: no prefix pass).
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
  int blk_prefix[ELLC_MAX_LEVELS + 1];   // prefix of the blocks (four tiles each) per level: blockIdx.x -> level, first tile position
  int level0;                  // ica_hinv: levels level0 + blockIdx.x
};

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

// blockIdx.x -> level and the tile position p of this wave (p >= ntiles: nothing to do)
__device__ __forceinline__ int prep_locate(const PrepArgs& a, int& p) {
  int level = 0;
  while (level + 1 < a.levels && (int)blockIdx.x >= a.blk_prefix[level + 1]) level++;
  p = ((int)blockIdx.x - a.blk_prefix[level]) * (ELLC_GN_THREADS / 64) + wave_index();
  return level;
}

__global__ __launch_bounds__(256) void prep_count(PrepArgs a) {
  int p;
  const int level = prep_locate(a, p);
  const LevelLayout Lay = a.lay[level];
  if (p >= Lay.ntiles) return;
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const int n = a.geom[level].n, ppt = Lay.ppt;
  const unsigned pix0 = (unsigned)as_const(Lay.tiles)[p] * (unsigned)(ppt << 6) + (threadIdx.x & 63u);
  const ELLC_GLOBAL float* depth = gptr(K.depth);
  float d[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const unsigned i = pix0 + (unsigned)(j * 64);
    d[j] = (j < ppt && i < (unsigned)n) ? depth[i] : 0.0f;
  }
  int tot = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) tot += __popcll(__ballot(d[j] > 0.0f));
  if ((threadIdx.x & 63) == 0) K.tile_count[p] = tot;
}

// Scatter: the wave parks the valid pixels of its tile in its LDS ring by ballot rank (tile_park), then runs densely over the
// parked entries — every lane has a valid pixel — computes the record (three IEEE divisions in the exact forms) and stores it;
// consecutive lanes write consecutive records. Without the LDS step the divisions would run for every wave that holds at least one
// valid pixel, i.e. about four times as often on a semi-dense map. No block barrier but the one in front of the sums of H.
template <int NEED>   // compile-time copy of PrepArgs::need: the FCA variant carries no Jacobian / H-sum code (and registers)
__global__ __launch_bounds__(256) void prep_scatter(PrepArgs a) {
  int p;
  const int level = prep_locate(a, p);
  const LevelLayout Lay = a.lay[level];
  const KfLevelDev K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const LevelGeom& g = a.geom[level];
  const int n = g.n;
  const int lane = threadIdx.x & 63;
  const bool on = p < Lay.ntiles;   // wave-uniform (a block's last waves may have no tile; they still take part in the H sums)
  const int ppt = Lay.ppt, T = ppt << 6;
  __shared__ BuildShared bsh;
  uint2* ring = bsh.ring[wave_index()];
  float* vring = bsh.vring[wave_index()];
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
  if (on) {
  // the second plane the records need rides with the depths: the variance (FCA) or the saved weights (constant-weight records)
  const ELLC_GLOBAL float* plane2 = (NEED == 8 || NEED == 2) ? var : wgt;
  const unsigned pix0 = (unsigned)as_const(Lay.tiles)[p] * (unsigned)T + (unsigned)lane;
  TileRegs tr;
  tile_load(gptr(K.depth), plane2, img, n, g.cols, g.sw, inv_cols, ppt, pix0, tr);
  // this tile's offset in its region = the counts prep_count left for the region's earlier tiles
  const int vb = as_const(Lay.owner)[p];
  const int tb = as_const(Lay.blk_begin)[vb], te = as_const(Lay.blk_begin)[vb + 1];
  int running = 0;
  for (int j0 = tb; j0 < p; j0 += 64) {   // wave-uniform
    int part = (j0 + lane < p) ? gptr(K.tile_count)[j0 + lane] : 0, tot;
    wave_inclusive_scan(part, tot);
    running += tot;
  }
  running = __builtin_amdgcn_readfirstlane(running);
  const unsigned region = (unsigned)tb * (unsigned)T;
  const int nvalid = tile_park(tr, ppt, pix0, 0, ring, vring);
  if (lane == 0 && p == te - 1) K.blk_count[vb] = running + nvalid;   // the region's last tile: its total
  const unsigned tile_off = region + (unsigned)running;
  if constexpr (NEED == 8 || NEED == 2) {
    // FCA records: everything a record needs was parked with the pixel
    for (int r0 = 0; r0 < nvalid; r0 += 64) {   // wave-uniform trip count
      const int r = r0 + lane;
      if (r < nvalid) {
        const uint2 e = ring[r];
        const float Z = __builtin_bit_cast(float, e.y), vv = vring[r];
        const uint32_t Ib = e.x >> 24;
        int x, y;
        pix_xy((int)(e.x & 0xffffffu), cols, inv_cols, x, y);
        const unsigned pos = tile_off + (unsigned)r;
        if constexpr (NEED == 8) {   // tolerance mode: one 16-byte record per pixel (FcaRecF)
          const uint32_t yI = __builtin_bit_cast(uint32_t, (float)y) | Ib;   // FcaRecF: y < 4096 as f32 has its 12 low bits clear
          const float dd = __builtin_amdgcn_rcpf(Z);
          const float pn = ((float)x - g.cx) * g.rfx;   // u / fx (fcaf_pixel forms v / fy from y)
          crec[pos] = (u32x4){yI, __builtin_bit_cast(uint32_t, pn), __builtin_bit_cast(uint32_t, vv), __builtin_bit_cast(uint32_t, dd)};
        } else {   // one 20-byte record per pixel (FcaRec): a 16-byte word and a 4-byte word
          const uint32_t xyI = (uint32_t)x | ((uint32_t)y << 12) | (Ib << 24);
          const double invZ = 1.0 / (double)Z;
          const unsigned long long zb = __builtin_bit_cast(unsigned long long, invZ);
          ELLC_GLOBAL char* rp = (ELLC_GLOBAL char*)K.crec + pos * (unsigned)sizeof(FcaRec);
          typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
          *(ELLC_GLOBAL u32x4a*)rp = (u32x4a){xyI, __builtin_bit_cast(uint32_t, Z), __builtin_bit_cast(uint32_t, vv), (uint32_t)zb};
          *(ELLC_GLOBAL uint32_t*)(rp + 16) = (uint32_t)(zb >> 32);
        }
      }
    }
  } else
  for (int r0 = 0; r0 < nvalid; r0 += 64) {   // wave-uniform trip count
    // the record at position p of the wave's region belongs to lane p mod 64, as in the Gauss-Newton launches that walk the region
    const int r = r0 + ((lane - running) & 63);
    if (r < nvalid) {
    const uint2 e = ring[r];
    const int i = (int)(e.x & 0xffffffu);
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
    const float Ikf = (need & 4) ? (float)pc : (float)(e.x >> 24);
    if (need & 1) {   // unfused ICA kernels read planes
      cxy[pos] = xy;
      cZ[pos] = Z;
      cI[pos] = Ikf;
      cW[pos] = vring[r];
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
      const float wsave = vring[r];
      const float X = (((float)x - cx) * Z) / fx;
      const float Y = (((float)y - cy) * Z) / fy;
      if (need & 16) {   // tolerance mode: one 16-byte word (ica_load_fast); twice a central difference of bytes is an integer below 2^15
        const uint32_t xyI = (uint32_t)x | ((uint32_t)y << 12) | (pc << 24);
        const uint32_t gxy = ((uint32_t)(int)(2.0f * gradx) & 0xffffu) | ((uint32_t)(int)(2.0f * grady) << 16);
        crec[pos] = (u32x4){xyI, __builtin_bit_cast(uint32_t, __builtin_amdgcn_rcpf(Z)), __builtin_bit_cast(uint32_t, wsave), gxy};
      } else {   // IcaRec: three 16-byte words
        ELLC_GLOBAL u32x4* ro = (ELLC_GLOBAL u32x4*)((ELLC_GLOBAL char*)K.irec + pos * (unsigned)sizeof(IcaRec));
        ro[0] = (u32x4){__builtin_bit_cast(uint32_t, X), __builtin_bit_cast(uint32_t, Y), __builtin_bit_cast(uint32_t, Z), __builtin_bit_cast(uint32_t, Ikf)};
        ro[1] = (u32x4){__builtin_bit_cast(uint32_t, wsave), __builtin_bit_cast(uint32_t, J[0]), __builtin_bit_cast(uint32_t, J[1]), __builtin_bit_cast(uint32_t, J[2])};
        ro[2] = (u32x4){__builtin_bit_cast(uint32_t, J[3]), __builtin_bit_cast(uint32_t, J[4]), __builtin_bit_cast(uint32_t, J[5]), 0u};
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
  }
  }
  if (need & 4) block_reduce_store<27>(hacc, K.hpart + (size_t)((int)blockIdx.x - a.blk_prefix[level]) * ELLC_PART_STRIDE);   // block-uniform condition
}

// ICA: H of one (keyframe slot, level) from the scatter blocks' sums (fixed-order f64 combine), then cv::Mat::inv(DECOMP_LU)
// (PixelWisePyramid.cpp:938-939). One block per (level, unique slot); the level's inverse is kept with the slot.
__global__ __launch_bounds__(ELLC_SOLVE_THREADS) void ica_hinv(PrepArgs a) {
  const int level = a.level0 + (int)blockIdx.x;
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const int T = (a.lay[level].ntiles + ELLC_GN_THREADS / 64 - 1) / (ELLC_GN_THREADS / 64);   // the blocks of prep_scatter
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
