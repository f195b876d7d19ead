// Compaction of a keyframe's valid pixels into the per-level lists the Gauss-Newton kernels iterate (reference: the
// mask / count of frame::calculateNonZeroDepthPts + updationOnPyrChange, Frame.cpp:295-327), plus everything about a
// pixel that does not depend on the pose: the FCA records (FcaRec) and, for the constant-weight path, the ICA records
// with the template-gradient Jacobian and the per-tile sums of H (PixelWisePyramid.cpp:561-680, :938).
#pragma once
#include "ellc_kernels_image.hpp"
#include "ellc_kernels_gn.hpp"

namespace ellc {

// ---------------------------------------------------------------------------------------------------
// Compaction of the keyframe's valid pixels (mask = depth_pyramid[l] > 0, Frame.cpp:295-301) into the block-owned regions of the
// level's layout (LevelLayout, ellc_device.hpp): the plane is cut into tiles of 256 * ppt pixels, a block of a Gauss-Newton launch
// owns whole tiles and its records lie contiguously in its region, raster order inside a tile, the block's tiles ascending. Two forms:
//  * prep_region (r05; the tolerance-mode FCA records): ONE launch, a block of 512 threads per region, the depth plane read ONCE. The
//    block walks its region in rounds of two tiles; per round it reads depth, variance and intensity with the tiles' coalesced
//    pattern, ranks the valid pixels (ballots, a 64-entry scan of the per-row counts, one barrier) and every lane stores the records of
//    its own valid pixels straight at region + running + rank. Nothing crosses a block: no count launch, no prefix over tiles.
//  * prep_count + prep_scatter<records> (all other record sets: their records cost divisions and gathers per valid pixel that only pay
//    on lanes that all hold one): a block per tile; the scatter parks the tile's valid pixels in LDS by rank and forms the records
//    densely; a tile's offset in its region is the sum of the counts of the region's earlier tiles — at most a few dozen, read by the
//    scatter block itself (r01-r04: one global list per level, hundreds of tile counts to sum).
struct PrepArgs {
  const LevelGeom* geom;
  const KfLevelDev* kf_tab;
  const LevelLayout* lay;      // [levels] the layout the consumers of these lists use
  const int* slots;            // unique keyframe slots
  int levels, max_kf;
  int need;                    // bit 0: planes Z / I / saved weight (unfused ICA kernels); bit 1: FcaRec records (FCA);
                               // bit 2: IcaRec records + per-tile sums of H (fused ICA schedule); bit 3: FcaRecF records
                               // (FCA in tolerance mode, cfg.arith = ELLC_ARITH_FAST); bit 4 (with bit 2): the ICA records in
                               // the tolerance mode's 16-byte form (IcaInF) instead of IcaRec
  int pos_prefix[ELLC_MAX_LEVELS + 1];   // prefix of the tiles per level (prep_count / prep_scatter: blockIdx.x -> level, tile)
  int ppt[ELLC_MAX_LEVELS];              // LevelLayout::ppt of every level (in the kernel arguments: the depth loads of a tile block
                                         // depend on nothing in memory)
  int blk_prefix[ELLC_MAX_LEVELS + 1];   // prefix of the regions per level (prep_region: blockIdx.x -> level, block)
  int level0;                  // ica_hinv: levels level0 + blockIdx.x
};

__device__ __forceinline__ int prep_locate(const int* prefix, int levels, int& local) {
  int l = 0;
  while (l + 1 < levels && (int)blockIdx.x >= prefix[l + 1]) l++;
  local = (int)blockIdx.x - prefix[l];
  return l;
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

// ---- single pass (tolerance-mode FCA records) ---------------------------------------------------------------------------------
#define ELLC_PREP_THREADS 512
struct PrepTile { float d[8], v[8]; uint32_t I[8]; };
// the planes of this thread's pixels of tile `tile`: pixels tile * T + j * 256 + t (j < ppt); zeros past the end of the plane
__device__ __forceinline__ void prep_tile_load(const KfLevelDev& K, const LevelGeom& g, float inv_cols, int ppt, int tile, int t, PrepTile& r) {
  const unsigned pix0 = (unsigned)tile * (unsigned)(ppt << 8) + (unsigned)t;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const unsigned i = pix0 + (unsigned)(j * 256);
    const bool on = j < ppt && i < (unsigned)g.n;
    int x = 0, y = 0;
    if (on) pix_xy((int)i, g.cols, inv_cols, x, y);
    r.d[j] = on ? gptr(K.depth)[i] : 0.0f;
    r.v[j] = on ? gptr(K.var)[i] : 0.0f;
    r.I[j] = on ? (uint32_t)gptr(K.img)[(unsigned)(y * g.sw + x)] : 0u;
  }
}
__global__ __launch_bounds__(ELLC_PREP_THREADS) void prep_region(PrepArgs a) {
  int sub;
  const int level = prep_locate(a.blk_prefix, a.levels, sub);
  const KfLevelDev K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];   // by value: uniform, lives in SGPRs
  const LevelLayout Lay = a.lay[level];
  const LevelGeom g = a.geom[level];
  const int tb = as_const(Lay.blk_begin)[sub], te = as_const(Lay.blk_begin)[sub + 1];
  const int ppt = Lay.ppt, T = ppt << 8;
  const unsigned region = (unsigned)tb * (unsigned)T;
  const float inv_cols = 1.0f / (float)g.cols;
  // two tiles per round: threads 0..255 take the round's first tile, 256..511 its second
  const int half = threadIdx.x >> 8, t = threadIdx.x & 255, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // 8 waves: 4 per tile
  __shared__ int cnt[2][64 + 1];   // [round parity][half * 32 + j * 4 + wave-of-the-tile], [64] = the round's total
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int running = 0;
  PrepTile cur;
  {
    const int jt = tb + half;
    if (jt < te) prep_tile_load(K, g, inv_cols, ppt, as_const(Lay.tiles)[jt], t, cur);
    else {
#pragma unroll
      for (int j = 0; j < 8; j++) { cur.d[j] = 0.0f; cur.v[j] = 0.0f; cur.I[j] = 0u; }
    }
  }
  int parity = 0;
  for (int j0 = tb; j0 < te; j0 += 2) {   // block-uniform
    const int jt = j0 + half;
    const unsigned pix0 = (jt < te ? (unsigned)as_const(Lay.tiles)[jt] : 0u) * (unsigned)T + (unsigned)t;
    // ranks: per (tile of the round, row j, wave) the number of valid pixels, then their exclusive scan in raster order of the round
    unsigned long long m[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      m[j] = __ballot(cur.d[j] > 0.0f);   // (rows j >= ppt and a missing second tile hold zeros)
      if (lane == 0) cnt[parity][half * 32 + j * 4 + (wave & 3)] = __popcll(m[j]);
    }
    __syncthreads();
    // the next round's planes go out before anything of this round waits on memory again
    PrepTile nxt;
    {
      const int jn = j0 + 2 + half;
      if (jn < te) prep_tile_load(K, g, inv_cols, ppt, as_const(Lay.tiles)[jn], t, nxt);
      else {
#pragma unroll
        for (int j = 0; j < 8; j++) { nxt.d[j] = 0.0f; nxt.v[j] = 0.0f; nxt.I[j] = 0u; }
      }
    }
    int excl, tot;
    {
      const int v = cnt[parity][lane];
      const int inc = wave_inclusive_scan(v, tot);   // every wave redundantly: no second barrier
      excl = inc - v;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int off = __shfl(excl, half * 32 + j * 4 + (wave & 3), 64);
      const float Z = cur.d[j];
      if (Z > 0.0f) {
        const unsigned pos = region + (unsigned)(running + off + __popcll(m[j] & lt));
        int x, y;
        pix_xy((int)(pix0 + (unsigned)(j * 256)), g.cols, inv_cols, x, y);
        *(ELLC_GLOBAL Rec12*)((ELLC_GLOBAL char*)K.crec + pos * 12u) =
            (Rec12){(uint32_t)x | ((uint32_t)y << 12) | (cur.I[j] << 24), __builtin_bit_cast(uint32_t, cur.v[j]), __builtin_bit_cast(uint32_t, __builtin_amdgcn_rcpf(Z))};
      }
    }
    running += tot;
    cur = nxt;
    parity ^= 1;   // (the next round writes the other set of counts: no barrier between a round's reads and the next round's writes)
  }
  if (threadIdx.x == 0) K.blk_count[sub] = running;
}

// ---- two passes (every other record set) ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_count(PrepArgs a) {
  int tile;
  const int level = prep_locate(a.pos_prefix, a.levels, tile);
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const int n = a.geom[level].n, ppt = a.ppt[level];
  const int p = as_const(a.lay[level].tile_pos)[tile];   // where the scatter's prefix reads this tile's count (its position in the region order)
  int c = 0;
  if (ppt == 8) {   // eight consecutive pixels per thread: two 16-byte loads
    const int i0 = tile * 2048 + (int)threadIdx.x * 8;
    float d[8];
    if (i0 + 7 < n) {
      const float4 u = *reinterpret_cast<const float4*>(K.depth + i0);
      const float4 v = *reinterpret_cast<const float4*>(K.depth + i0 + 4);
      d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = u.w; d[4] = v.x; d[5] = v.y; d[6] = v.z; d[7] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) d[j] = (i0 + j < n) ? K.depth[i0 + j] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) c += (d[j] > 0.0f) ? 1 : 0;
  } else {
    const unsigned pix0 = (unsigned)tile * (unsigned)(ppt << 8) + threadIdx.x;
    for (int j = 0; j < ppt; j++) {
      const unsigned i = pix0 + (unsigned)(j * 256);
      c += (i < (unsigned)n && gptr(K.depth)[i] > 0.0f) ? 1 : 0;
    }
  }
  __shared__ int ws[4];
  int tot;
  wave_inclusive_scan(c, tot);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) K.tile_count[p] = ws[0] + ws[1] + ws[2] + ws[3];   // by position
}

// Scatter, two phases per tile. Phase 1: thread t owns pixels base + j*256 + t (j < ppt), so the depth loads of a wave are
// contiguous; ballot ranks give every valid pixel its raster-order rank inside the tile (order = (j, wave, lane)), and (pixel index,
// depth) are parked in LDS at that rank. Phase 2 runs densely over the parked entries — every lane has a valid pixel — computes the
// record (three IEEE divisions in the exact forms) and stores it; consecutive lanes write consecutive records. Without the LDS step
// the divisions would run for every wave that holds at least one valid pixel, i.e. about four times as often on a semi-dense map.
template <int NEED>   // compile-time copy of PrepArgs::need: the FCA variant carries no Jacobian / H-sum code (and registers)
__global__ __launch_bounds__(256) void prep_scatter(PrepArgs a) {
  int tile;
  const int level = prep_locate(a.pos_prefix, a.levels, tile);
  const KfLevelDev K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const LevelGeom& g = a.geom[level];
  const int n = g.n, ppt = a.ppt[level], T = ppt << 8;
  const int base = tile * T + (int)threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int cnt[33];   // [j][wave] exclusive offsets, [32] = tile total
  __shared__ int before[4]; // per wave: valid pixels in the tiles of this block's region that precede this one
  __shared__ uint32_t s_idx[ELLC_TILE_MAX];
  __shared__ float s_Z[ELLC_TILE_MAX];
  // the depth loads first, then the loads of the tile counts: both sets are in flight together
  float d[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {   // (addresses from the block index alone: these go out before any table is read)
    const int i = base + j * 256;
    d[j] = gptr(K.depth)[(unsigned)min(i, n - 1)];
    if (!(j < ppt && i < n)) d[j] = 0.0f;
  }
  // where the tile lies in the layout — its position p in the region order, its owner, the owner's tiles [tb, te) — in ONE
  // 16-byte table entry (a scalar load), and its offset in its region = the sum of the counts prep_count left for the region's
  // earlier tiles: positions tb .. p - 1, contiguous (a few dozen at most)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 ti = ((const ELLC_CONST i32x4*)a.lay[level].tile_info)[tile];
  const int p = ti.x, tb = ti.y, te = ti.z, owner = ti.w;
  {
    int part = 0, tot;
    for (int i = tb + (int)threadIdx.x; i < p; i += 256) part += gptr(K.tile_count)[(unsigned)i];
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
  const int in_region = before[0] + before[1] + before[2] + before[3];
  const unsigned tile_off = (unsigned)(tb * T + in_region);
  if (threadIdx.x == 0 && p == te - 1) K.blk_count[owner] = in_region + nvalid;   // the region's last tile: its total
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
  // FCA records (20 bytes exact, 12 tolerance mode) are stored directly (r01 A/B: the two extra barriers per 256 records cost more than the partial-line
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
    const int i = (int)s_idx[r];
    const float Z = s_Z[r];
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
  if (need & 4) block_reduce_store<27>(hacc, K.hpart + (size_t)tile * ELLC_PART_STRIDE);   // block-uniform condition
}

// ICA: H of one (keyframe slot, level) from the per-tile sums (fixed-order f64 combine), then cv::Mat::inv(DECOMP_LU)
// (PixelWisePyramid.cpp:938-939). One block per (level, unique slot); the level's inverse is kept with the slot.
__global__ __launch_bounds__(ELLC_SOLVE_THREADS) void ica_hinv(PrepArgs a) {
  const int level = a.level0 + (int)blockIdx.x;
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const int T = a.lay[level].ntiles;
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
