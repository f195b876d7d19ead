// Semi-dense inverse-depth map kernels (reference: class depthMap, DepthPropagation.cpp). One thread per
// pixel; hypotheses are held structure-of-arrays. Per-pixel arithmetic follows the reference's f32 expression
// order with contraction off, so every stage except the global rescale sum is bit-reproducible on the CPU.
#pragma once
#include "ellc_context.hpp"
#include "ellc_kernels_gn.hpp"

namespace ellc {

// ExternVariable.h constants (line numbers in comments)
#define DM_MIN_ABS_GRAD_CREATE 1.0f        // :81
#define DM_MIN_ABS_GRAD_DECREASE 5.0f      // :82
#define DM_MIN_BLACKLIST (-1)              // :83
#define DM_MAX_DIFF_CONSTANT (40.0f * 40.0f)   // :85
#define DM_MAX_DIFF_GRAD_MULT (0.5f * 0.5f)    // :86
#define DM_VAR_RANDOM_INIT_INITIAL 0.125f  // :88
#define DM_MIN_EPL_GRAD_SQUARED (2.0f * 2.0f)      // :92
#define DM_MIN_EPL_LENGTH_SQUARED (1.0f * 1.0f)    // :93
#define DM_MIN_EPL_ANGLE_SQUARED (0.3f * 0.3f)     // :94
#define DM_MIN_DEPTH 0.05f                 // :98
#define DM_MAX_EPL_LENGTH_CROP 30.0f       // :101
#define DM_MIN_EPL_LENGTH_CROP 3.0f        // :102
#define DM_GRADIENT_SAMPLE_DIST 1.0f       // :105
#define DM_SAMPLE_POINT_TO_BORDER 7.0f     // :108
#ifndef DM_OBS_AHEAD
#define DM_OBS_AHEAD 4                     // taps in flight ahead of the line stereo's walk (do_line_stereo)
#endif
#define DM_MAX_ERROR_STEREO 1300.0f        // :111
#define DM_MIN_DISTANCE_ERROR_STEREO 1.5f  // :112
#define DM_STEREO_EPL_VAR_FAC 2.0f         // :115
#define DM_DIVISION_EPS 1e-10f             // :117
#define DM_CAMERA_PIXEL_NOISE 16           // :120
#define DM_VALIDITY_COUNTER_INITIAL_OBSERVE 5   // :122
#define DM_SUCC_VAR_INC_FAC 1.01f          // :124
#define DM_FAIL_VAR_INC_FAC 1.1f           // :125
#define DM_MAX_VAR (0.5f * 0.5f)           // :126
#define DM_VALIDITY_COUNTER_MAX 5.0f       // :133
#define DM_VALIDITY_COUNTER_MAX_VARIABLE 250.0f // :134
#define DM_VALIDITY_COUNTER_DEC 5.0f       // :135
#define DM_VALIDITY_COUNTER_INC 5.0f       // :136
#define DM_VAL_SUM_MIN_FOR_CREATE 30.0f    // :141
#define DM_VAL_SUM_MIN_FOR_UNBLACKLIST 100.0f   // :142
#define DM_VAL_SUM_MIN_FOR_KEEP 24.0f      // :143
#define DM_REG_DIST_VAR (0.075f * 0.075f * 1.0f * 1.0f)   // :145

struct Hyp {
  float id, ids, var, vars;
  int validity, bl;
  bool valid;
};
__device__ __forceinline__ Hyp hyp_load(const DepthSoA& s, int i) {
  Hyp h;
  h.id = s.invDepth[i]; h.ids = s.invDepthSmoothed[i]; h.var = s.variance[i]; h.vars = s.varianceSmoothed[i];
  h.validity = s.validity[i]; h.bl = s.blacklisted[i]; h.valid = s.isValid[i] != 0;
  return h;
}
__device__ __forceinline__ void hyp_store(const DepthSoA& s, int i, const Hyp& h) {
  s.invDepth[i] = h.id; s.invDepthSmoothed[i] = h.ids; s.variance[i] = h.var; s.varianceSmoothed[i] = h.vars;
  s.validity[i] = h.validity; s.blacklisted[i] = h.bl; s.isValid[i] = h.valid ? 1 : 0;
}

// u8 tap without the out-of-bounds sentinel (Frame.h:181, checkOutfBound = 0)
__device__ __forceinline__ float tap_plain(const uint8_t* img, int sw, int cols, int rows, float x, float y) {
  const Taps t = tap_point<false>(as_global(img), sw, cols, rows, x, y);
  return (t.I == -1.0f) ? 0.0f : t.I;   // four zero samples interpolate to 0 (NaN coordinates: reference UB)
}

// The same tap with its two loads issued ahead of its use. The line stereo is a chain of taps whose POSITIONS do not depend on
// the image (an epipolar walk of up to 30 + steps, one tap each): taken one at a time every step waits a trip to L2 / HBM
// (~1 us), and the launch lasts as long as its longest walk. raw_tap_load fetches the two rows of the 2 x 2 neighbourhood as
// (byte-unaligned) words for a position, clamped into the interior so that a look-ahead past the walk's end stays in the image;
// tap_plain_raw turns them into the tap when the position is interior — tap_point's interior expression, the same bits — and
// falls back to tap_plain (its per-tap bounds path) when it is not.
struct RawTap { uint32_t wb, wc; };
__device__ __forceinline__ RawTap raw_tap_load(const uint8_t* img, int sw, int cols, int rows, float x, float y) {
  const float fx0 = fminf(fmaxf(floorf(x), 1.0f), (float)(cols - 3)), fy0 = fminf(fmaxf(floorf(y), 1.0f), (float)(rows - 3));   // NaN -> 1
  const unsigned ob = __umul24((unsigned)(int)fy0, (unsigned)sw) + (unsigned)(int)fx0 - 1u;
  RawTap r;
  r.wb = load_u32_unaligned(as_global(img), ob);
  r.wc = load_u32_unaligned(as_global(img), ob + (unsigned)sw);
  return r;
}
__device__ __forceinline__ float tap_plain_raw(const uint8_t* img, int sw, int cols, int rows, float x, float y, const RawTap& r) {
  const float fx0 = floorf(x), fy0 = floorf(y);
  const bool interior = (__builtin_amdgcn_fmed3f(fx0, 1.0f, (float)(cols - 3)) == fx0) & (__builtin_amdgcn_fmed3f(fy0, 1.0f, (float)(rows - 3)) == fy0);
  float v;
  if (interior) {
    const float wx = x - fx0, wy = y - fy0;
    const float omx = 1.0f - wx, omy = 1.0f - wy;
    const float Pbb = byte_f32<1>(r.wb), Pbc = byte_f32<2>(r.wb), Pcb = byte_f32<1>(r.wc), Pcc = byte_f32<2>(r.wc);
    const float top = (omx * Pbb) + (wx * Pbc);
    const float btm = (omx * Pcb) + (wx * Pcc);
    v = (omy * top) + (wy * btm);
  } else {
    v = tap_plain(img, sw, cols, rows, x, y);
  }
  return v;
}

// raw_tap_load with the two loads issued by hand. The compiler's own wait-count insertion does not see through the walk's loop: it
// waits for EVERYTHING in flight before a slot is used (s_waitcnt vmcnt(0) in every step, whatever the look-ahead), so the walk
// keeps its own count: raw_tap_issue requests a slot, raw_tap_wait<N> waits until at most N younger requests are outstanding (vector
// loads return in issue order) and ties the slot's registers to the wait, so that nothing reads them earlier. A slot that may
// still be in flight when the walk ends is drained by raw_tap_wait<0> before its registers are reused.
__device__ __forceinline__ void raw_tap_issue(RawTap& r, const uint8_t* img, int sw, int cols, int rows, float x, float y) {
  const float fx0 = fminf(fmaxf(floorf(x), 1.0f), (float)(cols - 3)), fy0 = fminf(fmaxf(floorf(y), 1.0f), (float)(rows - 3));   // NaN -> 1
  const unsigned ob = __umul24((unsigned)(int)fy0, (unsigned)sw) + (unsigned)(int)fx0 - 1u, oc = ob + (unsigned)sw;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(r.wb) : "v"(ob), "s"(img) : "memory");
  asm volatile("global_load_dword %0, %1, %2" : "=v"(r.wc) : "v"(oc), "s"(img) : "memory");
}
template <int N>
__device__ __forceinline__ void raw_tap_wait(RawTap& r) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r.wb), "+v"(r.wc) : "n"(N));
}

// the tap at a position KNOWN to be interior (floor(x) in [1, cols - 3], floor(y) in [1, rows - 3]): tap_plain_raw's interior
// expression without its test and its fallback — straight-line code
__device__ __forceinline__ float tap_interior_raw(float x, float y, const RawTap& r) {
  const float fx0 = floorf(x), fy0 = floorf(y);
  const float wx = x - fx0, wy = y - fy0;
  const float omx = 1.0f - wx, omy = 1.0f - wy;
  const float Pbb = byte_f32<1>(r.wb), Pbc = byte_f32<2>(r.wb), Pcb = byte_f32<1>(r.wc), Pcc = byte_f32<2>(r.wc);
  const float top = (omx * Pbb) + (wx * Pbc);
  const float btm = (omx * Pcb) + (wx * Pcc);
  return (omy * top) + (wy * btm);
}

// ------------------------------------------------------------------------------------------------
// 5x5 / row-pair stencils over the hypothesis map. A block is DM_TX x DM_TY pixels; the four fields the stencils read
// (invDepth, variance, validity, isValid) of the tile plus a halo of DM_HALO pixels are staged in LDS once — 25 (or up to
// 35) neighbours per pixel straight from global memory made these kernels bound by the texture-address path, not by
// HBM. Pixels outside the image read as invalid (they are never inside a stencil the reference evaluates).
#define DM_TX 32
#define DM_TY 8
#define DM_HALO 3
#define DM_TW (DM_TX + 2 * DM_HALO)
#define DM_TH (DM_TY + 2 * DM_HALO)
struct DmTile {
  float id[DM_TH][DM_TW];
  float var[DM_TH][DM_TW];
  int validity[DM_TH][DM_TW];
  uint8_t valid[DM_TH][DM_TW];
};
// Tile of a stencil launch, XCD-aware: workgroups are dealt round-robin over the 8 XCDs (each with an L2 of its own), so with
// the plain numbering every tile's neighbours — whose pixels are its halo — sit on other XCDs and every halo line is fetched
// from memory once per XCD (r02 counters: dm_regularize read 3.2 x its algorithmic bytes). Renumbered, XCD k owns the k-th
// eighth of the tiles in raster order: a band of whole tile rows, whose halos its own L2 serves. Pure relabelling.
__device__ __forceinline__ bool dm_tile_of_block(int tiles_x, int tiles_total, int& bx, int& by) {
  const int per = (tiles_total + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || tile >= tiles_total) return false;
  by = tile / tiles_x;
  bx = tile - by * tiles_x;
  return true;
}
__device__ __forceinline__ void dm_tile_load_at(DmTile& t, const DepthSoA& in, int W, int H, int bx, int by) {
  const int x0 = bx * DM_TX - DM_HALO, y0 = by * DM_TY - DM_HALO;
  // (a fixed number of rounds over clamped cell numbers, every load unconditional: the loads of all rounds are in flight together —
  // with the cell count as the loop bound each round waited for its own; a clamped lane rewrites the last cell with its own value)
  constexpr int CELLS = DM_TW * DM_TH, NT = DM_TX * DM_TY;
#pragma unroll
  for (int r = 0; r < (CELLS + NT - 1) / NT; r++) {
    const int k = min((int)threadIdx.x + r * NT, CELLS - 1);
    const int ty = k / DM_TW, tx = k - ty * DM_TW;
    const int x = x0 + tx, y = y0 + ty;
    const bool inside = (x >= 0 && x < W && y >= 0 && y < H);
    const int j = inside ? (x + y * W) : 0;
    const uint8_t v = in.isValid[j];
    const float a = in.invDepth[j], b = in.variance[j];
    const int n = in.validity[j];
    t.valid[ty][tx] = inside ? v : (uint8_t)0;
    t.id[ty][tx] = inside ? a : 0.0f;
    t.var[ty][tx] = inside ? b : 0.0f;
    t.validity[ty][tx] = inside ? n : 0;
  }
  __syncthreads();
}

// The pixels of a tile that have work to do, gathered so that full waves process them: a semi-dense map holds a hypothesis at
// a fifth of its pixels, but nearly every wave of a pixel-per-lane launch holds at least one — and then runs the 25-neighbour
// loop with its IEEE divisions for all 64 lanes. Returns the number of candidates; list[k] = thread index of candidate k, in
// thread order (deterministic). Ends with a barrier.
__device__ __forceinline__ int dm_compact_candidates(bool cand, uint8_t* list, int* wave_count) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(cand);
  if (lane == 0) wave_count[wave] = __popcll(m);
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < (DM_TX * DM_TY) / 64; w++) {
    const int c = wave_count[w];
    if (w < wave) base += c;
    total += c;
  }
  if (cand) list[base + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = (uint8_t)threadIdx.x;
  __syncthreads();
  return total;
}

struct ExportPyrArgs {
  float* depth[4];     // levels 0..3 of the keyframe's depth pyramid (frame::depth_pyramid)
  float* var[4];       // depthMap::depthvararrptr
  int W, H, steps;     // steps = pyramid levels produced below level 0 (0..3)
};
__device__ __forceinline__ void depth_pyr_merge(const float d[4], const float v[4], float& od, float& ov) {
  float idepthSumsSum = 0.0f, ivarSumsSum = 0.0f;
  int num = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (v[q] > 0.0f) {
      const float ivar = 1.0f / v[q];
      ivarSumsSum += ivar;
      idepthSumsSum += ivar * 1.0f / d[q];
      num++;
    }
  }
  if (num > 0) {
    od = ivarSumsSum / idepthSumsSum;
    ov = (float)num / ivarSumsSum;
  } else {
    od = 0.0f;
    ov = -1.0f;
  }
}

// updateDepthImage for a block's own 32 x 8 tile, at the end of the kernel that produced the values: level 0 of the keyframe's depth /
// variance planes from the pixel's final smoothed values (nvalid: its flag after the border's invalidation, :1254-1315) and the
// 16 x 4, 8 x 2 and 4 x 1 cells of the next `steps` pyramid levels the tile covers (dm_export_pyramid's arithmetic; image sizes
// that are multiples of 32 x 8 and halve exactly `steps` times). Every thread of the block calls it.
__device__ __forceinline__ void dm_tile_export(const ExportPyrArgs& ex, int W, int H, int bx, int by, bool inside, int i, bool nvalid, float ids, float vars) {
  // ping-pong planes of the tile's pyramid cells: level l in [l & 1] (dm_export_pyramid with a 32 x 8 tile)
  __shared__ float ld[2][DM_TX * DM_TY], lv[2][DM_TX * DM_TY];
  float d = 0.0f, v = -1.0f;
  if (inside) {
    if (nvalid && ids >= -0.05f) {
      d = 1.0f / ids;
      v = vars;
    }
    ex.depth[0][i] = d;
    ex.var[0][i] = v;
  }
  ld[0][threadIdx.x] = d;
  lv[0][threadIdx.x] = v;
  __syncthreads();
  int ew = DM_TX, eh = DM_TY;
  for (int l = 1; l <= ex.steps; l++) {
    const int w2 = ew >> 1, h2 = eh >> 1;   // the tile's cells at level l
    const int wl = W >> l, hl = H >> l;
    const float* sd = ld[(l - 1) & 1];
    const float* sv = lv[(l - 1) & 1];
    if ((int)threadIdx.x < w2 * h2) {
      const int cy = (int)threadIdx.x / w2, cx = (int)threadIdx.x - cy * w2;
      const int q0 = (2 * cy) * ew + 2 * cx;
      const float d4[4] = {sd[q0], sd[q0 + 1], sd[q0 + ew], sd[q0 + ew + 1]};
      const float v4[4] = {sv[q0], sv[q0 + 1], sv[q0 + ew], sv[q0 + ew + 1]};
      float od, ov;
      depth_pyr_merge(d4, v4, od, ov);
      const int gx = ((bx * DM_TX) >> l) + cx, gy = ((by * DM_TY) >> l) + cy;
      if (gx < wl && gy < hl) {
        ex.depth[l][gx + gy * wl] = od;
        ex.var[l][gx + gy * wl] = ov;
      }
      ld[l & 1][cy * w2 + cx] = od;
      lv[l & 1][cy * w2 + cx] = ov;
    }
    __syncthreads();
    ew = w2; eh = h2;
  }
}

// regularizeDepthMap's 25-neighbour stencil (:1452-1530) for the pixel at index c of LDS planes with PITCH cells per row. The
// reference's two `continue`s become per-lane selects: a wave runs every neighbour anyway (one lane with a hypothesis there is
// enough, and a wave of gathered candidates always has one), and without the branches the 100 LDS reads of a pixel are issued
// ahead of the arithmetic instead of one dependent round trip per test. A rejected neighbour's terms are computed and dropped
// by the select, so every sum receives the same addends in the same order as the reference's loop. Returns 1: smoothed values in
// out_ids / out_vars, 2: invalidated + blacklist step, 3: invalidated (occluded). REMOVE_OCC = the reference's removeOcclusions.
// VALUES = false: only the verdict, which needs the validity sum and the occlusion counts but none of the 25 divisions — what
// the one-launch form (dm_reg_fill_reg) owes the pixels of its ring, whose smoothed values nobody reads.
template <int PITCH, bool REMOVE_OCC, bool VALUES>
__device__ __forceinline__ int dm_stencil25(const float* id, const float* var, const int* validity, const uint8_t* valid, int c, float& out_ids, float& out_vars) {
  const float did = id[c], dvar = var[c];
  float sum = 0.0f, val_sum = 0.0f, sumIvar = 0.0f;
  int numOccluding = 0, numNotOccluding = 0;
#pragma unroll
  for (int dx = -2; dx <= 2; dx++) {
    // (one column's masks at a time: scheduled as one region, the 25 pairs of lane masks outgrow the scalar registers and are
    // spilled through v_writelane)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dy = -2; dy <= 2; dy++) {
      const int q = c + dy * PITCH + dx;
      const bool there = valid[q] != 0;
      const float sid = id[q], svar = var[q];
      const float diff = sid - did;
      const bool apart = 1.0f * diff * diff > svar + dvar;
      const bool take = there && !apart;
      if (REMOVE_OCC) {
        numOccluding += (there && apart && sid > did) ? 1 : 0;
        numNotOccluding += take ? 1 : 0;
      }
      const float vs = val_sum + (float)validity[q];
      val_sum = take ? vs : val_sum;
      if (VALUES) {
        const float distFac = (float)(dx * dx + dy * dy) * DM_REG_DIST_VAR;
        const float ivar = 1.0f / (svar + distFac);
        const float s1 = sum + sid * ivar, s2 = sumIvar + ivar;
        sum = take ? s1 : sum;
        sumIvar = take ? s2 : sumIvar;
      }
    }
  }
  // (the values are formed whatever the verdict and the callers store them whatever the verdict: behind a test, the compiler
  // moves the 25 divisions behind it as a second pass over the neighbours and carries the 50 lane masks there through spills)
  if (VALUES) {
    sum = sum / sumIvar;
    out_ids = unzero_f(sum);
    out_vars = 1.0f / sumIvar;
  }
  return (val_sum < (float)(int)DM_VAL_SUM_MIN_FOR_KEEP) ? 2 : (REMOVE_OCC && numOccluding > numNotOccluding) ? 3 : 1;
}

// depthMap::regularizeDepthMap (:1436-1543). The stencil reads the snapshot's invDepth / variance / validity / isValid, a
// pixel's update writes only its own invDepthSmoothed / varianceSmoothed / blacklisted — which no stencil reads — and isValid:
// so the map is updated IN PLACE and only the new validity flags go to a plane of their own (valid_out, which the caller then
// swaps in): 13 bytes read (x the halo) and at most 13 written per pixel instead of 25 + 25 through a second copy of the map.
// (The tracked frame's and createKeyFrame's regularisations run inside dm_fill_reg / dm_reg_fill_reg; this is the stage alone.)
__global__ __launch_bounds__(DM_TX * DM_TY) void dm_regularize(DepthSoA s, uint8_t* __restrict__ valid_out, int W, int H, int removeOcclusions,
                                                               int tiles_x, int tiles_total, const int* __restrict__ gate) {
  __shared__ DmTile t;
  __shared__ uint8_t list[DM_TX * DM_TY];
  __shared__ int wave_count[(DM_TX * DM_TY) / 64];
  __shared__ float r_ids[DM_TX * DM_TY], r_vars[DM_TX * DM_TY];
  __shared__ uint8_t r_code[DM_TX * DM_TY];   // 0: unchanged, 1: smoothed values, 2: invalidated + blacklist step, 3: invalidated (occluded)
  int bx, by;
  if (!dm_tile_of_block(tiles_x, tiles_total, bx, by)) return;
  dm_tile_load_at(t, s, W, H, bx, by);
  const int tx = threadIdx.x & (DM_TX - 1), ty = threadIdx.x / DM_TX;
  const int x = bx * DM_TX + tx, y = by * DM_TY + ty;
  const bool inside = (x < W && y < H);
  const bool dvalid = t.valid[ty + DM_HALO][tx + DM_HALO] != 0;
  const bool open = (gate == nullptr) || (*gate != 0);   // a closed gate: the map passes through unchanged
  const bool cand = open && inside && y >= 3 && y < H - 3 && x >= 2 && x < W - 2 && dvalid;
  r_code[threadIdx.x] = 0;
  const int ncand = dm_compact_candidates(cand, list, wave_count);
  for (int k = threadIdx.x; k < ncand; k += DM_TX * DM_TY) {
    const int p = list[k];
    const int c = (p / DM_TX + DM_HALO) * DM_TW + (p & (DM_TX - 1)) + DM_HALO;
    float ids = 0.0f, vars = 0.0f;
    const int code = removeOcclusions ? dm_stencil25<DM_TW, true, true>(&t.id[0][0], &t.var[0][0], &t.validity[0][0], &t.valid[0][0], c, ids, vars)
                                      : dm_stencil25<DM_TW, false, true>(&t.id[0][0], &t.var[0][0], &t.validity[0][0], &t.valid[0][0], c, ids, vars);
    r_code[p] = (uint8_t)code;
    r_ids[p] = ids; r_vars[p] = vars;   // read only where the code is 1
  }
  __syncthreads();
  if (!inside) return;
  const int i = x + y * W;
  const int code = r_code[threadIdx.x];
  valid_out[i] = (dvalid && code < 2) ? 1 : 0;
  if (code == 1) {
    s.invDepthSmoothed[i] = r_ids[threadIdx.x];
    s.varianceSmoothed[i] = r_vars[threadIdx.x];
  } else if (code == 2) {
    s.blacklisted[i] = s.blacklisted[i] - 1;
  }
}

// depthMap::fillDepthHoles (:1317-1400) with buildValIntegralBuffer (:1403-1432) folded in: the reference
// indexes a per-row prefix sum as if it were a 2-D integral image, which evaluates to
//   val = sum_{x-2..x+2} validity(row y+2) - sum_{x-2..x+2} validity(row y-3)
// with rows outside [3, H-3) contributing 0 (never written, zero-initialised).
// A filled hole writes fields its neighbours' stencils read, so this stage keeps the second copy of the map (in -> out). The
// few pixels that pass the validity test are gathered (dm_compact_candidates) before the 25-neighbour average with its 50
// divisions.
__global__ __launch_bounds__(DM_TX * DM_TY) void dm_fill_holes(DepthSoA in, DepthSoA out, const float* __restrict__ maxgrad, int W, int H,
                                                               int tiles_x, int tiles_total, const int* __restrict__ gate) {
  __shared__ DmTile t;
  __shared__ uint8_t list[DM_TX * DM_TY];
  __shared__ int wave_count[(DM_TX * DM_TY) / 64];
  __shared__ float r_id[DM_TX * DM_TY];
  int bx, by;
  if (!dm_tile_of_block(tiles_x, tiles_total, bx, by)) return;
  dm_tile_load_at(t, in, W, H, bx, by);
  const int tx = threadIdx.x & (DM_TX - 1), ty = threadIdx.x / DM_TX;
  const int x = bx * DM_TX + tx, y = by * DM_TY + ty;
  const bool inside = (x < W && y < H);
  const int i = inside ? (x + y * W) : 0;
  const int cx = tx + DM_HALO, cy = ty + DM_HALO;
  Hyp d;
  d.valid = false; d.bl = 0;
  if (inside) d = hyp_load(in, i);
  bool fill = false;
  const bool open = (gate == nullptr) || (*gate != 0);
  if (open && inside && y >= 3 && y < H - 3 && x >= 3 && x < W - 2 && !d.valid && !(maxgrad[i] < DM_MIN_ABS_GRAD_DECREASE)) {
    int val = 0;
    const int ya = y + 2, yb = y - 3;
    if (ya >= 3 && ya < H - 3)
      for (int dx = -2; dx <= 2; dx++) { if (t.valid[cy + 2][cx + dx]) val += t.validity[cy + 2][cx + dx]; }
    if (yb >= 3 && yb < H - 3)
      for (int dx = -2; dx <= 2; dx++) { if (t.valid[cy - 3][cx + dx]) val -= t.validity[cy - 3][cx + dx]; }
    fill = (d.bl >= DM_MIN_BLACKLIST && (float)val > DM_VAL_SUM_MIN_FOR_CREATE) || (float)val > DM_VAL_SUM_MIN_FOR_UNBLACKLIST;
  }
  const int ncand = dm_compact_candidates(fill, list, wave_count);
  for (int k = threadIdx.x; k < ncand; k += DM_TX * DM_TY) {
    const int p = list[k];
    const int px = (p & (DM_TX - 1)) + DM_HALO, py = p / DM_TX + DM_HALO;
    float sumIdepthObs = 0.0f, sumIVarObs = 0.0f;
    for (int dy = -2; dy < 3; dy++)
      for (int dx = -2; dx < 3; dx++) {
        if (!t.valid[py + dy][px + dx]) continue;
        const float v = t.var[py + dy][px + dx];
        sumIdepthObs += t.id[py + dy][px + dx] / v;
        sumIVarObs += 1.0f / v;
      }
    r_id[p] = unzero_f(sumIdepthObs / sumIVarObs);
  }
  __syncthreads();
  if (!inside) return;
  if (fill) {
    d.id = r_id[threadIdx.x];
    d.var = DM_VAR_RANDOM_INIT_INITIAL;
    d.validity = 0;
    d.valid = true;
    d.bl = 0;
    d.ids = -1.0f;
    d.vars = -1.0f;
  }
  hyp_store(out, i, d);
}

// ------------------------------------------------------------------------------------------------
// createKeyFrame's three stencil stages in ONE launch (r04): regularizeDepthMap(removeOcclusions) -> fillDepthHoles ->
// regularizeDepthMap(false) (DepthPropagation.cpp:1775-1777). A block owns a 32 x 8 tile and recomputes the earlier stages on the
// ring of pixels the later ones read: the second regularisation reads the filled map two pixels around the tile, the fill reads
// the first regularisation's validity flags two columns / three rows up and two down around that, and the first regularisation the
// snapshot two pixels around that — a snapshot of (32 + 12) x (8 + 13) pixels in LDS (13 KB). Per pixel the operations and their
// order are those of dm_regularize and dm_fill_holes (the three-launch form stays for the other callers and is what the tests
// compare this one with, bit for bit); candidates of every stage are gathered first so that full waves run the 25-neighbour loops.
// The result goes to the other copy of the map (the stencils of neighbouring blocks read this one).
#define DM_FX 6                  // snapshot halo: columns left / right
#define DM_FYU 7                 //   rows above
#define DM_FYD 6                 //   rows below
#define DM_FW (DM_TX + 2 * DM_FX)
#define DM_FH (DM_TY + DM_FYU + DM_FYD)
struct DmFused {
  float id[DM_FH][DM_FW], var[DM_FH][DM_FW];
  int validity[DM_FH][DM_FW];
  uint8_t valid0[DM_FH][DM_FW], valid1[DM_FH][DM_FW], valid2[DM_FH][DM_FW];
  uint8_t code1[DM_FH][DM_FW];            // first regularisation: 0 unchanged, 1 smoothed, 2 invalidated + blacklist step, 3 invalidated (occluded)
  uint16_t list[DM_FH * DM_FW], list_own[DM_TX * DM_TY];
  int count, count_own, count_fill;
  float ids1[DM_TX * DM_TY], vars1[DM_TX * DM_TY];   // the tile's own smoothed values of the first / second regularisation
  float ids3[DM_TX * DM_TY], vars3[DM_TX * DM_TY];
  uint8_t code3[DM_TX * DM_TY];
};
// regularizeDepthMap's stencil for the pixel at (cx, cy) of the snapshot
template <bool VALUES>
__device__ __forceinline__ int dm_reg_stencil(const DmFused& t, const uint8_t (*valid)[DM_FW], int cx, int cy, int removeOcclusions, float& out_ids, float& out_vars) {
  const int c = cy * DM_FW + cx;
  return removeOcclusions ? dm_stencil25<DM_FW, true, VALUES>(&t.id[0][0], &t.var[0][0], &t.validity[0][0], &valid[0][0], c, out_ids, out_vars)
                          : dm_stencil25<DM_FW, false, VALUES>(&t.id[0][0], &t.var[0][0], &t.validity[0][0], &valid[0][0], c, out_ids, out_vars);
}
__global__ __launch_bounds__(DM_TX * DM_TY) void dm_reg_fill_reg(DepthSoA in, DepthSoA out, const float* __restrict__ maxgrad, int W, int H, int removeOcclusions,
                                                                 int tiles_x, int tiles_total, double* __restrict__ part) {
  __shared__ DmFused t;
  int bx, by;
  if (!dm_tile_of_block(tiles_x, tiles_total, bx, by)) return;
  constexpr int NT = DM_TX * DM_TY;
  const int x0 = bx * DM_TX - DM_FX, y0 = by * DM_TY - DM_FYU;   // image position of snapshot cell (0, 0)
#pragma unroll
  for (int r = 0; r < (DM_FW * DM_FH + NT - 1) / NT; r++) {   // (as dm_tile_load_at: every round's loads in flight together)
    const int k = min((int)threadIdx.x + r * NT, DM_FW * DM_FH - 1);
    const int ty = k / DM_FW, tx = k - ty * DM_FW;
    const int x = x0 + tx, y = y0 + ty;
    const bool inside = (x >= 0 && x < W && y >= 0 && y < H);
    const int j = inside ? (x + y * W) : 0;
    const uint8_t lv = in.isValid[j];
    const float a = in.invDepth[j], b = in.variance[j];
    const int n = in.validity[j];
    const uint8_t v = inside ? lv : (uint8_t)0;
    t.valid0[ty][tx] = v; t.valid1[ty][tx] = v; t.valid2[ty][tx] = v;
    t.code1[ty][tx] = 0;
    t.id[ty][tx] = inside ? a : 0.0f;
    t.var[ty][tx] = inside ? b : 0.0f;
    t.validity[ty][tx] = inside ? n : 0;
  }
  if (threadIdx.x == 0) { t.count = 0; t.count_own = 0; t.count_fill = 0; }
  __syncthreads();
  // ---- stage 1: regularizeDepthMap(removeOcclusions) where the fill will look: columns -4 .. +4, rows -5 .. +4 around the tile.
  // The tile's own candidates (smoothed values wanted) and the ring's (verdict only) go to two lists, run by different waves.
  constexpr int R1W = DM_TX + 8, R1H = DM_TY + 9;   // columns -4 .. +4, rows -5 .. +4
  for (int q = threadIdx.x; q < R1W * R1H; q += NT) {
    const int ry = q / R1W, tx = q - ry * R1W + (DM_FX - 4), ty = ry + (DM_FYU - 5);
    const int k = ty * DM_FW + tx;
    const int x = x0 + tx, y = y0 + ty;
    if (!(x >= 2 && x < W - 2 && y >= 3 && y < H - 3 && t.valid0[ty][tx])) continue;
    const int ox = tx - DM_FX, oy = ty - DM_FYU;
    if (ox >= 0 && ox < DM_TX && oy >= 0 && oy < DM_TY) t.list_own[atomicAdd(&t.count_own, 1)] = (uint16_t)k;
    else t.list[atomicAdd(&t.count, 1)] = (uint16_t)k;
  }
  __syncthreads();
  const int n1_own = t.count_own, n1_own_pad = (n1_own + 63) & ~63, n1 = n1_own_pad + t.count;
  for (int k = threadIdx.x; k < n1; k += NT) {
    if (k < n1_own_pad) {   // wave-uniform: the padding is a whole number of waves
      if (k >= n1_own) continue;
      const int c = t.list_own[k], cy = c / DM_FW, cx = c - cy * DM_FW;
      float ids = 0.0f, vars = 0.0f;
      const int code = dm_reg_stencil<true>(t, t.valid0, cx, cy, removeOcclusions, ids, vars);
      t.code1[cy][cx] = (uint8_t)code;
      if (code >= 2) { t.valid1[cy][cx] = 0; t.valid2[cy][cx] = 0; }
      const int o = (cy - DM_FYU) * DM_TX + (cx - DM_FX);
      t.ids1[o] = ids; t.vars1[o] = vars;   // read only where the code is 1
    } else {
      const int c = t.list[k - n1_own_pad], cy = c / DM_FW, cx = c - cy * DM_FW;
      float ids, vars;
      const int code = dm_reg_stencil<false>(t, t.valid0, cx, cy, removeOcclusions, ids, vars);
      t.code1[cy][cx] = (uint8_t)code;
      if (code >= 2) { t.valid1[cy][cx] = 0; t.valid2[cy][cx] = 0; }
    }
  }
  __syncthreads();
  // ---- stage 2: fillDepthHoles two pixels around the tile (dm_fill_holes' test and average on the flags of stage 1)
  constexpr int R2W = DM_TX + 4, R2H = DM_TY + 4;   // two pixels around the tile
#pragma unroll
  for (int r = 0; r < (R2W * R2H + NT - 1) / NT; r++) {
    const int q = (int)threadIdx.x + r * NT;
    const int qq = min(q, R2W * R2H - 1);
    const int ry = qq / R2W, tx = qq - ry * R2W + (DM_FX - 2), ty = ry + (DM_FYU - 2);
    const int k = ty * DM_FW + tx;
    const int x = x0 + tx, y = y0 + ty;
    const bool in_test = q < R2W * R2H && x >= 3 && x < W - 2 && y >= 3 && y < H - 3;
    const int i = in_test ? x + y * W : 0;
    const float mg = maxgrad[i];            // both loads before the tests (one round trip, not one per test)
    const int bl0 = in.blacklisted[i];
    if (!in_test || t.valid1[ty][tx]) continue;
    if (mg < DM_MIN_ABS_GRAD_DECREASE) continue;
    int val = 0;
    const int ya = y + 2, yb = y - 3;
    if (ya >= 3 && ya < H - 3)
      for (int dx = -2; dx <= 2; dx++) { if (t.valid1[ty + 2][tx + dx]) val += t.validity[ty + 2][tx + dx]; }
    if (yb >= 3 && yb < H - 3)
      for (int dx = -2; dx <= 2; dx++) { if (t.valid1[ty - 3][tx + dx]) val -= t.validity[ty - 3][tx + dx]; }
    const int bl1 = bl0 - (t.code1[ty][tx] == 2 ? 1 : 0);   // the blacklist counter after stage 1
    if ((bl1 >= DM_MIN_BLACKLIST && (float)val > DM_VAL_SUM_MIN_FOR_CREATE) || (float)val > DM_VAL_SUM_MIN_FOR_UNBLACKLIST)
      t.list[atomicAdd(&t.count_fill, 1)] = (uint16_t)k;
  }
  __syncthreads();
  const int n2 = t.count_fill;
  for (int k = threadIdx.x; k < n2; k += NT) {
    const int c = t.list[k], py = c / DM_FW, px = c - py * DM_FW;
    float sumIdepthObs = 0.0f, sumIVarObs = 0.0f;
    for (int dy = -2; dy < 3; dy++)
      for (int dx = -2; dx < 3; dx++) {
        if (!t.valid1[py + dy][px + dx]) continue;
        const float v = t.var[py + dy][px + dx];
        sumIdepthObs += t.id[py + dy][px + dx] / v;
        sumIVarObs += 1.0f / v;
      }
    const float nid = unzero_f(sumIdepthObs / sumIVarObs);
    // a filled pixel was invalid in stage 1's flags, which is all the averages above look at: its cell can be written at once
    t.id[py][px] = nid;
    t.var[py][px] = DM_VAR_RANDOM_INIT_INITIAL;
    t.validity[py][px] = 0;
    t.valid2[py][px] = 2;   // valid, and filled here
  }
  __syncthreads();
  // ---- stage 3: regularizeDepthMap(false) on the tile, on the filled map
  const int tx = threadIdx.x & (DM_TX - 1), ty = threadIdx.x / DM_TX;
  const int cx = tx + DM_FX, cy = ty + DM_FYU;
  const int x = bx * DM_TX + tx, y = by * DM_TY + ty;
  const bool inside = (x < W && y < H);
  t.code3[threadIdx.x] = 0;
  const bool cand3 = inside && y >= 3 && y < H - 3 && x >= 2 && x < W - 2 && t.valid2[cy][cx];
  __shared__ uint8_t list3[NT];
  __shared__ int wave_count[NT / 64];
  const int n3 = dm_compact_candidates(cand3, list3, wave_count);
  for (int k = threadIdx.x; k < n3; k += NT) {
    const int p = list3[k];
    float ids = 0.0f, vars = 0.0f;
    const int code = dm_reg_stencil<true>(t, t.valid2, (p & (DM_TX - 1)) + DM_FX, p / DM_TX + DM_FYU, 0, ids, vars);
    t.code3[p] = (uint8_t)code;
    t.ids3[p] = ids; t.vars3[p] = vars;   // read only where the code is 1
  }
  __syncthreads();
  // ---- the pixel's state after the three stages, into the other copy of the map
  double acc = 0.0, cnt = 0.0;
  if (inside) {
    const int i = x + y * W;
    Hyp d = hyp_load(in, i);
    const int c1 = t.code1[cy][cx], c3 = t.code3[threadIdx.x];
    if (c1 == 1) { d.ids = t.ids1[threadIdx.x]; d.vars = t.vars1[threadIdx.x]; }
    else if (c1 == 2) d.bl = d.bl - 1;
    if (t.valid2[cy][cx] == 2) {   // filled by stage 2
      d.id = t.id[cy][cx];
      d.var = DM_VAR_RANDOM_INIT_INITIAL;
      d.validity = 0;
      d.bl = 0;
      d.ids = -1.0f;
      d.vars = -1.0f;
    }
    if (c3 == 1) { d.ids = t.ids3[threadIdx.x]; d.vars = t.vars3[threadIdx.x]; }
    else if (c3 == 2) d.bl = d.bl - 1;
    d.valid = t.valid2[cy][cx] != 0 && c3 < 2;
    hyp_store(out, i, d);
    if (d.valid) { acc = (double)d.ids; cnt = 1.0; }
  }
  // makeInvDepthOne's sum (:1546-1587), first stage: this tile's sum of the smoothed inverse depths and their count (f64, fixed
  // order) for dm_export_pyramid<true>, which finishes the sum, rescales and exports
  if (part == nullptr) return;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { acc += __shfl_xor(acc, m, 64); cnt += __shfl_xor(cnt, m, 64); }
  __shared__ double wsum[NT / 64], wcnt[NT / 64];
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = acc; wcnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = wsum[0], n = wcnt[0];
    for (int w = 1; w < NT / 64; w++) { a += wsum[w]; n += wcnt[w]; }
    const int tile = by * tiles_x + bx;
    part[2 * tile] = a; part[2 * tile + 1] = n;
  }
}

// ------------------------------------------------------------------------------------------------
// doRegularization(removeOcclusions) = fillDepthHoles + regularizeDepthMap (:1627-1635) in ONE launch, the tracked frame's form (and,
// with EXPORT, updateDepthImage for the tile behind it: dm_tile_export): dm_reg_fill_reg without its first stage. The
// regularisation reads the filled map two pixels around the tile and the fill the flags two columns / three rows up and two down
// around that: a snapshot of (32 + 8) x (8 + 9) pixels. A closed gate (ellc_track_frame) passes the map through unchanged — to the
// other copy, which the caller swaps in either way.
#define DM_GX 4
#define DM_GYU 5
#define DM_GYD 4
#define DM_GW (DM_TX + 2 * DM_GX)
#define DM_GH (DM_TY + DM_GYU + DM_GYD)
struct DmFillReg {
  float id[DM_GH][DM_GW], var[DM_GH][DM_GW];
  int validity[DM_GH][DM_GW];
  uint8_t valid0[DM_GH][DM_GW], valid2[DM_GH][DM_GW];   // flags before / after the fill (2: filled here)
  uint16_t list[(DM_TX + 4) * (DM_TY + 4)];
  int count_fill;
  float ids3[DM_TX * DM_TY], vars3[DM_TX * DM_TY];
  uint8_t code3[DM_TX * DM_TY];
};
template <bool EXPORT>
__global__ __launch_bounds__(DM_TX * DM_TY) void dm_fill_reg(DepthSoA in, DepthSoA out, const float* __restrict__ maxgrad, int W, int H, int removeOcclusions,
                                                             int tiles_x, int tiles_total, const int* __restrict__ gate, ExportPyrArgs ex) {
  __shared__ DmFillReg t;
  int bx, by;
  if (!dm_tile_of_block(tiles_x, tiles_total, bx, by)) return;
  constexpr int NT = DM_TX * DM_TY;
  const int x0 = bx * DM_TX - DM_GX, y0 = by * DM_TY - DM_GYU;   // image position of snapshot cell (0, 0)
  const bool open = (gate == nullptr) || (*gate != 0);
#pragma unroll
  for (int r = 0; r < (DM_GW * DM_GH + NT - 1) / NT; r++) {   // (as dm_tile_load_at: every round's loads in flight together)
    const int k = min((int)threadIdx.x + r * NT, DM_GW * DM_GH - 1);
    const int ty = k / DM_GW, tx = k - ty * DM_GW;
    const int x = x0 + tx, y = y0 + ty;
    const bool inside = (x >= 0 && x < W && y >= 0 && y < H);
    const int j = inside ? (x + y * W) : 0;
    const uint8_t lv = in.isValid[j];
    const float a = in.invDepth[j], b = in.variance[j];
    const int n = in.validity[j];
    const uint8_t v = inside ? lv : (uint8_t)0;
    t.valid0[ty][tx] = v; t.valid2[ty][tx] = v;
    t.id[ty][tx] = inside ? a : 0.0f;
    t.var[ty][tx] = inside ? b : 0.0f;
    t.validity[ty][tx] = inside ? n : 0;
  }
  if (threadIdx.x == 0) t.count_fill = 0;
  __syncthreads();
  // ---- fillDepthHoles two pixels around the tile (dm_fill_holes' test and average)
  constexpr int R2W = DM_TX + 4, R2H = DM_TY + 4;
#pragma unroll
  for (int r = 0; r < (R2W * R2H + NT - 1) / NT; r++) {
    const int q = (int)threadIdx.x + r * NT;
    const int qq = min(q, R2W * R2H - 1);
    const int ry = qq / R2W, tx = qq - ry * R2W + (DM_GX - 2), ty = ry + (DM_GYU - 2);
    const int x = x0 + tx, y = y0 + ty;
    const bool in_test = open && q < R2W * R2H && x >= 3 && x < W - 2 && y >= 3 && y < H - 3;
    const int i = in_test ? x + y * W : 0;
    const float mg = maxgrad[i];            // both loads before the tests (one round trip, not one per test)
    const int bl0 = in.blacklisted[i];
    if (!in_test || t.valid0[ty][tx]) continue;
    if (mg < DM_MIN_ABS_GRAD_DECREASE) continue;
    int val = 0;
    const int ya = y + 2, yb = y - 3;
    if (ya >= 3 && ya < H - 3)
      for (int dx = -2; dx <= 2; dx++) { if (t.valid0[ty + 2][tx + dx]) val += t.validity[ty + 2][tx + dx]; }
    if (yb >= 3 && yb < H - 3)
      for (int dx = -2; dx <= 2; dx++) { if (t.valid0[ty - 3][tx + dx]) val -= t.validity[ty - 3][tx + dx]; }
    if ((bl0 >= DM_MIN_BLACKLIST && (float)val > DM_VAL_SUM_MIN_FOR_CREATE) || (float)val > DM_VAL_SUM_MIN_FOR_UNBLACKLIST)
      t.list[atomicAdd(&t.count_fill, 1)] = (uint16_t)(ty * DM_GW + tx);
  }
  __syncthreads();
  const int n2 = t.count_fill;
  for (int k = threadIdx.x; k < n2; k += NT) {
    const int c = t.list[k], py = c / DM_GW, px = c - py * DM_GW;
    float sumIdepthObs = 0.0f, sumIVarObs = 0.0f;
    for (int dy = -2; dy < 3; dy++)
      for (int dx = -2; dx < 3; dx++) {
        if (!t.valid0[py + dy][px + dx]) continue;
        const float v = t.var[py + dy][px + dx];
        sumIdepthObs += t.id[py + dy][px + dx] / v;
        sumIVarObs += 1.0f / v;
      }
    const float nid = unzero_f(sumIdepthObs / sumIVarObs);
    // a filled pixel was invalid in the flags the averages look at: its cell can be written at once
    t.id[py][px] = nid;
    t.var[py][px] = DM_VAR_RANDOM_INIT_INITIAL;
    t.validity[py][px] = 0;
    t.valid2[py][px] = 2;   // valid, and filled here
  }
  __syncthreads();
  // ---- regularizeDepthMap on the tile, on the filled map
  const int tx = threadIdx.x & (DM_TX - 1), ty = threadIdx.x / DM_TX;
  const int cx = tx + DM_GX, cy = ty + DM_GYU;
  const int x = bx * DM_TX + tx, y = by * DM_TY + ty;
  const bool inside = (x < W && y < H);
  t.code3[threadIdx.x] = 0;
  const bool cand3 = open && inside && y >= 3 && y < H - 3 && x >= 2 && x < W - 2 && t.valid2[cy][cx];
  __shared__ uint8_t list3[NT];
  __shared__ int wave_count[NT / 64];
  const int n3 = dm_compact_candidates(cand3, list3, wave_count);
  for (int k = threadIdx.x; k < n3; k += NT) {
    const int p = list3[k];
    float ids = 0.0f, vars = 0.0f;
    const int c = (p / DM_TX + DM_GYU) * DM_GW + (p & (DM_TX - 1)) + DM_GX;
    const int code = removeOcclusions ? dm_stencil25<DM_GW, true, true>(&t.id[0][0], &t.var[0][0], &t.validity[0][0], &t.valid2[0][0], c, ids, vars)
                                      : dm_stencil25<DM_GW, false, true>(&t.id[0][0], &t.var[0][0], &t.validity[0][0], &t.valid2[0][0], c, ids, vars);
    t.code3[p] = (uint8_t)code;
    t.ids3[p] = ids; t.vars3[p] = vars;   // read only where the code is 1
  }
  __syncthreads();
  // ---- the pixel's state after both stages, into the other copy of the map
  const int i = x + y * W;
  Hyp d;
  d.valid = false; d.ids = 0.0f; d.vars = 0.0f;
  if (inside) {
    d = hyp_load(in, i);
    const int c3 = t.code3[threadIdx.x];
    if (t.valid2[cy][cx] == 2) {   // filled
      d.id = t.id[cy][cx];
      d.var = DM_VAR_RANDOM_INIT_INITIAL;
      d.validity = 0;
      d.bl = 0;
      d.ids = -1.0f;
      d.vars = -1.0f;
    }
    if (c3 == 1) { d.ids = t.ids3[threadIdx.x]; d.vars = t.vars3[threadIdx.x]; }
    else if (c3 == 2) d.bl = d.bl - 1;
    d.valid = t.valid2[cy][cx] != 0 && c3 < 2;
    if (EXPORT && (y < 3 || y >= H - 3 || x < 3 || x >= W - 3)) d.valid = false;   // updateDepthImage clears the border's flags (:1254-1315)
    hyp_store(out, i, d);
  }
  // (a closed gate leaves the keyframe's planes alone as well: the alignment that has not ended yet — its continuation — reads them,
  // and they need not be this map's export: ellc_keyframe_set_depth. Block-uniform.)
  if constexpr (EXPORT) { if (open) dm_tile_export(ex, W, H, bx, by, inside, i, d.valid, d.ids, d.vars); }
}

// depthMap::updateDepthImage (:1254-1315), per-pixel part: invalidate the 3-px border, export level 0
__global__ void dm_export_level0(DepthSoA s, float* __restrict__ depthMat, float* __restrict__ vararr, int W, int H) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) return;
  const int i = x + y * W;
  bool valid = s.isValid[i] != 0;
  if (y < 3 || y >= H - 3 || x < 3 || x >= W - 3) {
    valid = false;
    s.isValid[i] = 0;
  }
  const float ids = s.invDepthSmoothed[i];
  if (valid && ids >= -0.05f) {
    depthMat[i] = 1.0f / ids;
    vararr[i] = s.varianceSmoothed[i];
  } else {
    depthMat[i] = 0.0f;
    vararr[i] = -1.0f;
  }
}

// updateDepthImage's export (dm_export_level0) and up to three levels of buildInvVarDepth (depth_pyr_level,
// ellc_kernels_image.hpp) in ONE launch: a block exports a 32 x 32 tile of level 0 and reduces it 2x2 -> 16x16 -> 8x8 -> 4x4
// through LDS, writing every level. Only for level sizes that halve exactly (W >> l == 2 (W >> (l + 1)) for the levels
// produced): the reference reads a source level with the stride 2 * (destination width), which is the source's own width
// exactly then; other sizes take the per-level kernels. Same operations per value as the kernels it replaces.
// RESCALE (createKeyFrame): makeInvDepthOne (:1546-1587) goes first, in the same launch — every block finishes the sum over the
// per-tile partials dm_reg_fill_reg left (fixed order: the same bits in every block), scales the four value fields of its valid
// pixels as dm_rescale does and exports the scaled values; block (0, 0) leaves the factor at factor_out for the host.
template <bool RESCALE>
__global__ __launch_bounds__(256) void dm_export_pyramid(DepthSoA s, ExportPyrArgs a, const double* __restrict__ part, int nparts, float* __restrict__ factor_out) {
  __shared__ float ld[2][32 * 32], lv[2][32 * 32];   // ping-pong: level l in [l & 1]
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int W = a.W, H = a.H;
  float f = 1.0f, f2 = 1.0f;
  if (RESCALE) {
    __shared__ double sa[256], sc[256];
    const int t = threadIdx.x;
    double acc = 0.0, cnt = 0.0;
    for (int p = t; p < nparts; p += 256) { acc += part[2 * p]; cnt += part[2 * p + 1]; }
    sa[t] = acc; sc[t] = cnt;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (t < off) { sa[t] += sa[t + off]; sc[t] += sc[t + off]; }
      __syncthreads();
    }
    f = (float)sc[0] / (float)sa[0];   // rescaleFactor = numIdepth / sumIdepth (f32)
    f2 = f * f;
    if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) *factor_out = f;
  }
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int ty = i >> 5, tx = i & 31;
    const int x = bx + tx, y = by + ty;
    float d = 0.0f, v = -1.0f;
    if (x < W && y < H) {
      const int p = x + y * W;
      bool valid = s.isValid[p] != 0;
      float ids = s.invDepthSmoothed[p];
      float vs = 0.0f;
      if (RESCALE && valid) {
        s.invDepth[p] *= f;
        ids *= f;
        s.invDepthSmoothed[p] = ids;
        s.variance[p] *= f2;
        vs = s.varianceSmoothed[p] * f2;
        s.varianceSmoothed[p] = vs;
      }
      if (y < 3 || y >= H - 3 || x < 3 || x >= W - 3) {
        valid = false;
        s.isValid[p] = 0;
      }
      if (valid && ids >= -0.05f) {
        d = 1.0f / ids;
        v = RESCALE ? vs : s.varianceSmoothed[p];
      }
      a.depth[0][p] = d;
      a.var[0][p] = v;
    }
    ld[0][i] = d;
    lv[0][i] = v;
  }
  __syncthreads();
  int edge = 32;
  for (int l = 1; l <= a.steps; l++) {
    const int e2 = edge >> 1;             // tile edge at level l
    const int wl = W >> l, hl = H >> l;
    const float* sd = ld[(l - 1) & 1];
    const float* sv = lv[(l - 1) & 1];
    for (int i = threadIdx.x; i < e2 * e2; i += 256) {
      const int ty = i / e2, tx = i - ty * e2;
      const int x = (bx >> l) + tx, y = (by >> l) + ty;
      const int q0 = (2 * ty) * edge + 2 * tx;
      const float d4[4] = {sd[q0], sd[q0 + 1], sd[q0 + edge], sd[q0 + edge + 1]};
      const float v4[4] = {sv[q0], sv[q0 + 1], sv[q0 + edge], sv[q0 + edge + 1]};
      float od, ov;
      depth_pyr_merge(d4, v4, od, ov);
      if (x < wl && y < hl) {
        a.depth[l][x + y * wl] = od;
        a.var[l][x + y * wl] = ov;
      }
      ld[l & 1][ty * e2 + tx] = od;
      lv[l & 1][ty * e2 + tx] = ov;
    }
    __syncthreads();
    edge = e2;
  }
}

// depthMap::makeInvDepthOne (:1546-1587): sum of invDepthSmoothed over valid pixels and their count.
// Stage 1: per-block f64 partials (fixed order); stage 2: one block combines them with the same fixed tree.
__global__ __launch_bounds__(256) void dm_sum_stage1(DepthSoA s, int n, double* __restrict__ part) {
  double acc = 0.0, cnt = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (s.isValid[i]) { acc += (double)s.invDepthSmoothed[i]; cnt += 1.0; }
  __shared__ double sa[256], sc[256];
  sa[threadIdx.x] = acc; sc[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { sa[threadIdx.x] += sa[threadIdx.x + off]; sc[threadIdx.x] += sc[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = sa[0]; part[2 * blockIdx.x + 1] = sc[0]; }
}
// every block redoes the second stage of the sum (the fixed tree of stage 1 again, over the nblocks <= 256 partials: same bits everywhere) and
// rescales its pixels; block 0 leaves the factor at factor_out for the host
__global__ __launch_bounds__(256) void dm_rescale(DepthSoA s, int n, const double* __restrict__ part, int nblocks, float* __restrict__ factor_out) {
  __shared__ double sa[256], sc[256];
  const int t = threadIdx.x;
  sa[t] = (t < nblocks) ? part[2 * t] : 0.0;
  sc[t] = (t < nblocks) ? part[2 * t + 1] : 0.0;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) { sa[t] += sa[t + off]; sc[t] += sc[t + off]; }
    __syncthreads();
  }
  const float f = (float)sc[0] / (float)sa[0];   // rescaleFactor = numIdepth / sumIdepth (f32)
  if (blockIdx.x == 0 && t == 0) *factor_out = f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !s.isValid[i]) return;
  const float f2 = f * f;
  s.invDepth[i] *= f;
  s.invDepthSmoothed[i] *= f;
  s.variance[i] *= f2;
  s.varianceSmoothed[i] *= f2;
}

// (one atomic per BLOCK of a small grid: an atomic per wave of a pixel-per-lane grid — 4 800 adds to one word at 640x480 —
// serialises at the memory side and took 55 us)
__global__ __launch_bounds__(256) void dm_count_valid(DepthSoA s, int n, int* count) {
  int mine = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) mine += s.isValid[i] ? 1 : 0;
  __shared__ int part[4];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) mine += __shfl_xor(mine, m, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = part[0] + part[1] + part[2] + part[3];
    if (tot) atomicAdd(count, tot);
  }
}

// ------------------------------------------------------------------------------------------------
// depthMap::propagateDepth (:1003-1157). The reference is a serial raster-order scatter whose collisions are
// resolved by read-modify-write on the target. Here: (1) dm_prop_project: every valid source computes its candidate and
// target in parallel and enters the target's bucket; (2) dm_prop_fold: one thread per target applies the reference's
// occlusion / EKF-merge fold to its sources in ascending source index. Targets are independent of one another, so this
// reproduces the serial result exactly.
struct PropArgs {
  DepthSoA src, dst;
  const uint8_t* oldImg;     // old keyframe level-0 image
  const uint8_t* newImg;     // new keyframe level-0 image
  const float* newMaxGrad;   // new keyframe maxAbsGradient
  int W, H, sw;
  float R[9], t[3];          // new <- old (SE3poseThisWrtOther of the new keyframe)
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  int* tgt; float* nid; float* nvar; int* nval;
  int* cnt;      // per target: sources that map to it (zero between calls: dm_prop_fold leaves it so)
  int* slots;    // per target: the first DM_PROP_SLOTS of them, in arrival order
};
#define DM_PROP_SLOTS 4

__global__ void dm_prop_project(PropArgs a) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= a.W || y >= a.H) return;
  const int i = x + y * a.W;
  // wipe the destination map (:1009-1014) and the per-target selection slot
  a.dst.isValid[i] = 0;
  a.dst.blacklisted[i] = 0;
  int target = -1;
  float new_idepth = 0.0f, new_var = 0.0f;
  if (a.src.isValid[i]) {
    const float ids = a.src.invDepthSmoothed[i];
    const float k0 = (float)x * a.fxi + a.cxi, k1 = (float)y * a.fyi + a.cyi, k2 = 1.0f;
    float pn[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float s = (a.R[r * 3 + 0] * k0 + a.R[r * 3 + 1] * k1) + a.R[r * 3 + 2] * k2;
      pn[r] = s / ids + a.t[r];
    }
    new_idepth = 1.0f / pn[2];
    const float u_new = pn[0] * new_idepth * a.fx + a.cx;
    const float v_new = pn[1] * new_idepth * a.fy + a.cy;
    if (u_new > 2.1f && v_new > 2.1f && u_new < (float)a.W - 3.1f && v_new < (float)a.H - 3.1f) {
      const int newIDX = (int)(u_new + 0.5f) + ((int)(v_new + 0.5f)) * a.W;
      const float destAbsGrad = a.newMaxGrad[i];   // read at the SOURCE coordinates (Q14)
      const float sourceColor = (float)a.oldImg[(size_t)y * a.sw + x];
      const float destColor = tap_plain(a.newImg, a.sw, a.W, a.H, u_new, v_new);
      const float residual = destColor - sourceColor;
      const bool drop = (residual * residual / (DM_MAX_DIFF_CONSTANT + DM_MAX_DIFF_GRAD_MULT * destAbsGrad * destAbsGrad) > 1.0f) ||
                        (destAbsGrad < DM_MIN_ABS_GRAD_DECREASE);
      if (!drop) {
        float r4 = new_idepth / ids;
        r4 *= r4;
        r4 *= r4;
        new_var = r4 * a.src.invDepth[i];   // sic: inverse depth, not variance (Q14)
        target = newIDX;
      }
    }
  }
  a.tgt[i] = target;
  a.nid[i] = new_idepth;
  a.nvar[i] = new_var;
  a.nval[i] = a.src.validity[i];
  if (target >= 0) {   // bucket the source with its target (dm_prop_fold)
    const int pos = atomicAdd(&a.cnt[target], 1);
    if (pos < DM_PROP_SLOTS) a.slots[target * DM_PROP_SLOTS + pos] = i;
  }
}

// One source folded into its target's hypothesis (:1090-1148): occlusion check, then create or EKF-merge. The target's fields
// live in registers while its sources are applied one after the other.
struct PropTarget { float id, var; int validity; bool valid, touched; };
__device__ __forceinline__ void prop_fold_one(PropTarget& T, float new_idepth, float new_var, int src_validity) {
  bool tvalid = T.valid;
  if (tvalid) {   // occlusion check (:1090-1107)
    const float diff = T.id - new_idepth;
    if (1.0f * diff * diff > new_var + T.var) {
      if (new_idepth < T.id) return;
      tvalid = false;
    }
  }
  if (!tvalid) {
    T.id = new_idepth;
    T.var = new_var;
    T.validity = src_validity;
  } else {   // EKF merge (:1124-1148)
    const float tvar = T.var, tid = T.id;
    const float w = new_var / (tvar + new_var);
    const float merged = w * tid + (1.0f - w) * new_idepth;
    int mv = src_validity + T.validity;
    if ((float)mv > DM_VALIDITY_COUNTER_MAX + DM_VALIDITY_COUNTER_MAX_VARIABLE) mv = (int)(DM_VALIDITY_COUNTER_MAX + DM_VALIDITY_COUNTER_MAX_VARIABLE);
    T.id = merged;
    T.var = 1.0f / (1.0f / tvar + 1.0f / new_var);
    T.validity = mv;
  }
  T.valid = true;
  T.touched = true;
}

// One thread per TARGET: its sources, sorted by source index — the reference's raster order — are folded one after the other.
// No rounds, no host in the loop (r02: rounds of atomicMin selection with a host check every four); the serial result exactly
// (targets are independent). A target with more than DM_PROP_SLOTS sources (a map shrinking by more than 2 x: never seen)
// finds them by scanning the source list in order — slow for that one thread, still exact.
__global__ __launch_bounds__(256) void dm_prop_fold(PropArgs a, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int c = a.cnt[t];
  if (c == 0) return;
  a.cnt[t] = 0;   // ready for the next propagation
  PropTarget T;
  T.id = 0.0f; T.var = 0.0f; T.validity = 0; T.valid = false; T.touched = false;   // the destination map was wiped by dm_prop_project
  if (c > DM_PROP_SLOTS) {
    for (int i = 0; i < n; i++)
      if (a.tgt[i] == t) prop_fold_one(T, a.nid[i], a.nvar[i], a.nval[i]);
  } else {
  int s[DM_PROP_SLOTS];
#pragma unroll
  for (int k = 0; k < DM_PROP_SLOTS; k++) s[k] = (k < c) ? a.slots[t * DM_PROP_SLOTS + k] : 0x7fffffff;
  // sorting network for four keys (ascending)
#define DM_CSWAP(i, j) { const int lo = min(s[i], s[j]), hi = max(s[i], s[j]); s[i] = lo; s[j] = hi; }
  DM_CSWAP(0, 1) DM_CSWAP(2, 3) DM_CSWAP(0, 2) DM_CSWAP(1, 3) DM_CSWAP(1, 2)
#undef DM_CSWAP
#pragma unroll
  for (int k = 0; k < DM_PROP_SLOTS; k++)
    if (k < c) prop_fold_one(T, a.nid[s[k]], a.nvar[s[k]], a.nval[s[k]]);
  }
  if (T.touched) {
    a.dst.invDepth[t] = T.id;
    a.dst.variance[t] = T.var;
    a.dst.varianceSmoothed[t] = -1.0f;
    a.dst.invDepthSmoothed[t] = -1.0f;
    a.dst.validity[t] = T.validity;
    a.dst.isValid[t] = 1;
    a.dst.blacklisted[t] = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// observe: depthMap::observeDepthRow (:191-263) -> observeDepthCreate (:267-308) / observeDepthUpdate (:888-999)
// with makeAndCheckEPL (:311-384) and doLineStereo (:397-885).
struct ObsArgs {
  DepthSoA s;
  const uint8_t* kfImg;      // keyframe level-0 image
  const uint8_t* curImg;     // current frame level-0 image
  const float* kfMaxGrad;
  int W, H, sw;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  float otw_t[3];            // SE3poseOtherWrtThis_t of the current frame (keyframe w.r.t. current)
  float Kr[9], Kt[3];        // K_SE3poseThisWrtOther_r / _t
  float Rr[9], tt[3];        // SE3poseThisWrtOther_r / _t
  const struct ObsMats* mats;   // dm_observe<true>: the five matrices above come from here (device memory, track_setup_wave)
  const int* gate;              //   and nothing is done unless *gate != 0
  int* list;                    // work list of the two observe kernels: DM_OBS_REGIONS regions of region_cap pixel indices each,
  int region_cap;               //   a region filled with creations from its front and updates from its back
  float2* list_ep;              //   beside every list entry: the epipolar direction makeAndCheckEPL gave its pixel (the walk does not compute it again)
  int* ctr;                     //   [2 r + kind] entries of region r: this call's counters (zero when dm_observe_select starts)
  int* ctr_next;                //   the next call's counters: dm_observe_walk leaves them zero (the two sets alternate from call to call)
};

// The matrices of frame::calculateSE3poseOtherWrtThis (Frame.cpp:376-413) that observeDepthRow uses, for a frame whose
// poseWrtOrigin is `pose` against the keyframe (poseWrtOrigin = 0): [R|t] of Other-w.r.t.-This and This-w.r.t.-Other and K R,
// K t of the latter (f32 products, summed left to right as Eigen's 3x3 f32 product does). Host and device: the tracked-frame
// call builds them on the device from the alignment's result (track_setup_wave, in its finish kernel), every other caller on the host.
struct ObsMats {
  float otw_t[3], Kr[9], Kt[3], Rr[9], tt[3];
};
// (device: exp / log as real function calls — inlined, the setup kernel is 15 000 instructions of straight-line code that one
// lane executes once: it ran for 19 us, most of it instruction fetch)
__host__ __device__ __attribute__((noinline)) void obs_exp_se3_f32(const float* pose, float* S) { exp_se3_f32(pose, S); }
__host__ __device__ __attribute__((noinline)) void obs_log_se3_f32(const float* S, float* pose) { log_se3_f32(S, pose); }
ELLC_HD void build_obs_mats(const float* Kmat, const float* pose, ObsMats& m) {
  float rel[6], otw[12], two[12];
  {
    // concatenateOriginPose(other->poseWrtOrigin = 0, poseWrtOrigin, .) = log(exp(0) exp(pose)^-1): exp(0) is the identity and a
    // product with it is exact in compose_f32, so this is concat_origin_f32(zero, pose, rel) bit for bit, two exps shorter
    float B[12], Bi[12];
    obs_exp_se3_f32(pose, B);
    invert_f32(B, Bi);
    obs_log_se3_f32(Bi, rel);
  }
  obs_exp_se3_f32(rel, otw);
  invert_f32(otw, two);
  for (int r = 0; r < 3; r++) {
    for (int q = 0; q < 3; q++) {
      float sum = 0;
      for (int k = 0; k < 3; k++) sum += Kmat[r * 3 + k] * two[k * 4 + q];
      m.Kr[r * 3 + q] = sum;
      m.Rr[r * 3 + q] = two[r * 4 + q];
    }
    float sum = 0;
    for (int k = 0; k < 3; k++) sum += Kmat[r * 3 + k] * two[k * 4 + 3];
    m.Kt[r] = sum;
    m.tt[r] = two[r * 4 + 3];
    m.otw_t[r] = otw[r * 4 + 3];
  }
}

// Tracked-frame call (ellc_track_frame): the alignment's finish kernel turns the pose it has just computed into poseWrtOrigin and
// the observation's matrices (track_setup_wave below) and sets gate = 1 when the alignment's schedule has ended (a state-driven
// schedule may need a continuation that only the host can start: the depth stages behind it then do nothing and the host runs
// them afterwards).
// exp_se3_f32 by one wave whose lanes all hold the same twist: lane 3 r + k evaluates entry (r, k) — the operations exp_se3
// performs for it (exp_se3_entry) — and every lane receives the twelve results: the same bits as exp_se3_f32 at a ninth of its
// dependent f64 arithmetic per lane
__device__ __forceinline__ void wave_exp_se3_f32(const float* pose, float* S) {
  const int lane = threadIdx.x & 63;
  const int l9 = min(lane, 8), r3 = l9 / 3, k3 = l9 - 3 * r3;
  double Rrk, Vv;
  exp_se3_entry((double)pose[0], (double)pose[1], (double)pose[2], (double)pose[3], (double)pose[4], (double)pose[5], r3, k3, Rrk, Vv);
  const double trow = (Vv + __shfl_down(Vv, 1)) + __shfl_down(Vv, 2);   // t[r] in lanes 0, 3, 6
  const float Rf = (float)Rrk, Tf = (float)trow;
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int k = 0; k < 3; k++) S[r * 4 + k] = __shfl(Rf, 3 * r + k);
    S[r * 4 + 3] = __shfl(Tf, 3 * r);
  }
}
// log as a real call with its operands in REGISTERS (twelve in, six out): through pointers the caller's arrays live in scratch, and a
// kernel that calls it — gn_fca_persist, gn_fused_finish — carried 176 bytes of private memory per lane for this one call (r05 verdict)
struct ObsF12 { float v[12]; };
struct ObsF6 { float v[6]; };
__device__ __attribute__((noinline)) ObsF6 obs_log_se3_regs(ObsF12 S) {
  ObsF6 o;
  log_se3_f32(S.v, o.v);
  return o;
}
__device__ __forceinline__ void obs_log_se3_wave(const float* S, float* pose) {
  ObsF12 a;
#pragma unroll
  for (int i = 0; i < 12; i++) a.v[i] = S[i];
  const ObsF6 o = obs_log_se3_regs(a);
#pragma unroll
  for (int i = 0; i < 6; i++) pose[i] = o.v[i];
}
// build_obs_mats, every lane of the wave evaluating it on the same pose (the three exps through wave_exp_se3_f32)
__device__ __forceinline__ void build_obs_mats_wave(const float* Kmat, const float* pose, ObsMats& m) {
  float rel[6], otw[12], two[12];
  {
    float B[12], Bi[12];
    wave_exp_se3_f32(pose, B);
    invert_f32(B, Bi);
    obs_log_se3_wave(Bi, rel);
  }
  wave_exp_se3_f32(rel, otw);
  invert_f32(otw, two);
  for (int r = 0; r < 3; r++) {
    for (int q = 0; q < 3; q++) {
      float sum = 0;
      for (int k = 0; k < 3; k++) sum += Kmat[r * 3 + k] * two[k * 4 + q];
      m.Kr[r * 3 + q] = sum;
      m.Rr[r * 3 + q] = two[r * 4 + q];
    }
    float sum = 0;
    for (int k = 0; k < 3; k++) sum += Kmat[r * 3 + k] * two[k * 4 + 3];
    m.Kt[r] = sum;
    m.tt[r] = two[r * 4 + 3];
    m.otw_t[r] = otw[r * 4 + 3];
  }
}
// the body of the tracked-frame setup, run by the first wave of gn_fused_finish behind the alignment's last solve: pose ->
// poseWrtOrigin = concatenateRelativePose(pose, 0) (ImageFunc.cpp:305; the keyframe's own poseWrtOrigin is zero) -> the matrices
__device__ void track_setup_wave(const float* pose, const float* Kmat, ObsMats* mats) {
  float pwo[6], E[12];
  wave_exp_se3_f32(pose, E);     // concat_relative_f32(pose, 0, pwo) = log(exp(pose) exp(0)): the product with the identity is exact
  obs_log_se3_wave(E, pwo);
  ObsMats m;
  build_obs_mats_wave(Kmat, pwo, m);
  if ((threadIdx.x & 63) == 0) *mats = m;
}
// number of valid hypotheses, then its copy into host-visible memory (the seeds figure main.cpp writes beside the pose, counted
// BEFORE the frame's observation)
// (a single launch in front of the alignment instead of clear + count + copy: one 16-byte word per thread, the blocks' sums
// meet in acc[0]; the last block to arrive — ticket acc[1] — moves the total to the host-visible word and leaves both zero)
__device__ void dm_count_valid_body(const uint8_t* __restrict__ valid, int n, int* __restrict__ acc, int* __restrict__ host_visible, int block, int nblocks) {
  int mine = 0;
  const int n16 = n >> 4;
  const uint4* v16 = (const uint4*)valid;   // plane buffers are 256-byte aligned
  const int bd = (int)blockDim.x;   // 1024 (dm_count_valid_block, the staging launch) or 256 (riding in gn_fca_persist's launch)
  const int i = block * bd + threadIdx.x;
  if (i < n16) {
    const uint4 w = v16[i];
    const unsigned q[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; k++)   // non-zero bytes of a word
      mine += ((q[k] & 0xffu) != 0) + ((q[k] & 0xff00u) != 0) + ((q[k] & 0xff0000u) != 0) + ((q[k] & 0xff000000u) != 0);
  }
  if (block == 0)
    for (int k = (n16 << 4) + threadIdx.x; k < n; k += bd) mine += valid[k] ? 1 : 0;
  __shared__ int part[16];
  if (threadIdx.x < 16) part[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) mine += __shfl_xor(mine, m, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < 16; w++) tot += part[w];
    atomicAdd(&acc[0], tot);
    __threadfence();
    if (atomicAdd(&acc[1], 1) == nblocks - 1) {
      // the count, then the number of counts this context has made (acc[2]): the host takes the figure when the number is the one
      // it expects (ellc_track_frame) — the counting blocks that ride in the alignment's resident launch (r06) are not ordered
      // against the block that writes the alignment's result, and a host that polls that result could read the previous frame's
      // count (seen once in six runs with two processes on one GPU: test_loop_closure_batch_sharded_over_two_processes)
      const int count = atomicExch(&acc[0], 0);
      acc[1] = 0;
      const int seq = acc[2] + 1;
      acc[2] = seq;
      *(volatile int*)host_visible = count;
      __threadfence_system();
      *(volatile int*)(host_visible + 1) = seq;
    }
  }
}
__global__ __launch_bounds__(1024) void dm_count_valid_block(const uint8_t* __restrict__ valid, int n, int* __restrict__ acc, int* __restrict__ host_visible) {
  dm_count_valid_body(valid, n, acc, host_visible, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ float dot3f(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
// Vector3f::dot of the reference (DepthPropagation.cpp:835-848): Eigen's fixed-size redux sums three terms as e0 + (e1 + e2)
// (Redux.h, redux_novec_unroller: func(first half, second half)) — not the left-to-right sum of a coefficient-based matrix product
// (dot3f). r06; through the cancellation in (dot0 - oldX dot2) / nominator the two orders differ by up to ~1e-6 relative in the depth.
__device__ __forceinline__ float dot3f_redux(const float* a, const float* b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

__device__ inline bool make_and_check_epl(const ObsArgs& a, int x, int y, float& pepx, float& pepy) {
  const float epx = -a.fx * a.otw_t[0] + a.otw_t[2] * ((float)x - a.cx);
  const float epy = -a.fy * a.otw_t[1] + a.otw_t[2] * ((float)y - a.cy);
  const float se = epx + epy;
  if (se != se) return false;
  const float eplLengthSquared = epx * epx + epy * epy;
  if (eplLengthSquared < DM_MIN_EPL_LENGTH_SQUARED) return false;
  const uint8_t* c = a.kfImg + (size_t)y * a.sw;
  const float gx = (float)((int)c[x + 1] - (int)c[x - 1]);
  const float gy = (float)((int)c[x + a.sw] - (int)c[x - a.sw]);
  float eplGradSquared = gx * epx + gy * epy;
  eplGradSquared = eplGradSquared * eplGradSquared / eplLengthSquared;
  if (eplGradSquared < DM_MIN_EPL_GRAD_SQUARED) return false;
  if (eplGradSquared / (gx * gx + gy * gy) < DM_MIN_EPL_ANGLE_SQUARED) return false;
  const float fac = DM_GRADIENT_SAMPLE_DIST / sqrtf(eplLengthSquared);
  pepx = epx * fac;
  pepy = epy * fac;
  return true;
}

// returns the reference's float error code / best error; outputs valid when the return value is >= 0
__device__ inline float do_line_stereo(const ObsArgs& a, float u, float v, float epxn, float epyn, float min_idepth, float prior_idepth,
                                       float max_idepth, float& result_idepth, float& result_var) {
  const int W = a.W, H = a.H;
  const float NaNf = __builtin_nanf("");
  float KinvP[3] = {a.fxi * u + a.cxi, a.fyi * v + a.cyi, 1.0f};
  float pInf[3] = {dot3f(a.Kr, KinvP), dot3f(a.Kr + 3, KinvP), dot3f(a.Kr + 6, KinvP)};
  const float pRealZ = pInf[2] / prior_idepth + a.Kt[2];
  const float rescaleFactor = pRealZ * prior_idepth;
  const float firstX = u - 2 * epxn * rescaleFactor, firstY = v - 2 * epyn * rescaleFactor;
  const float lastX = u + 2 * epxn * rescaleFactor, lastY = v + 2 * epyn * rescaleFactor;
  if (firstX <= 0 || firstX >= (float)(W - 2) || firstY <= 0 || firstY >= (float)(H - 2) || lastX <= 0 || lastX >= (float)(W - 2) ||
      lastY <= 0 || lastY >= (float)(H - 2))
    return -1.0f;
  if (!(rescaleFactor > 0.7f && rescaleFactor < 1.4f)) return -1.0f;

  // (the five keyframe taps: loads issued together, see raw_tap_load)
  const float kx_p1 = u + epxn * rescaleFactor, ky_p1 = v + epyn * rescaleFactor, kx_m1 = u - epxn * rescaleFactor, ky_m1 = v - epyn * rescaleFactor;
  const float kx_m2 = u - 2 * epxn * rescaleFactor, ky_m2 = v - 2 * epyn * rescaleFactor, kx_p2 = u + 2 * epxn * rescaleFactor, ky_p2 = v + 2 * epyn * rescaleFactor;
  const RawTap rk_p1 = raw_tap_load(a.kfImg, a.sw, W, H, kx_p1, ky_p1), rk_m1 = raw_tap_load(a.kfImg, a.sw, W, H, kx_m1, ky_m1);
  const RawTap rk_0 = raw_tap_load(a.kfImg, a.sw, W, H, u, v);
  const RawTap rk_m2 = raw_tap_load(a.kfImg, a.sw, W, H, kx_m2, ky_m2), rk_p2 = raw_tap_load(a.kfImg, a.sw, W, H, kx_p2, ky_p2);
  // ... and the four neighbours of (u, v) the tail's keyframe gradient is made of (u, v are whole pixels, 3 <= x < W - 3: the
  // bilinear gradient tap of Frame.h:283-394 collapses to the central differences of its sample at weight 1, bit for bit);
  // all of them are used behind the geometry below, so that it runs while they are in flight
  const unsigned kc = __umul24((unsigned)(int)v, (unsigned)a.sw) + (unsigned)(int)u;
  const uint8_t kg_l = as_global(a.kfImg)[kc - 1u], kg_r = as_global(a.kfImg)[kc + 1u];
  const uint8_t kg_u = as_global(a.kfImg)[kc - (unsigned)a.sw], kg_d = as_global(a.kfImg)[kc + (unsigned)a.sw];

  float pClose[3] = {pInf[0] + a.Kt[0] * max_idepth, pInf[1] + a.Kt[1] * max_idepth, pInf[2] + a.Kt[2] * max_idepth};
  if (pClose[2] < 0.001f) {
    max_idepth = (0.001f - pInf[2]) / a.Kt[2];
    for (int i = 0; i < 3; i++) pClose[i] = pInf[i] + a.Kt[i] * max_idepth;
  }
  { const float z = pClose[2]; for (int i = 0; i < 3; i++) pClose[i] = pClose[i] / z; }
  float pFar[3] = {pInf[0] + a.Kt[0] * min_idepth, pInf[1] + a.Kt[1] * min_idepth, pInf[2] + a.Kt[2] * min_idepth};
  if (pFar[2] < 0.001f || max_idepth < min_idepth) return -1.0f;
  { const float z = pFar[2]; for (int i = 0; i < 3; i++) pFar[i] = pFar[i] / z; }
  { const float q = pFar[0] + pClose[0]; if (q != q) return -4.0f; }

  float incx = pClose[0] - pFar[0];
  float incy = pClose[1] - pFar[1];
  const float eplLength = sqrtf(incx * incx + incy * incy);
  if (eplLength == 0.0f || isinf(eplLength)) return -4.0f;   // reference: (!eplLength) > 0 || isinf
  if (eplLength > DM_MAX_EPL_LENGTH_CROP) {
    pClose[0] = pFar[0] + incx * DM_MAX_EPL_LENGTH_CROP / eplLength;
    pClose[1] = pFar[1] + incy * DM_MAX_EPL_LENGTH_CROP / eplLength;
  }
  incx *= DM_GRADIENT_SAMPLE_DIST / eplLength;
  incy *= DM_GRADIENT_SAMPLE_DIST / eplLength;
  pFar[0] -= incx; pFar[1] -= incy;
  pClose[0] += incx; pClose[1] += incy;
  if (eplLength < DM_MIN_EPL_LENGTH_CROP) {
    const float pad = (DM_MIN_EPL_LENGTH_CROP - (eplLength)) / 2.0f;
    pFar[0] -= incx * pad; pFar[1] -= incy * pad;
    pClose[0] += incx * pad; pClose[1] += incy * pad;
  }
  const float Bd = DM_SAMPLE_POINT_TO_BORDER;
  const float WB = (float)W - Bd, HB = (float)H - Bd;
  if (pFar[0] <= Bd || pFar[0] >= WB || pFar[1] <= Bd || pFar[1] >= HB) return -1.0f;
  if (pClose[0] <= Bd || pClose[0] >= WB || pClose[1] <= Bd || pClose[1] >= HB) {
    if (pClose[0] <= Bd) {
      const float toAdd = (Bd - pClose[0]) / incx;
      pClose[0] += toAdd * incx; pClose[1] += toAdd * incy;
    } else if (pClose[0] >= WB) {
      const float toAdd = (WB - pClose[0]) / incx;
      pClose[0] += toAdd * incx; pClose[1] += toAdd * incy;
    }
    if (pClose[1] <= Bd) {
      const float toAdd = (Bd - pClose[1]) / incy;
      pClose[0] += toAdd * incx; pClose[1] += toAdd * incy;
    } else if (pClose[1] >= HB) {
      const float toAdd = (HB - pClose[1]) / incy;
      pClose[0] += toAdd * incx; pClose[1] += toAdd * incy;
    }
    const float fincx = pClose[0] - pFar[0];
    const float fincy = pClose[1] - pFar[1];
    const float newEplLength = sqrtf(fincx * fincx + fincy * fincy);
    if (pClose[0] <= Bd || pClose[0] >= WB || pClose[1] <= Bd || pClose[1] >= HB || newEplLength < 8.0f) return -1.0f;
  }

  float cpx = pFar[0], cpy = pFar[1];
  // The walk (DepthPropagation.cpp:612-710), laid out for the wave instead of for one pixel (r04):
  //  * the taps of the next four steps are in flight while a step is evaluated: (lx, ly) runs four steps ahead of (cpx, cpy) through
  //    the walk's own recurrence (the same additions from the same start: the same positions, bit for bit), and the four tap slots are
  //    NAMED — the loop body is four steps, step j consumes slot j and refills it — because a queue that shifts (q[d] = q[d + 1], r03)
  //    compiles to register copies at the loop's back edge, and a copy of a slot waits for the load that has just been issued into
  //    it: r03's look-ahead never was in flight across an iteration (s_waitcnt vmcnt(0) in every step);
  //  * the loop is uniform over the wave: every lane steps until the last lane's segment ends (`alive` guards a lane's bookkeeping;
  //    the refills are unconditional — a finished lane's look-ahead position is clamped into the image by raw_tap_load), so there
  //    is no per-lane exit whose merges would bring the copies back;
  //  * the reference alternates two sets of differences (e1A.. / e1B..) by the parity of its loop counter: with an even number of
  //    steps in the body the parity is a step's position in it, not a run-time test.
  float lx = cpx, ly = cpy;
  const RawTap rc_m2 = raw_tap_load(a.curImg, a.sw, W, H, cpx - 2.0f * incx, cpy - 2.0f * incy);
  const RawTap rc_m1 = raw_tap_load(a.curImg, a.sw, W, H, cpx - incx, cpy - incy);
  const RawTap rc_0 = raw_tap_load(a.curImg, a.sw, W, H, cpx, cpy);
  const RawTap rc_p1 = raw_tap_load(a.curImg, a.sw, W, H, cpx + incx, cpy + incy);
  RawTap q0, q1, q2, q3;   // requested by hand, see raw_tap_issue: every use is behind a raw_tap_wait
  raw_tap_issue(q0, a.curImg, a.sw, W, H, lx + 2 * incx, ly + 2 * incy); lx += incx; ly += incy;
  raw_tap_issue(q1, a.curImg, a.sw, W, H, lx + 2 * incx, ly + 2 * incy); lx += incx; ly += incy;
  raw_tap_issue(q2, a.curImg, a.sw, W, H, lx + 2 * incx, ly + 2 * incy); lx += incx; ly += incy;
  raw_tap_issue(q3, a.curImg, a.sw, W, H, lx + 2 * incx, ly + 2 * incy); lx += incx; ly += incy;
  const float realVal_p1 = tap_plain_raw(a.kfImg, a.sw, W, H, kx_p1, ky_p1, rk_p1);
  const float realVal_m1 = tap_plain_raw(a.kfImg, a.sw, W, H, kx_m1, ky_m1, rk_m1);
  const float realVal = tap_plain_raw(a.kfImg, a.sw, W, H, u, v, rk_0);
  const float realVal_m2 = tap_plain_raw(a.kfImg, a.sw, W, H, kx_m2, ky_m2, rk_m2);
  const float realVal_p2 = tap_plain_raw(a.kfImg, a.sw, W, H, kx_p2, ky_p2, rk_p2);
  // (the walk's first four taps: interior without a test, as the walk's own, see walk_step)
  float val_cp_m2 = tap_interior_raw(cpx - 2.0f * incx, cpy - 2.0f * incy, rc_m2);
  float val_cp_m1 = tap_interior_raw(cpx - incx, cpy - incy, rc_m1);
  float val_cp = tap_interior_raw(cpx, cpy, rc_0);
  float val_cp_p1 = tap_interior_raw(cpx + incx, cpy + incy, rc_p1);

  int loopCounter = 0;   // the same in every lane that is still walking
  float best_match_x = -1, best_match_y = -1;
  float best_match_err = __builtin_inff(), second_best_match_err = __builtin_inff();   // float = 1e50
  float best_match_errPre = NaNf, best_match_errPost = NaNf, best_match_DiffErrPre = NaNf, best_match_DiffErrPost = NaNf;
  bool bestWasLastLoop = false;
  float eeLast = -1;
  float e1A = NaNf, e1B = NaNf, e2A = NaNf, e2B = NaNf, e3A = NaNf, e3B = NaNf, e4A = NaNf, e4B = NaNf, e5A = NaNf, e5B = NaNf;
  int loopCBest = -1, loopCSecond = -1;
  // the walk is bounded: the segment is at most MAX_EPL_LENGTH_CROP + padding long and inside the image
  const int loopCap = W + H;
  bool alive = true;
  // one step of the walk; PAR_A: an even value of the reference's loop counter. Returns false when no lane of the wave walks on.
  auto walk_step = [&](RawTap& slot, const bool PAR_A) -> bool {
    alive = alive && ((((incx < 0) == (cpx > pClose[0]) && (incy < 0) == (cpy > pClose[1])) || loopCounter == 0) && loopCounter < loopCap);
    if (__builtin_amdgcn_ballot_w64(alive) == 0ull) return false;
    float val_cp_p2 = 0.0f;
    raw_tap_wait<6>(slot);   // the slot's two loads are followed by the six of the three younger slots
    // (interior without a test: both ends of the segment lie more than SAMPLE_POINT_TO_BORDER = 7 pixels inside the image, a walking
    // lane's position has not passed the far end by a whole step, and the tap sits two steps ahead of it: more than 4 pixels inside)
    if (alive) val_cp_p2 = tap_interior_raw(cpx + 2 * incx, cpy + 2 * incy, slot);
    raw_tap_issue(slot, a.curImg, a.sw, W, H, lx + 2 * incx, ly + 2 * incy);   // refilled behind its use, for the step four on
    lx += incx; ly += incy;
    if (alive) {
      float ee = 0;
      if (PAR_A) {
        e1A = val_cp_p2 - realVal_p2; ee += e1A * e1A;
        e2A = val_cp_p1 - realVal_p1; ee += e2A * e2A;
        e3A = val_cp - realVal;       ee += e3A * e3A;
        e4A = val_cp_m1 - realVal_m1; ee += e4A * e4A;
        e5A = val_cp_m2 - realVal_m2; ee += e5A * e5A;
      } else {
        e1B = val_cp_p2 - realVal_p2; ee += e1B * e1B;
        e2B = val_cp_p1 - realVal_p1; ee += e2B * e2B;
        e3B = val_cp - realVal;       ee += e3B * e3B;
        e4B = val_cp_m1 - realVal_m1; ee += e4B * e4B;
        e5B = val_cp_m2 - realVal_m2; ee += e5B * e5B;
      }
      if (ee < best_match_err) {
        second_best_match_err = best_match_err;
        loopCSecond = loopCBest;
        best_match_err = ee;
        loopCBest = loopCounter;
        best_match_errPre = eeLast;
        best_match_DiffErrPre = e1A * e1B + e2A * e2B + e3A * e3B + e4A * e4B + e5A * e5B;
        best_match_errPost = -1;
        best_match_DiffErrPost = -1;
        best_match_x = cpx;
        best_match_y = cpy;
        bestWasLastLoop = true;
      } else {
        if (bestWasLastLoop) {
          best_match_errPost = ee;
          best_match_DiffErrPost = e1A * e1B + e2A * e2B + e3A * e3B + e4A * e4B + e5A * e5B;
          bestWasLastLoop = false;
        }
        if (ee < second_best_match_err) {
          second_best_match_err = ee;
          loopCSecond = loopCounter;
        }
      }
      eeLast = ee;
      val_cp_m2 = val_cp_m1; val_cp_m1 = val_cp; val_cp = val_cp_p1; val_cp_p1 = val_cp_p2;
      cpx += incx;
      cpy += incy;
    }
    loopCounter++;
    return true;
  };
  for (;;) {
    if (!walk_step(q0, true)) break;
    if (!walk_step(q1, false)) break;
    if (!walk_step(q2, true)) break;
    if (!walk_step(q3, false)) break;
  }
  raw_tap_wait<0>(q0); raw_tap_wait<0>(q1); raw_tap_wait<0>(q2); raw_tap_wait<0>(q3);   // nothing of the walk is in flight beyond this point
  if (best_match_err > 4.0f * DM_MAX_ERROR_STEREO) return -3.0f;
  {
    int dl = loopCBest - loopCSecond;
    if (dl < 0) dl = -dl;
    if ((float)dl > 1.0f && DM_MIN_DISTANCE_ERROR_STEREO * best_match_err > second_best_match_err) return -2.0f;
  }
  bool didSubpixel = false;
  {
    const float gradPre_pre = -(best_match_errPre - best_match_DiffErrPre);
    const float gradPre_this = +(best_match_err - best_match_DiffErrPre);
    const float gradPost_this = -(best_match_err - best_match_DiffErrPost);
    const float gradPost_post = +(best_match_errPost - best_match_DiffErrPost);
    bool interpPost = false, interpPre = false;
    if (best_match_errPre < 0 || best_match_errPost < 0) {
    } else if ((gradPre_pre < 0) ^ (gradPre_this < 0)) {
      if ((gradPost_post < 0) ^ (gradPost_this < 0)) {
      } else interpPre = true;
    } else if ((gradPost_post < 0) ^ (gradPost_this < 0)) {
      interpPost = true;
    }
    if (interpPre) {
      const float d = gradPre_this / (gradPre_this - gradPre_pre);
      best_match_x -= d * incx;
      best_match_y -= d * incy;
      best_match_err = best_match_err - 2 * d * gradPre_this - (gradPre_pre - gradPre_this) * d * d;
      didSubpixel = true;
    } else if (interpPost) {
      const float d = gradPost_this / (gradPost_this - gradPost_post);
      best_match_x += d * incx;
      best_match_y += d * incy;
      best_match_err = best_match_err + 2 * d * gradPost_this + (gradPost_post - gradPost_this) * d * d;
      didSubpixel = true;
    }
  }
  const float sampleDist = DM_GRADIENT_SAMPLE_DIST * rescaleFactor;
  float gradAlongLine = 0;
  float tmp = realVal_p2 - realVal_p1; gradAlongLine += tmp * tmp;
  tmp = realVal_p1 - realVal; gradAlongLine += tmp * tmp;
  tmp = realVal - realVal_m1; gradAlongLine += tmp * tmp;
  tmp = realVal_m1 - realVal_m2; gradAlongLine += tmp * tmp;
  gradAlongLine /= sampleDist * sampleDist;
  if (best_match_err > DM_MAX_ERROR_STEREO + sqrtf(gradAlongLine) * 20) return -3.0f;

  float idnew_best_match, alpha;
  if (incx * incx > incy * incy) {
    const float oldX = a.fxi * best_match_x + a.cxi;
    const float nominator = (oldX * a.tt[2] - a.tt[0]);
    const float dot0 = dot3f_redux(KinvP, a.Rr);
    const float dot2 = dot3f_redux(KinvP, a.Rr + 6);
    idnew_best_match = (dot0 - oldX * dot2) / nominator;
    alpha = incx * a.fxi * (dot0 * a.tt[2] - dot2 * a.tt[0]) / (nominator * nominator);
  } else {
    const float oldY = a.fyi * best_match_y + a.cyi;
    const float nominator = (oldY * a.tt[2] - a.tt[1]);
    const float dot1 = dot3f_redux(KinvP, a.Rr + 3);
    const float dot2 = dot3f_redux(KinvP, a.Rr + 6);
    idnew_best_match = (dot1 - oldY * dot2) / nominator;
    alpha = incy * a.fxi * (dot1 * a.tt[2] - dot2 * a.tt[1]) / (nominator * nominator);   // FX_INV, as the reference (Q19)
  }
  if (idnew_best_match < 0) return -2.0f;
  const float photoDispError = 4.0f * (float)DM_CAMERA_PIXEL_NOISE / (gradAlongLine + DM_DIVISION_EPS);
  const float trackingErrorFac = 0.25f * 1.0f;
  const float g0 = 0.5f * ((float)kg_r - (float)kg_l), g1 = 0.5f * ((float)kg_d - (float)kg_u);   // keyframe gradient at (u, v), see above
  float geoDispError = (g0 * epxn + g1 * epyn) + DM_DIVISION_EPS;
  geoDispError = trackingErrorFac * trackingErrorFac * (g0 * g0 + g1 * g1) / (geoDispError * geoDispError);
  result_var = alpha * alpha * ((didSubpixel ? 0.05f : 0.5f) * sampleDist * sampleDist + geoDispError + photoDispError);
  result_idepth = idnew_best_match;
  return best_match_err;
}

// observeDepthRow for one pixel whose epipolar direction (epx, epy) passed makeAndCheckEPL: observeDepthCreate (:267-308) or
// observeDepthUpdate (:888-999). One call site of do_line_stereo for both — its search range is what differs — so that a wave
// holding creations and updates runs the 2 300 instructions around the walk once, not once per kind.
__device__ __forceinline__ void observe_pixel(const ObsArgs& a, int x, int y, int idx, float epx, float epy) {
  Hyp t = hyp_load(a.s, idx);
  const float mg = a.kfMaxGrad[idx];
  const bool create = !t.valid;
  float min_idepth = 0.0f, prior_idepth = 1.0f, max_idepth = 1.0f / DM_MIN_DEPTH;
  if (!create) {
    const float sv = sqrtf(t.vars);
    min_idepth = t.ids - sv * DM_STEREO_EPL_VAR_FAC;
    max_idepth = t.ids + sv * DM_STEREO_EPL_VAR_FAC;
    if (min_idepth < 0) min_idepth = 0;
    if (max_idepth > 1 / DM_MIN_DEPTH) max_idepth = 1 / DM_MIN_DEPTH;
    prior_idepth = t.ids;
  }
  float rid = 0.0f, rvar = 0.0f;
  const float error = do_line_stereo(a, (float)x, (float)y, epx, epy, min_idepth, prior_idepth, max_idepth, rid, rvar);
  if (create) {
    // observeDepthCreate
    if (error == -3.0f || error == -2.0f) t.bl--;
    if (error < 0 || rvar > DM_MAX_VAR) {
      a.s.blacklisted[idx] = t.bl;
      return;
    }
    t.id = unzero_f(rid);
    t.var = rvar;
    t.ids = -1.0f;
    t.vars = -1.0f;
    t.validity = DM_VALIDITY_COUNTER_INITIAL_OBSERVE;
    t.valid = true;
    t.bl = 0;
    hyp_store(a.s, idx, t);
  } else {
    // observeDepthUpdate
    const float diff = rid - t.ids;
    if (error == -1.0f) return;
    if (error == -2.0f) {
      t.validity = (int)((float)t.validity - DM_VALIDITY_COUNTER_DEC);
      if (t.validity < 0) t.validity = 0;
      t.var *= DM_FAIL_VAR_INC_FAC;
      if (t.var > DM_MAX_VAR) {
        t.valid = false;
        t.bl--;
      }
      hyp_store(a.s, idx, t);
      return;
    }
    if (error == -3.0f || error == -4.0f) return;
    if (1.0f * diff * diff > rvar + t.vars) {
      t.var *= DM_FAIL_VAR_INC_FAC;
      if (t.var > DM_MAX_VAR) t.valid = false;
      hyp_store(a.s, idx, t);
      return;
    }
    float id_var = t.var * DM_SUCC_VAR_INC_FAC;
    const float w = rvar / (rvar + id_var);
    const float new_idepth = (1 - w) * rid + w * t.id;
    t.id = unzero_f(new_idepth);
    id_var = id_var * w;
    if (id_var < t.var) t.var = id_var;
    t.validity = (int)((float)t.validity + DM_VALIDITY_COUNTER_INC);
    const float cap = DM_VALIDITY_COUNTER_MAX + mg * (DM_VALIDITY_COUNTER_MAX_VARIABLE) / 255.0f;
    if ((float)t.validity > cap) t.validity = (int)cap;
    hyp_store(a.s, idx, t);
  }
}

// the matrices of a tracked-frame call come from device memory (track_setup_wave)
__device__ __forceinline__ void obs_load_mats(ObsArgs& b) {
  const ObsMats& m = *b.mats;
#pragma unroll
  for (int i = 0; i < 3; i++) { b.otw_t[i] = m.otw_t[i]; b.Kt[i] = m.Kt[i]; b.tt[i] = m.tt[i]; }
#pragma unroll
  for (int i = 0; i < 9; i++) { b.Kr[i] = m.Kr[i]; b.Rr[i] = m.Rr[i]; }
}

// observeDepthRow in two launches. dm_observe_select, one lane per pixel: the cheap tests (:191-263) and makeAndCheckEPL; the
// pixels that go on to the line stereo are appended to a device-wide work list — creations (no hypothesis: the search spans
// the whole inverse-depth range, a walk of ~30 steps) and updates (a few steps) apart. dm_observe_walk: lane k takes entry k of
// all creations followed by all updates, so every wave is full and, but for the one that straddles the boundary, holds walks
// of one kind. (One launch with a pixel per lane ran every wave as long as its longest walk with a third of its lanes active:
// 19.7 M wave instructions, 47 us in r02; compacted and sorted per 32 x 8 tile, r03, still 12 000 instructions per tile in two
// waves, one of them a tenth full: 42 us.) The pixels are independent, so their order does not matter: same results.
//   The list is DM_OBS_REGIONS regions, block b appending to region b mod DM_OBS_REGIONS with one atomic per kind —
// creations from the region's front, updates from its back (a region holds every pixel of its blocks: the ends never meet).
// (One list with one pair of counters and an atomic per wave: 9 600 atomics on two words serialise, the select launch took 98 us;
// 64 regions: 24 us; with the waves' counts combined in LDS, one atomic per block and kind.) A walk block scans
// the 2 x 64 counters in LDS and finds an entry's region by bisection. Two sets of counters alternate from call to call: a call's
// set is zero when its select launch starts, because the walk launch of the call before cleared it (ObsArgs::ctr_next). DEV: the
// tracked-frame call — nothing else is done while the gate is closed.
template <bool DEV>
__global__ __launch_bounds__(256) void dm_observe_select(ObsArgs a, RideWeights rw) {
  if (DEV) {
    const int tiles_y = (a.H + 7) / 8;
    if ((int)blockIdx.y >= tiles_y) {   // (block-uniform) the tracking call's saved weights, riding along (RideWeights): whatever the gate says
      const int r = ((int)blockIdx.y - tiles_y) * (int)gridDim.x + (int)blockIdx.x;
      if (r < rw.n)
        saved_weights_all_body(rw.kf_tab, rw.kf_slot, rw.geom, rw.state, rw.max_kf, rw.fast_records, 0, r / rw.per_level, r % rw.per_level, rw.per_level,
                               (int)threadIdx.x, 256);
      return;
    }
    if (*a.gate == 0) return;
    obs_load_mats(a);
  }
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  const int lane = threadIdx.x & 63;
  const int idx = x + y * a.W;
  int kind = -1;   // 0: creation, 1: update
  float epx = 0.0f, epy = 0.0f;
  if (x >= 3 && x < a.W - 3 && y >= 3 && y < a.H - 3) {
    const bool hasHypothesis = a.s.isValid[idx] != 0;
    const float mg = a.kfMaxGrad[idx];
    if (hasHypothesis && mg < DM_MIN_ABS_GRAD_DECREASE) {
      a.s.isValid[idx] = 0;
    } else if (!(mg < DM_MIN_ABS_GRAD_CREATE || a.s.blacklisted[idx] < DM_MIN_BLACKLIST)) {
      if (make_and_check_epl(a, x, y, epx, epy)) kind = hasHypothesis ? 1 : 0;   // (else create: -1 / update: -5, no state change)
    }
  }
  // one atomic per block and kind: the waves' counts meet in LDS
  __shared__ int cnt[2][4], base[2];
  const int wave = threadIdx.x >> 6;
  unsigned long long mine = 0ull;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const unsigned long long m = __ballot(kind == k);
    if (lane == 0) cnt[k][wave] = __popcll(m);
    if (kind == k) mine = m;
  }
  __syncthreads();
  const int region = (int)((blockIdx.y * gridDim.x + blockIdx.x) % DM_OBS_REGIONS);
  if (threadIdx.x < 2) {
    const int k = threadIdx.x, total = cnt[k][0] + cnt[k][1] + cnt[k][2] + cnt[k][3];
    base[k] = total > 0 ? atomicAdd(&a.ctr[2 * region + k], total) : 0;
  }
  __syncthreads();
  if (kind >= 0) {
    int pos = base[kind] + __popcll(mine & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    for (int w = 0; w < wave; w++) pos += cnt[kind][w];
    if (pos < a.region_cap) {
      const size_t e = (size_t)region * a.region_cap + (size_t)(kind == 0 ? pos : a.region_cap - 1 - pos);
      a.list[e] = x | (y << 16);               // (W, H < 65536)
      a.list_ep[e] = make_float2(epx, epy);
    }
  }
}

template <bool DEV>
__global__ __launch_bounds__(256) void dm_observe_walk(ObsArgs a) {
  // the NEXT call's counters (nobody reads or adds to them during this launch) — also when the gate is closed: the host alternates
  // the two sets whether or not a call does anything. (r03: the last block to finish cleared the one set, found by a counter every
  // block incremented: 1 200 atomics on one word per call.)
  if (blockIdx.x == 0 && threadIdx.x < 2 * DM_OBS_REGIONS) a.ctr_next[threadIdx.x] = 0;
  if (DEV) {
    if (*a.gate == 0) return;
    obs_load_mats(a);
  }
  // exclusive prefixes of the regions' creation / update counts (wave 0 and wave 1 scan one kind each)
  __shared__ int pre[2][DM_OBS_REGIONS + 1];
  {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w < 2) {
      static_assert(DM_OBS_REGIONS == 64, "one lane per region");
      const int c = min(max(a.ctr[2 * lane + w], 0), a.region_cap);
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
      }
      pre[w][lane + 1] = incl;
      if (lane == 0) pre[w][0] = 0;
    }
  }
  __syncthreads();
  const int C = pre[0][DM_OBS_REGIONS], U = pre[1][DM_OBS_REGIONS];
  // A wave takes one chunk of 64 entries of ONE kind; the creation chunks first. (r04 measured the update chunks spread evenly among
  // the creation chunks, so that every CU gets the same mix: 33.7 against 32.2 us — the launch is bound by its instructions, not by
  // which SIMD draws which waves.)
  const int nC = (C + 63) >> 6, nU = (U + 63) >> 6, nchunks = nC + nU;
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const int j = (int)blockIdx.x * 4 + wave;
  if (j < nchunks) {
    const int kind = (j < nC) ? 0 : 1;
    const int kk = (kind ? j - nC : j) * 64 + lane;
    if (kk < (kind ? U : C)) {
      int lo = 0;   // the region r with pre[r] <= kk < pre[r + 1]
#pragma unroll
      for (int step = DM_OBS_REGIONS / 2; step >= 1; step >>= 1)
        if (pre[kind][lo + step] <= kk) lo += step;
      const int off = kk - pre[kind][lo];
      const size_t e = (size_t)lo * a.region_cap + (size_t)(kind ? a.region_cap - 1 - off : off);
      const int xy = a.list[e];
      const float2 ep = a.list_ep[e];
      const int x = xy & 0xffff, y = xy >> 16;
      observe_pixel(a, x, y, x + y * a.W, ep.x, ep.y);
    }
  }
}

}  // namespace ellc
