// Device-side data layout shared by the HIP kernels and the context (host) code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ELLC_MAX_LEVELS 8
#define ELLC_PART_STRIDE 32     // floats per block partial record (27 used: 21 upper-triangular H + 6 b)
#define ELLC_NBLK_MAX 256       // max accumulate-blocks per alignment
#define ELLC_GN_THREADS 256

namespace ellc {

// Geometry of one pyramid level. cols/rows are the sizes the reference iterates (height/2^l), sw/sh the
// stored image sizes (pyrDown rounds up; they differ for odd sizes — Frame.cpp:110-117 vs :175-179).
struct LevelGeom {
  int cols, rows, sw, sh;
  int n;                      // cols*rows
  float fx, fy, cx, cy;       // GetIntrinsic(level), UserDefinedFunc.cpp:34-50
  float rfx, rfy;             // RN(1/fx), RN(1/fy): for the exact division-by-constant sequence (div_const)
  int divc_ok;                // 1 when div_const was verified exhaustively for this fx, fy (all 2^23 mantissas)
  // per-level Jacobian tables (values that depend on the column or the row only; double where the
  // reference's pow() promotes the sub-expression to double, PixelWisePyramid.cpp:296-303)
  const double* colA;         // fx + u^2/fx              [cols]
  const float* colB;          // (fy*u)/fx                [cols]
  const double* rowA;         // -(fy + v^2/fy)           [rows]
  const float* rowB;          // -((fx*v)/fy)             [rows]
};

// One valid (depth > 0) keyframe pixel as the exact FCA pixel pass reads it: 20 bytes, written by prep_scatter (r04; 32 bytes
// before, and the dense 1280x960 launch moved 2.09 x its algorithmic bytes at the box's HBM ceiling). Position and keyframe
// intensity share one word (x: bits 0-11, y: bits 12-23, intensity: bits 24-31: width, height <= 4096); Z, the variance, and
// invZ = 1.0 / (double)Z, the reference's pow(depth, -1) (:296) — an IEEE f64 division computed once per keyframe update instead of
// once per pixel and iteration. The back-projection X = ((x - cx) Z) / fx, Y likewise (PixelWisePyramid.cpp:236-240) is formed
// again per pixel: with the level's verified division-by-constant (div_const) it costs three instructions a coordinate and gives
// the correctly rounded quotient, the bits `/` gives.
struct __attribute__((packed, aligned(4))) FcaRec {
  uint32_t xyI;
  float Z, var;
  uint32_t invZ_lo, invZ_hi;   // the f64 as two words (the record is 4-byte aligned)
};

// The same pixel as the tolerance-mode FCA pass reads it (cfg.arith = ELLC_ARITH_FAST): **12 bytes** (r05; 16 in r02-r04, 32 before):
// {x | y << 12 | I << 24, variance, d = 1 / Z}. Z itself is not needed — the pass warps (p, q, 1) + t d, the point divided by Z (see
// fcaf_pixel) — and p = (x - cx) / fx, q = (y - cy) / fy come from the integer fields with a conversion and one fma each
// (width, height <= 4096). r04 stored p and y as floats to save those six instructions per pixel; r05 found the pipeline of three
// streams ~80 % HBM-bound in aggregate — a launch group reads its records 32 times — and the quarter fewer bytes worth far more than
// the instructions: batch pipeline 0.1323 -> 0.1251 ms per step, level-0 launch 48.1 -> 43.8 us (interleaved A/B on one box).
struct __attribute__((packed, aligned(4))) FcaRecF {
  uint32_t xyI;
  float var, d;
};

// One valid keyframe pixel as the constant-weight (ICA) pixel pass reads it: 48 bytes. Everything here is independent
// of the pose and of the current frame: back-projection, keyframe intensity, the saved weight and the template-gradient
// steepest-descent row (PixelWisePyramid.cpp:561-680), so the compaction writes it once per ellc_align.
struct __attribute__((aligned(16))) IcaRec {
  float X, Y, Z, Ikf;
  float W, sd0, sd1, sd2;
  float sd3, sd4, sd5, pad;
};

// One keyframe (template) slot at one level: dense planes + the compacted list of pixels with depth > 0
// (frame::calculateNonZeroDepthPts, Frame.cpp:295-301) in raster order.
struct KfLevelDev {
  uint8_t* img;               // sw*sh
  float* depth;               // n   (frame::depth_pyramid[l])
  float* var;                 // n   (depthMap::depthvararrptr[l])
  float* weight;              // n   (frame::weight_pyramid[l])
  uint32_t* cxy;              // compact: y<<16 | x
  float* cZ;                  // compact depth (ICA)
  float* cI;                  // compact keyframe intensity as f32 (ICA)
  FcaRec* crec;               // compact FCA records (same order as cxy)
  IcaRec* irec;               // compact ICA records (same order as cxy)
  float* hpart;               // ICA: per-tile partial sums of H = sum W J^T J, 32 floats per tile (21 used)
  float* hinv;                // ICA: inverse of the level's H (36 floats), one per keyframe slot and level
  float* cW;                  // compact saved weight (ICA)
  float* wlast;               // compact weight of the most recent iteration (for saveWeights)
  float* sd;                  // ICA steepest-descent planes, 6 x cap (plane k at sd + k*cap)
  int* count;                 // V = number of compact entries
  int* tile_count;            // per-tile (ELLC_TILE pixels) counts, then exclusive offsets
  float* idepth;              // n: v_rcp_f32 of `depth` where depth > 0, else 0 — kept for slots with the dense hint only (gn_fca_dense4 reads it in place of `depth`)
  double* invz;               // n: 1.0 / (double)depth, the exact record's pow(depth, -1) — contexts of the exact mode only, slots with the dense hint only (gn_fca_dense_x)
};

struct FrLevelDev {
  uint8_t* img;               // sw*sh
};

// Per-alignment state that persists across the launches of one ellc_align.
#define DM_OBS_REGIONS 64   // regions of the depth map's observation work list (dm_observe_select / dm_observe_walk)

struct AlignState {
  float pose[6];
  float S[12];                // exp(pose^) rounded to f32: r11 r12 r13 t1 | r21.. t2 | r31.. t3
  float delta[6];
  float weighted;
  int level_done;             // level terminated by weightedPose < 1 (-1: none)
  int pending;                // fused schedule: the previous launch left partial sums that are not solved yet
  int cur_level;              // state-driven schedule: the level of the pending sums / of the next pixel pass; -1: the schedule has ended
  int it_in_level;            //   iterations of cur_level already solved
  int iters[ELLC_MAX_LEVELS];
  float H[36];
  float b[6];
  float Hinv[36];
};

// What ellc_align returns per alignment; lives in pinned host memory and is written by the last kernel of a schedule
// (zero-copy), so fetching a result is a stream synchronisation, not a device-to-host copy.
struct AlignResult {
  float pose[6];
  float weighted;
  int iters[ELLC_MAX_LEVELS];
  int pad;
};

}  // namespace ellc
