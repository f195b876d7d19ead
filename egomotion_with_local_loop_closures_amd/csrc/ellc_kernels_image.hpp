// Image-side kernels: u8 pyramid (cv::pyrDown restated), gradient planes, max-gradient map, depth/variance
// pyramid, and the per-level compaction of the keyframe's valid (depth > 0) pixels.
#pragma once
#include "ellc_device.hpp"

namespace ellc {

#ifndef ELLC_GLOBAL
#define ELLC_GLOBAL __attribute__((address_space(1)))
#endif
template <class T>
__device__ __forceinline__ const ELLC_GLOBAL T* gptr(const T* p) { return (const ELLC_GLOBAL T*)p; }
template <class T>
__device__ __forceinline__ ELLC_GLOBAL T* gptr_rw(T* p) { return (ELLC_GLOBAL T*)p; }

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
  return p;
}

// frame::constructImagePyramids (Frame.cpp:170-182) — cv::pyrDown on CV_8UC1: separable [1 4 6 4 1],
// BORDER_REFLECT_101, integer accumulation, (sum + 128) >> 8, dst = ((w+1)/2, (h+1)/2).
__global__ void pyr_down_u8(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst, int dw, int dh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dw || y >= dh) return;
  const int wk[5] = {1, 4, 6, 4, 1};
  int cxs[5];
#pragma unroll
  for (int k = 0; k < 5; k++) cxs[k] = reflect101(2 * x + k - 2, sw);
  int v = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const uint8_t* r = src + (size_t)reflect101(2 * y + j - 2, sh) * sw;
    int h = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) h += wk[k] * (int)r[cxs[k]];
    v += wk[j] * h;
  }
  dst[(size_t)y * dw + x] = (uint8_t)((v + 128) >> 8);
}

// frame::calculateGradient (Frame.cpp:185-285) at one level, planes rows x cols
__device__ __forceinline__ void grad_at(const uint8_t* __restrict__ img, int sw, int cols, int rows, int x, int y, float& gx, float& gy) {
  const int xm = x > 0 ? x - 1 : 0, xp = x < cols - 1 ? x + 1 : cols - 1;
  const int ym = y > 0 ? y - 1 : 0, yp = y < rows - 1 ? y + 1 : rows - 1;
  const float sx = (x == 0 || x == cols - 1) ? 1.0f : 0.5f;
  const float sy = (y == 0 || y == rows - 1) ? 1.0f : 0.5f;
  const float dx = (float)img[(size_t)y * sw + xp] - (float)img[(size_t)y * sw + xm];
  const float dy = (float)img[(size_t)yp * sw + x] - (float)img[(size_t)ym * sw + x];
  gx = (x == 0 || x == cols - 1) ? dx : 0.5f * dx;
  gy = (y == 0 || y == rows - 1) ? dy : 0.5f * dy;
  (void)sx; (void)sy;
}

__global__ void gradient_planes(const uint8_t* __restrict__ img, int sw, int cols, int rows, float* __restrict__ gx, float* __restrict__ gy) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  float a, b;
  grad_at(img, sw, cols, rows, x, y, a, b);
  gx[(size_t)y * cols + x] = a;
  gy[(size_t)y * cols + x] = b;
}

// frame::buildMaxGradients (Frame.cpp:618-674), three passes
__global__ void maxgrad_magnitude(const uint8_t* __restrict__ img, int sw, int w, int h, float* __restrict__ mag) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  float gx, gy;
  grad_at(img, sw, w, h, x, y, gx, gy);
  const float a = gx * gx, b = gy * gy;
  mag[(size_t)y * w + x] = sqrtf(a + b);
}
__global__ void maxgrad_vertical(const float* __restrict__ mag, int w, int h, float* __restrict__ tmp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  float v = 0.0f;
  if (y >= 1 && y < h - 1) {
    const float g1 = fmaxf(mag[(size_t)y * w + x], mag[(size_t)(y - 1) * w + x]);
    v = fmaxf(g1, mag[(size_t)(y + 1) * w + x]);
  }
  tmp[(size_t)y * w + x] = v;
}
__global__ __launch_bounds__(256) void maxgrad_horizontal(const float* __restrict__ mag, const float* __restrict__ tmp, int w, int h, float* __restrict__ out, int* count) {
  int hits = 0;
  const int n = w * h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int y = i / w, x = i - y * w;
    float v = mag[i];   // border pixels keep the raw magnitude
    if (y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {
      const float g1 = fmaxf(tmp[i - 1], tmp[i]);
      v = fmaxf(g1, tmp[i + 1]);
      if (v >= 5.0f) hits++;   // MIN_ABS_GRAD_DECREASE
    }
    out[i] = v;
  }
  __shared__ int sh[256];
  sh[threadIdx.x] = hits;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0 && sh[0]) atomicAdd(count, sh[0]);
}

// depthMap::buildInvVarDepth, one level (DepthPropagation.cpp:1637-1719); the reference's source stride
// is 2*width of the destination. src_depth_is_mat: level-0 source holds keyFrame->depth (0 = invalid).
__global__ void depth_pyr_level(const float* __restrict__ sd, const float* __restrict__ sv, float* __restrict__ dd, float* __restrict__ dv,
                                int width, int height) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= width || y >= height) return;
  const int sw = 2 * width;
  const int idx = 2 * (x + y * sw);
  const int offs[4] = {0, 1, sw, sw + 1};
  float idepthSumsSum = 0.0f, ivarSumsSum = 0.0f;
  int num = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float var = sv[idx + offs[q]];
    if (var > 0.0f) {
      const float ivar = 1.0f / var;
      ivarSumsSum += ivar;
      idepthSumsSum += ivar * 1.0f / sd[idx + offs[q]];
      num++;
    }
  }
  const int o = x + y * width;
  if (num > 0) {
    dd[o] = ivarSumsSum / idepthSumsSum;
    dv[o] = (float)num / ivarSumsSum;
  } else {
    dd[o] = 0.0f;
    dv[o] = -1.0f;
  }
}

// ---------------------------------------------------------------------------------------------------
// Compaction of the keyframe's valid pixels (mask = depth_pyramid[l] > 0, Frame.cpp:295-301), raster
// order preserved. Three launches cover all levels of all listed keyframe slots.
#define ELLC_TILE 2048          // pixels per block: 256 threads x 8 consecutive pixels (two float4 loads)

struct PrepArgs {
  const LevelGeom* geom;
  const KfLevelDev* kf_tab;
  const int* slots;            // unique keyframe slots
  int levels, max_kf;
  int need;                    // bit 0: planes Z / I / saved weight for the ICA path; bit 1: FcaRec records for the FCA path
  int tile_begin[ELLC_MAX_LEVELS + 1];   // prefix of tiles per level
  int tile0, level0;           // this launch covers tiles tile0 + blockIdx.x (count / scatter), levels level0 + blockIdx.x (scan)
};

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
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
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
  if (threadIdx.x == 0) K.tile_count[local] = ws[0] + ws[1] + ws[2] + ws[3];
}

// one block per (level, slot): exclusive scan of the tile counts in place, total -> count
__global__ __launch_bounds__(256) void prep_scan(PrepArgs a) {
  const int level = a.level0 + (int)blockIdx.x;
  const KfLevelDev& K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const int T = a.tile_begin[level + 1] - a.tile_begin[level];
  const int per = (T + 255) / 256;
  const int t0 = threadIdx.x * per;
  int s = 0;
  for (int i = t0; i < min(T, t0 + per); i++) s += K.tile_count[i];
  __shared__ int ws[4];
  int tot;
  const int inc = wave_inclusive_scan(s, tot);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = tot;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wbase += ws[w];
  int run = wbase + inc - s;   // exclusive prefix of this thread's span
  for (int i = t0; i < min(T, t0 + per); i++) {
    const int c = K.tile_count[i];
    K.tile_count[i] = run;
    run += c;
  }
  if (threadIdx.x == 255) *K.count = wbase + inc;
}

// Scatter, two phases per tile of ELLC_TILE pixels. Phase 1: thread t owns pixels base + j*256 + t (j = 0..7), so the
// depth loads of a wave are contiguous; ballot ranks give every valid pixel its raster-order rank inside the tile
// (order = (j, wave, lane)), and (pixel index, depth) are parked in LDS at that rank. Phase 2 runs densely over the
// parked entries — every lane has a valid pixel — computes the record (three IEEE divisions) and stores it; consecutive
// lanes write consecutive records. Without the LDS step the divisions would run for every wave that holds at least one
// valid pixel, i.e. about four times as often on a semi-dense map.
__global__ __launch_bounds__(256) void prep_scatter(PrepArgs a) {
  int local;
  const int level = prep_level_of(a, a.tile0 + (int)blockIdx.x, local);
  const KfLevelDev K = a.kf_tab[level * a.max_kf + a.slots[blockIdx.y]];
  const LevelGeom& g = a.geom[level];
  const int n = g.n;
  const int base = local * ELLC_TILE + (int)threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int cnt[33];   // [j][wave] exclusive offsets, [32] = tile total
  __shared__ uint32_t s_idx[ELLC_TILE];
  __shared__ float s_Z[ELLC_TILE];
  float d[8];
  unsigned long long m[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = base + j * 256;
    d[j] = (i < n) ? gptr(K.depth)[(unsigned)i] : 0.0f;
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
  const unsigned tile_off = (unsigned)K.tile_count[local];
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
  const int cols = g.cols, sw = g.sw, need = a.need;
  const float fx = g.fx, fy = g.fy, cx = g.cx, cy = g.cy;
  for (int r = (int)threadIdx.x; r < nvalid; r += 256) {
    const int i = (int)s_idx[r];
    const float Z = s_Z[r];
    const unsigned pos = tile_off + (unsigned)r;
    int y = (int)(((float)i + 0.5f) * inv_cols);   // i < 2^24: exact conversion; corrected below
    if (y * cols > i) y--;
    if ((y + 1) * cols <= i) y++;
    const int x = i - y * cols;
    const uint32_t xy = ((uint32_t)y << 16) | (uint32_t)x;
    const float Ikf = (float)img[(unsigned)(y * sw + x)];
    cxy[pos] = xy;
    if (need & 1) {   // ICA reads planes
      cZ[pos] = Z;
      cI[pos] = Ikf;
      cW[pos] = wgt[(unsigned)i];
    }
    if (need & 2) {   // FCA reads one 32-byte record per pixel (FcaRec), stored as two 16-byte words
      const float X = (((float)x - cx) * Z) / fx;
      const float Y = (((float)y - cy) * Z) / fy;
      const double invZ = 1.0 / (double)Z;
      const unsigned long long zb = __builtin_bit_cast(unsigned long long, invZ);
      const u32x4 lo = {xy, __builtin_bit_cast(uint32_t, Z), __builtin_bit_cast(uint32_t, var[(unsigned)i]), __builtin_bit_cast(uint32_t, Ikf)};
      const u32x4 hi = {__builtin_bit_cast(uint32_t, X), __builtin_bit_cast(uint32_t, Y), (uint32_t)zb, (uint32_t)(zb >> 32)};
      crec[2u * pos] = lo;
      crec[2u * pos + 1u] = hi;
    }
  }
}

// globalOptimize::calculateImageHistogram (GlobalOptimize.cpp:68): cv::calcHist, 256 uniform bins over [0,256).
// Integer counts (LDS-privatised, then one integer atomic per bin and block) => order-independent, deterministic.
__global__ __launch_bounds__(256) void hist256_u8(const uint8_t* __restrict__ img, int sw, int cols, int rows, unsigned* __restrict__ bins) {
  __shared__ unsigned sh[256];
  sh[threadIdx.x] = 0;
  __syncthreads();
  const int n = cols * rows;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int y = i / cols, x = i - y * cols;
    atomicAdd(&sh[img[(size_t)y * sw + x]], 1u);
  }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&bins[threadIdx.x], sh[threadIdx.x]);
}

// frame::finaliseWeights (Frame.cpp:678-695): weight_pyramid[l] /= numWeightsAdded[l] — cv evaluates
// Mat / int as a*(1/n) through convertTo (32f -> 32f, f32 work type): v * (float)(1.0/n) + 0.
__global__ void scale_plane(float* p, int n, float s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * s + 0.0f;
}

}  // namespace ellc
