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

// The whole u8 pyramid below one source level in ONE launch (up to three pyrDown steps): a block owns a PT x PT tile of the
// deepest level it produces and computes, through LDS, everything above it that the tile depends on — the (2PT+3)^2 region of
// the level above, the (4PT+9)^2 region two above, from the (8PT+21)^2 region of the source — writing the part of each level
// it owns (a 2x / 4x larger tile; the halo is recomputed by the neighbours: 1.9x / 2.4x redundant arithmetic at PT = 4 on images
// of a few hundred KB, against two kernel boundaries saved; PT = 8 gave 80 blocks at 640x480 and a 17 us launch). cv::pyrDown on CV_8UC1 (frame::constructImagePyramids, Frame.cpp:170-182): separable [1 4 6 4 1], BORDER_REFLECT_101, integer accumulation, (sum + 128) >> 8, dst = ((w + 1) / 2, (h + 1) / 2).
// (r03, measured and dropped: the separable 5 + 5 form — row sums of the region above in LDS, then the column pass, 10 reads per value
// instead of 25 — 12.96 against 11.45 us at 640x480: the launch is bound by its barriers and index arithmetic, not by its LDS reads.)
// Regions are kept in image coordinates clipped to the level (REFLECT_101 is applied to coordinates, and a reflected
// coordinate of a position the tile needs lies inside the clipped region).
#define ELLC_PT 4
struct PyrChainArgs {
  const uint8_t* src;            // level l
  uint8_t* dst[3];               // levels l+1 .. l+3 (only the first `steps` are used)
  int w[4], h[4];                // stored sizes of levels l .. l+3
  int steps;                     // 1..3
};
__device__ __forceinline__ int pyr5(const uint8_t* t, int stride, int x0, int y0, int lox, int loy, int sw, int sh, int x, int y) {
  // (sum over the 5x5 [1 4 6 4 1]^2 window centred at (2x, 2y) of the level above + 128) >> 8; t holds that level from (lox, loy)
  const int wk[5] = {1, 4, 6, 4, 1};
  int cx[5];
#pragma unroll
  for (int k = 0; k < 5; k++) cx[k] = reflect101(2 * x + k - 2, sw) - lox;
  int v = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const uint8_t* r = t + (reflect101(2 * y + j - 2, sh) - loy) * stride;
    int hsum = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) hsum += wk[k] * (int)r[cx[k]];
    v += wk[j] * hsum;
  }
  (void)x0; (void)y0;
  return (v + 128) >> 8;
}
__global__ __launch_bounds__(256) void pyr_down_chain_u8(PyrChainArgs a) {
  constexpr int R1 = 2 * ELLC_PT + 3, R2 = 2 * R1 + 3, R3 = 2 * R2 + 3;   // 11, 25, 53: region edge one, two, three levels above the tile
  __shared__ uint8_t lds[R3 * R3 + R2 * R2 + R1 * R1];   // three regions: 53^2, 25^2, 11^2
  const int S = a.steps;
  // owned tile at the deepest produced level (level index S relative to the source)
  int lox[4], loy[4], hix[4], hiy[4];   // needed region per level (relative index 0 = source), [lo, hi) clipped to the level
  lox[S] = blockIdx.x * ELLC_PT; loy[S] = blockIdx.y * ELLC_PT;
  hix[S] = min(a.w[S], lox[S] + ELLC_PT); hiy[S] = min(a.h[S], loy[S] + ELLC_PT);
  for (int l = S - 1; l >= 0; l--) {
    lox[l] = max(0, 2 * lox[l + 1] - 2); loy[l] = max(0, 2 * loy[l + 1] - 2);
    hix[l] = min(a.w[l], 2 * (hix[l + 1] - 1) + 3); hiy[l] = min(a.h[l], 2 * (hiy[l + 1] - 1) + 3);
  }
  // the level with relative index l lives in region l + 3 - S (so the source of a 3-step chain is the 53^2 region, of a
  // 1-step chain the 11^2 region)
  auto buf = [&](int l) {
    const int k = l + 3 - S;
    return lds + (k == 0 ? 0 : (k == 1 ? R3 * R3 : R3 * R3 + R2 * R2));
  };
  auto stride = [&](int l) { return hix[l] - lox[l]; };
  {   // source region
    const int wdt = stride(0), hgt = hiy[0] - loy[0];
    uint8_t* t = buf(0);
    for (int i = threadIdx.x; i < wdt * hgt; i += 256) {
      const int y = i / wdt, x = i - y * wdt;
      t[i] = a.src[(size_t)(loy[0] + y) * a.w[0] + lox[0] + x];
    }
  }
  __syncthreads();
  for (int l = 1; l <= S; l++) {
    const int wdt = stride(l), hgt = hiy[l] - loy[l];
    const uint8_t* up = buf(l - 1);
    const int ustride = stride(l - 1);
    // what this block owns of level l: the 2^(S-l) times larger tile (the region beyond it is halo, recomputed by neighbours)
    const int sc = ELLC_PT << (S - l);
    const int ox0 = blockIdx.x * sc, oy0 = blockIdx.y * sc, ox1 = ox0 + sc, oy1 = oy0 + sc;
    for (int i = threadIdx.x; i < wdt * hgt; i += 256) {
      const int yy = i / wdt, xx = i - yy * wdt;
      const int x = lox[l] + xx, y = loy[l] + yy;
      const int v = pyr5(up, ustride, 0, 0, lox[l - 1], loy[l - 1], a.w[l - 1], a.h[l - 1], x, y);
      if (l < S) buf(l)[i] = (uint8_t)v;
      if (x >= ox0 && x < ox1 && y >= oy0 && y < oy1) a.dst[l - 1][(size_t)y * a.w[l] + x] = (uint8_t)v;
    }
    __syncthreads();
  }
}

// The uploaded image, from its pinned staging buffer (host memory, read across PCIe by the kernel itself) to level 0 of the slot: 16
// bytes per lane. In place of a copy-engine transfer in front of the pyramid launch: copy engine -> compute queue is a hand-over
// between hardware queues on the same stream, which cost the upload path tens of microseconds per frame.
typedef uint32_t ingest_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ingest_copy_u8(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src_host, size_t bytes) {
  const size_t n16 = bytes >> 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    ((ingest_u32x4*)dst)[i] = __builtin_nontemporal_load((const ingest_u32x4*)src_host + i);
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[(n16 << 4) + threadIdx.x] = src_host[(n16 << 4) + threadIdx.x];
}

// frame::buildMaxGradients (Frame.cpp:618-674) in one launch: a 32 x 8 tile of outputs per block; the gradient magnitude of
// the tile plus a one-pixel ring goes through LDS, then the vertical and the horizontal 3-maximum with the reference's
// border rules (the three-pass form of r01 — magnitude, vertical, horizontal — is in the history: same operations per value, same
// order of the two fmaxf). count += pixels >= MIN_ABS_GRAD_DECREASE (zeroed by the caller).
__global__ __launch_bounds__(256) void maxgrad_fused(const uint8_t* __restrict__ img, int sw, int w, int h, float* __restrict__ out, int* count) {
  constexpr int TW = 32, TH = 8;
  __shared__ float mag[(TH + 2) * (TW + 2)];
  __shared__ float tmp[TH * (TW + 2)];
  __shared__ int hits_sh[4];
  const int bx = blockIdx.x * TW, by = blockIdx.y * TH;
  for (int i = threadIdx.x; i < (TH + 2) * (TW + 2); i += 256) {
    const int yy = i / (TW + 2), xx = i - yy * (TW + 2);
    const int x = bx + xx - 1, y = by + yy - 1;
    float m = 0.0f;
    if (x >= 0 && x < w && y >= 0 && y < h) {
      float gx, gy;
      grad_at(img, sw, w, h, x, y, gx, gy);
      const float p = gx * gx, q = gy * gy;
      m = sqrtf(p + q);
    }
    mag[i] = m;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TH * (TW + 2); i += 256) {   // vertical 3-maximum for the tile's rows, all TW + 2 columns
    const int yy = i / (TW + 2), xx = i - yy * (TW + 2);
    const int y = by + yy;
    float v = 0.0f;
    if (y >= 1 && y < h - 1) {
      const float g1 = fmaxf(mag[(yy + 1) * (TW + 2) + xx], mag[yy * (TW + 2) + xx]);
      v = fmaxf(g1, mag[(yy + 2) * (TW + 2) + xx]);
    }
    tmp[i] = v;
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x = bx + tx, y = by + ty;
  int hit = 0;
  if (x < w && y < h) {
    float v = mag[(ty + 1) * (TW + 2) + tx + 1];   // border pixels keep the raw magnitude
    if (y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {
      const float g1 = fmaxf(tmp[ty * (TW + 2) + tx], tmp[ty * (TW + 2) + tx + 1]);
      v = fmaxf(g1, tmp[ty * (TW + 2) + tx + 2]);
      hit = (v >= 5.0f) ? 1 : 0;   // MIN_ABS_GRAD_DECREASE
    }
    out[(size_t)y * w + x] = v;
  }
  const unsigned long long m = __ballot(hit != 0);
  if ((threadIdx.x & 63) == 0) hits_sh[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = hits_sh[0] + hits_sh[1] + hits_sh[2] + hits_sh[3];
    if (tot) atomicAdd(count, tot);
  }
}

// 1 / depth of a dense keyframe level as the list-free kernels use it: the reciprocal the compaction stores in a tolerance-mode
// record (v_rcp_f32), 0 where the pixel holds no depth. Written once per upload (ellc_keyframe_set_depth) for slots that carry the
// dense hint; gn_fca_dense4 then reads it in place of the depth plane (four reciprocals and four selects per thread and step less).
__global__ __launch_bounds__(256) void idepth_plane(const float* __restrict__ depth, float* __restrict__ idepth, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float z = depth[i];
    idepth[i] = z > 0.0f ? __builtin_amdgcn_rcpf(z) : 0.0f;
  }
}

// The exact mode's counterpart: pow(depth, -1) in double as the compaction stores it in a 20-byte record (1.0 / (double)Z: the same
// expression, the same bits), for the list-free exact kernel (gn_fca_dense_x) — an f64 division per pixel and ITERATION otherwise.
__global__ __launch_bounds__(256) void invz_plane(const float* __restrict__ depth, double* __restrict__ invz, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float z = depth[i];
    invz[i] = z > 0.0f ? 1.0 / (double)z : 1.0;
  }
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
