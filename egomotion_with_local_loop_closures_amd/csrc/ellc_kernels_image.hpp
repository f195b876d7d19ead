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
