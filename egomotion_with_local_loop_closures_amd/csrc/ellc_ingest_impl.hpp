// Frame ingest (reference: frame::frame(VideoCapture), Frame.cpp:45-75): the decoded BGR frame ->
// grey (cvtColor BGR2GRAY) -> undistort (5-coefficient model, new camera matrix from getOptimalNewCameraMatrix with
// alpha = 0) -> resize by 1/DIM_FACTOR (INTER_LINEAR) -> level 0 of a frame slot, then the u8 pyramid. Decoding stays with
// the caller (no codecs here). OpenCV 3.0.0 is not part of the reference tree; its algorithms are restated:
//   * cvtColor 8u: (1868 B + 9617 G + 4899 R + 8192) >> 14
//   * getOptimalNewCameraMatrix: 9x9 grid through cvUndistortPoints (5 fixed-point iterations, double, points stored as
//     f32), inscribed rectangle, fx0 = (w-1)/inner.width ... ; the result takes the camera matrix' type (f32 here)
//   * undistort: stripes of max(1, 4096/w) rows; per stripe the new camera matrix with cy - y0 is inverted (3x3 closed
//     form in double) and initUndistortRectifyMap walks the row with running sums (_x += ir[0] ...), evaluates the
//     distortion model in double and stores fixed-point coordinates (1/32 px): CV_16SC2 + CV_16UC1
//   * remap INTER_LINEAR 8u, BORDER_CONSTANT(0): weights (32-fy)(32-fx)*32 etc. (sum 32768), (sum + 16384) >> 15
//   * resize by 1/4, INTER_LINEAR 8u: source coordinate 4 dx + 1.5 => equal weights on columns 4dx+1, 4dx+2 (rows
//     likewise); the fixed-point passes reduce to (p00 + p01 + p10 + p11 + 2) >> 2
// The maps do not depend on the frame: they are built once (host, sequential in the reference's order so that the running
// sums round the same way) for the 2x2 source pixels each output pixel reads, and kept on the device.
#pragma once
#include "ellc_context.hpp"
#include <cmath>
#include <vector>

namespace ellc {

struct IngestMapEntry {   // one undistorted source pixel: integer source position and the 5+5 bit fraction index
  short ix, iy;
  unsigned short frac;    // (fy5 << 5) | fx5
  unsigned short pad;
};

// ---- host: OpenCV's camera algebra, double unless stated ------------------------------------------------
static inline void ingest_inv3x3(const double* S, double* t) {   // cv::Mat::inv, 3x3 CV_64F closed form
  const double d0 = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d0 == 0.0) { for (int i = 0; i < 9; i++) t[i] = 0.0; return; }
  const double d = 1. / d0;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

// cvGetOptimalNewCameraMatrix(alpha = 0, newImgSize = imgSize, centerPrincipalPoint = false); K, dist are the f32 values
// the reference passes (cam_K, cam_distort); the result is rounded to f32 like the Mat it is returned in
static inline void ingest_optimal_new_camera(const float K[4] /*fx fy cx cy*/, const float dist[5], int w, int h, float Knew[4]) {
  const int N = 9;
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double ifx = 1. / fx, ify = 1. / fy;
  const double k[5] = {dist[0], dist[1], dist[2], dist[3], dist[4]};
  float iX0 = -3.402823466e+38f, iX1 = 3.402823466e+38f, iY0 = -3.402823466e+38f, iY1 = 3.402823466e+38f;
  for (int yy = 0; yy < N; yy++)
    for (int xx = 0; xx < N; xx++) {
      const float px = (float)xx * w / (N - 1), py = (float)yy * h / (N - 1);
      double x = (px - cx) * ifx, y = (py - cy) * ify;
      const double x0 = x, y0 = y;
      for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
      }
      const float ux = (float)x, uy = (float)y;
      if (xx == 0) iX0 = std::max(iX0, ux);
      if (xx == N - 1) iX1 = std::min(iX1, ux);
      if (yy == 0) iY0 = std::max(iY0, uy);
      if (yy == N - 1) iY1 = std::min(iY1, uy);
    }
  const float iw = iX1 - iX0, ih = iY1 - iY0;
  const double fx0 = (w - 1) / iw;   // int / float: f32 division, as in the reference library
  const double fy0 = (h - 1) / ih;
  const double cx0 = -fx0 * iX0;
  const double cy0 = -fy0 * iY0;
  Knew[0] = (float)fx0; Knew[1] = (float)fy0; Knew[2] = (float)cx0; Knew[3] = (float)cy0;
}

// Fixed-point undistortion map of the full-size image, row by row in cv::undistort's stripe order
static inline void ingest_build_maps(const float K[4], const float dist[5], const float Knew[4], int w, int h, int do_undistort,
                                     std::vector<IngestMapEntry>& map) {
  map.assign((size_t)w * h, IngestMapEntry{0, 0, 0, 0});
  if (!do_undistort) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) map[(size_t)y * w + x] = IngestMapEntry{(short)x, (short)y, 0, 0};
    return;
  }
  const double fx = K[0], fy = K[1], u0 = K[2], v0 = K[3];
  const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
  const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(w, 1)), h);
  const double cyn = Knew[3];
  for (int y0 = 0; y0 < h; y0 += stripe0) {
    const int stripe = std::min(stripe0, h - y0);
    const double Ar[9] = {(double)Knew[0], 0, (double)Knew[2], 0, (double)Knew[1], cyn - y0, 0, 0, 1};
    double ir[9];
    ingest_inv3x3(Ar, ir);
    for (int i = 0; i < stripe; i++) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double ww = 1. / _w, x = _x * ww, y = _y * ww;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2);
        const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
        const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
        const int iu = (int)std::nearbyint(u * 32), iv = (int)std::nearbyint(v * 32);   // saturate_cast<int>: round half to even
        IngestMapEntry e;
        // saturate_cast<short>, as cv::convertMaps stores the integer part (a distortion model evaluated far outside the image
        // can leave the 16-bit range; such entries only have to stay out of bounds, not wrap around into the image)
        e.ix = (short)std::max(-32768, std::min(32767, iu >> 5));
        e.iy = (short)std::max(-32768, std::min(32767, iv >> 5));
        e.frac = (unsigned short)((iv & 31) * 32 + (iu & 31));
        e.pad = 0;
        map[(size_t)(y0 + i) * w + j] = e;
      }
    }
  }
}

// ---- device ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int ingest_gray(const uint8_t* __restrict__ bgr, int w, int h, int x, int y) {
  if (x < 0 || x >= w || y < 0 || y >= h) return 0;   // BORDER_CONSTANT, value 0
  const uint8_t* p = bgr + ((size_t)y * w + x) * 3;
  return (1868 * (int)p[0] + 9617 * (int)p[1] + 4899 * (int)p[2] + 8192) >> 14;
}

// one thread per output pixel: four undistorted source pixels (remap), averaged (resize by 1/factor = 1/4)
__global__ __launch_bounds__(256) void ingest_frame(const uint8_t* __restrict__ bgr, int w, int h, const IngestMapEntry* __restrict__ map4,
                                                    uint8_t* __restrict__ out, int ow, int oh, int sw, uint8_t* __restrict__ gray_dbg,
                                                    uint8_t* __restrict__ und_dbg) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
  if (dx >= ow || dy >= oh) return;
  int sum = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const IngestMapEntry e = map4[((size_t)dy * ow + dx) * 4 + q];
    const int fx5 = e.frac & 31, fy5 = e.frac >> 5;
    const int w00 = (32 - fy5) * (32 - fx5) * 32, w01 = (32 - fy5) * fx5 * 32, w10 = fy5 * (32 - fx5) * 32, w11 = fy5 * fx5 * 32;
    const int g00 = ingest_gray(bgr, w, h, e.ix, e.iy), g01 = ingest_gray(bgr, w, h, e.ix + 1, e.iy);
    const int g10 = ingest_gray(bgr, w, h, e.ix, e.iy + 1), g11 = ingest_gray(bgr, w, h, e.ix + 1, e.iy + 1);
    const int v = (g00 * w00 + g01 * w01 + g10 * w10 + g11 * w11 + (1 << 14)) >> 15;
    sum += v;
    if (und_dbg) und_dbg[((size_t)dy * ow + dx) * 4 + q] = (uint8_t)v;
  }
  out[(size_t)dy * sw + dx] = (uint8_t)((sum + 2) >> 2);
  if (gray_dbg) gray_dbg[(size_t)dy * ow + dx] = (uint8_t)ingest_gray(bgr, w, h, 4 * dx + 1, 4 * dy + 1);
}

}  // namespace ellc

extern "C" {

ellc_status ellc_ingest_configure(ellc_ctx* c, int orig_w, int orig_h, float fx, float fy, float cx, float cy, const float* dist5,
                                  int do_undistort, float* new_camera4) {
  ELLC_ENTER(c);
  using namespace ellc;
  if (!c || orig_w < 8 || orig_h < 8 || (do_undistort && !dist5)) return fail(c, ELLC_ERR_BAD_ARG, "ellc_ingest_configure: bad argument");
  if (orig_w != 4 * c->cfg.width || orig_h != 4 * c->cfg.height)
    return fail(c, ELLC_ERR_BAD_ARG, "ellc_ingest_configure: the context size must be the input size / 4 (DIM_FACTOR, ExternVariable.h:41)");
  if (orig_w > 32767 || orig_h > 32767) return fail(c, ELLC_ERR_BAD_ARG, "ellc_ingest_configure: image too large for the 16-bit map");
  const float K[4] = {fx, fy, cx, cy};
  float d[5] = {0, 0, 0, 0, 0};
  if (dist5) for (int i = 0; i < 5; i++) d[i] = dist5[i];
  float Knew[4] = {fx, fy, cx, cy};
  if (do_undistort) ingest_optimal_new_camera(K, d, orig_w, orig_h, Knew);
  if (new_camera4) for (int i = 0; i < 4; i++) new_camera4[i] = Knew[i];
  std::vector<IngestMapEntry> full;
  ingest_build_maps(K, d, Knew, orig_w, orig_h, do_undistort, full);
  const int ow = c->cfg.width, oh = c->cfg.height;
  std::vector<IngestMapEntry> need((size_t)ow * oh * 4);
  for (int dy = 0; dy < oh; dy++)
    for (int dx = 0; dx < ow; dx++)
      for (int q = 0; q < 4; q++)
        need[((size_t)dy * ow + dx) * 4 + q] = full[(size_t)(4 * dy + 1 + (q >> 1)) * orig_w + (4 * dx + 1 + (q & 1))];
  if (c->ingest_map) (void)hipFree(c->ingest_map);
  if (c->ingest_bgr) (void)hipFree(c->ingest_bgr);
  c->ingest_map = nullptr;
  c->ingest_bgr = nullptr;
  hipError_t ie = hipMalloc(&c->ingest_map, need.size() * sizeof(IngestMapEntry));
  if (ie == hipSuccess) ie = hipMalloc((void**)&c->ingest_bgr, (size_t)orig_w * orig_h * 3);
  if (ie == hipSuccess) ie = hipMemcpyAsync(c->ingest_map, need.data(), need.size() * sizeof(IngestMapEntry), hipMemcpyHostToDevice, c->stream);
  if (ie == hipSuccess) ie = hipStreamSynchronize(c->stream);
  if (ie != hipSuccess) {   // all or nothing: a half-configured ingest would pass the "configured" check of ellc_frame_ingest_bgr
    if (c->ingest_map) (void)hipFree(c->ingest_map);
    if (c->ingest_bgr) (void)hipFree(c->ingest_bgr);
    c->ingest_map = nullptr;
    c->ingest_bgr = nullptr;
    return fail(c, ELLC_ERR_HIP, std::string("ellc_ingest_configure: ") + hipGetErrorString(ie));
  }
  c->ingest_w = orig_w;
  c->ingest_h = orig_h;
  return ELLC_OK;
}

ellc_status ellc_frame_ingest_bgr(ellc_ctx* c, int slot, const uint8_t* bgr, uint8_t* gray_probe, uint8_t* undistorted_probe) {
  ELLC_ENTER(c);
  using namespace ellc;
  if (!c || !bgr || !slot_ok(slot, c->cfg.max_frames)) return fail(c, ELLC_ERR_BAD_ARG, "ellc_frame_ingest_bgr: bad argument");
  if (!c->ingest_map) return fail(c, ELLC_ERR_NOT_READY, "ellc_frame_ingest_bgr: call ellc_ingest_configure first");
  const int w = c->ingest_w, h = c->ingest_h, ow = c->cfg.width, oh = c->cfg.height;
  ELLC_HIP(c, hipMemcpyAsync(c->ingest_bgr, bgr, (size_t)w * h * 3, hipMemcpyHostToDevice, c->stream));
  uint8_t *gd = nullptr, *ud = nullptr;
  if (gray_probe) ELLC_HIP(c, hipMalloc(&gd, (size_t)ow * oh));
  if (undistorted_probe && hipMalloc(&ud, (size_t)ow * oh * 4) != hipSuccess) {
    if (gd) (void)hipFree(gd);
    return fail(c, ELLC_ERR_HIP, "ellc_frame_ingest_bgr: cannot allocate the probe buffer");
  }
  uint8_t* img[ELLC_MAX_LEVELS];
  for (int l = 0; l < c->L; l++) img[l] = c->fr_tab_h[(size_t)l * c->cfg.max_frames + slot].img;
  const LevelGeom* g = c->geom_h;
  dim3 blk(32, 8);
  hipLaunchKernelGGL(ingest_frame, grid2d(ow, oh, blk), blk, 0, c->stream, c->ingest_bgr, w, h, (const IngestMapEntry*)c->ingest_map, img[0], ow, oh,
                     g[0].sw, gd, ud);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && build_image_pyramid(c, img, c->stream) != ELLC_OK) e = hipErrorUnknown;
  if (e == hipSuccess && gd) e = hipMemcpyAsync(gray_probe, gd, (size_t)ow * oh, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess && ud) e = hipMemcpyAsync(undistorted_probe, ud, (size_t)ow * oh * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // the host frame buffer may be pageable
  if (gd) (void)hipFree(gd);
  if (ud) (void)hipFree(ud);
  if (e != hipSuccess) return fail(c, ELLC_ERR_HIP, std::string("ellc_frame_ingest_bgr: ") + hipGetErrorString(e));
  c->fr_has_image[slot] = 1;
  c->fr_maxgrad_valid[slot] = 0;
  return ELLC_OK;
}

}  // extern "C"
