// ELLC ORACLE (test infrastructure) — se(3) algebra, LU inverse, image side.
// See ellc_oracle.hpp for the header note ("parity unpinned", citations).
#include "ellc_oracle.hpp"
#include <cmath>
#include <cstring>
#include <cfloat>
#include <algorithm>

namespace ellc_oracle {

Intrin get_intrinsic(const Config& c, int level) {
  // UserDefinedFunc.cpp:40-43 : ORIG_FX / pow(2, pyrlevel) evaluated in double, stored float.
  double s = std::pow(2.0, level);
  Intrin k;
  k.fx = (float)((double)c.fx / s);
  k.fy = (float)((double)c.fy / s);
  k.cx = (float)((double)c.cx / s);
  k.cy = (float)((double)c.cy / s);
  return k;
}

// ------------------------------------------------------------------ se(3)
// Coefficients A = sin(t)/t, B = (1-cos t)/t^2, C = (t - sin t)/t^3 (all even in t).
static void abc_coeffs(double t2, double& A, double& B, double& C) {
  double t = std::sqrt(t2);
  if (t < 0.1) {
    // Taylor in t^2; remainder < 1e-19 for t < 0.1
    A = 1.0 + t2 * (-1.0 / 6 + t2 * (1.0 / 120 + t2 * (-1.0 / 5040 + t2 * (1.0 / 362880 - t2 / 39916800.0))));
    B = 0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 - t2 / 479001600.0))));
    C = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 - t2 / 6227020800.0))));
  } else {
    double s = std::sin(t), h = std::sin(0.5 * t);
    A = s / t;
    B = 2.0 * h * h / t2;
    C = (t - s) / (t2 * t);
  }
}

void se3_exp_d(const double p[6], double T[16]) {
  double wx = p[0], wy = p[1], wz = p[2];
  double t2 = wx * wx + wy * wy + wz * wz;
  double A, B, C;
  abc_coeffs(t2, A, B, C);
  // W = hat(w); W2 = W*W = w w^T - t2 I
  double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9] = {wx * wx - t2, wx * wy, wx * wz, wy * wx, wy * wy - t2, wy * wz, wz * wx, wz * wy, wz * wz - t2};
  double R[9], V[9];
  for (int i = 0; i < 9; i++) {
    double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = I + A * W[i] + B * W2[i];
    V[i] = I + B * W[i] + C * W2[i];
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[r * 4 + c] = R[r * 3 + c];
    T[r * 4 + 3] = V[r * 3 + 0] * p[3] + V[r * 3 + 1] * p[4] + V[r * 3 + 2] * p[5];
  }
  T[12] = T[13] = T[14] = 0.0;
  T[15] = 1.0;
}

void se3_log_d(const double T[16], double p[6]) {
  double R00 = T[0], R01 = T[1], R02 = T[2], R10 = T[4], R11 = T[5], R12 = T[6], R20 = T[8], R21 = T[9], R22 = T[10];
  double c = 0.5 * (R00 + R11 + R22 - 1.0);
  double rx = 0.5 * (R21 - R12), ry = 0.5 * (R02 - R20), rz = 0.5 * (R10 - R01);
  const double s2 = rx * rx + ry * ry + rz * rz;   // sin^2 of the rotation angle, from the skew part
  double s = std::sqrt(s2);
  double t = std::atan2(s, c);
  double wx, wy, wz;
  if (c > 0.9 && s2 < 1e-2) {
    // Small rotation (every Gauss-Newton update, every frame-to-keyframe pose): angle / sin = asin(s)/s as a series in
    // s^2 (truncation < 2e-16 for s^2 < 0.01). The input is an f32-ROUNDED matrix, i.e. not exactly orthogonal; then
    // atan2(s, c) (which also looks at the trace) and asin(s) differ by up to one f32 ulp of the result (measured: in
    // 35 % of random small poses), both ~2.5e-8 away from the exact log of the rounded matrix, neither closer to Eigen's
    // f32 Schur/Pade log. The skew-part form is used so that this restatement and the device code, which evaluates the
    // same series (no sqrt / atan2 on its critical path), agree bit for bit and per-pixel comparisons downstream of a
    // pose (depth observation / propagation) stay exact.
    const double f = 1.0 + s2 * (1.0 / 6.0 + s2 * (3.0 / 40.0 + s2 * (15.0 / 336.0 + s2 * (105.0 / 3456.0 + s2 * (945.0 / 42240.0 +
                     s2 * (10395.0 / 599040.0 + s2 * (135135.0 / 9676800.0)))))));
    wx = rx * f; wy = ry * f; wz = rz * f;
  } else if (c > -0.99) {
    const double f = (s < 1e-8) ? 1.0 : t / s;
    wx = rx * f; wy = ry * f; wz = rz * f;
  } else {
    // near pi: axis from the symmetric part, sign from the skew part
    double omc = 1.0 - c;
    double ax = std::sqrt(std::max(0.0, (R00 - c) / omc));
    double ay = std::sqrt(std::max(0.0, (R11 - c) / omc));
    double az = std::sqrt(std::max(0.0, (R22 - c) / omc));
    if (s > 1e-12) {
      if (rx < 0) ax = -ax;
      if (ry < 0) ay = -ay;
      if (rz < 0) az = -az;
    } else {
      // exactly pi: fix signs from off-diagonal symmetric terms relative to the largest component
      double sxy = R01 + R10, sxz = R02 + R20, syz = R12 + R21;
      if (ax >= ay && ax >= az) { if (sxy < 0) ay = -ay; if (sxz < 0) az = -az; }
      else if (ay >= az) { if (sxy < 0) ax = -ax; if (syz < 0) az = -az; }
      else { if (sxz < 0) ax = -ax; if (syz < 0) ay = -ay; }
    }
    double n = std::sqrt(ax * ax + ay * ay + az * az);
    if (n > 0) { ax /= n; ay /= n; az /= n; }
    wx = ax * t; wy = ay * t; wz = az * t;
  }
  double t2 = wx * wx + wy * wy + wz * wz;
  // V^-1 = I - W/2 + D W^2,  D = (1 - A/(2B))/t^2
  double D;
  if (t2 < 0.01) {
    D = 1.0 / 12 + t2 * (1.0 / 720 + t2 * (1.0 / 30240 + t2 * (1.0 / 1209600 + t2 / 47900160.0)));
  } else {
    double A, B, C;
    abc_coeffs(t2, A, B, C);
    D = (1.0 - A / (2.0 * B)) / t2;
  }
  double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9] = {wx * wx - t2, wx * wy, wx * wz, wy * wx, wy * wy - t2, wy * wz, wz * wx, wz * wy, wz * wz - t2};
  double tx = T[3], ty = T[7], tz = T[11];
  double v[3];
  for (int r = 0; r < 3; r++) {
    double m0 = (r == 0 ? 1.0 : 0.0) - 0.5 * W[r * 3 + 0] + D * W2[r * 3 + 0];
    double m1 = (r == 1 ? 1.0 : 0.0) - 0.5 * W[r * 3 + 1] + D * W2[r * 3 + 1];
    double m2 = (r == 2 ? 1.0 : 0.0) - 0.5 * W[r * 3 + 2] + D * W2[r * 3 + 2];
    v[r] = m0 * tx + m1 * ty + m2 * tz;
  }
  p[0] = wx; p[1] = wy; p[2] = wz; p[3] = v[0]; p[4] = v[1]; p[5] = v[2];
}

void se3_exp(const float pose[6], float T[16]) {
  double p[6], Td[16];
  for (int i = 0; i < 6; i++) p[i] = pose[i];
  se3_exp_d(p, Td);
  for (int i = 0; i < 16; i++) T[i] = (float)Td[i];
}

void se3_log(const float T[16], float pose[6]) {
  double Td[16], p[6];
  for (int i = 0; i < 16; i++) Td[i] = T[i];
  se3_log_d(Td, p);
  for (int i = 0; i < 6; i++) pose[i] = (float)p[i];
}

// f32 4x4 product (Eigen f32 GEMM restated: sums taken in double, rounded once to f32)
static void mat4_mul_f32(const float A[16], const float B[16], float C[16]) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += (double)A[r * 4 + k] * (double)B[k * 4 + c];
      C[r * 4 + c] = (float)s;
    }
}

// inverse of an SE(3) matrix held in f32 (Eigen .inverse() restated analytically in double)
static void se3_inverse_f32(const float T[16], float Ti[16]) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Ti[r * 4 + c] = T[c * 4 + r];
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)T[k * 4 + r] * (double)T[k * 4 + 3];
    Ti[r * 4 + 3] = (float)(-s);
  }
  Ti[12] = Ti[13] = Ti[14] = 0.f;
  Ti[15] = 1.f;
}

void concatenate_relative_pose(const float a[6], const float b[6], float out[6]) {
  float A[16], B[16], C[16];
  se3_exp(a, A);
  se3_exp(b, B);
  mat4_mul_f32(A, B, C);
  float o[6];
  se3_log(C, o);
  std::memcpy(out, o, sizeof(o));
}

void concatenate_origin_pose(const float a[6], const float b[6], float out[6]) {
  float A[16], B[16], Bi[16], C[16];
  se3_exp(a, A);
  se3_exp(b, B);
  se3_inverse_f32(B, Bi);
  mat4_mul_f32(A, Bi, C);
  float o[6];
  se3_log(C, o);
  std::memcpy(out, o, sizeof(o));
}

void inv_lie_pose(const float a[6], float out[6]) {
  float A[16], Ai[16];
  se3_exp(a, A);
  se3_inverse_f32(A, Ai);
  float o[6];
  se3_log(Ai, o);
  std::memcpy(out, o, sizeof(o));
}

// ------------------------------------------------------------------ LU inverse
// OpenCV 3.0.0 modules/core/src/lapack.cpp LUImpl<float> applied to (A | I), as cv::Mat::inv does
// for n > 3 (PixelWisePyramid.cpp:451). Pivot threshold std::numeric_limits<float>::epsilon().
int lu_inverse_f32(const float* Ain, int n, float* out) {
  std::vector<float> A(Ain, Ain + n * n);
  float* b = out;
  for (int i = 0; i < n * n; i++) b[i] = 0.f;
  for (int i = 0; i < n; i++) b[i * n + i] = 1.f;
  for (int i = 0; i < n; i++) {
    int k = i;
    for (int j = i + 1; j < n; j++)
      if (std::fabs(A[j * n + i]) > std::fabs(A[k * n + i])) k = j;
    if (std::fabs(A[k * n + i]) < FLT_EPSILON) {
      for (int q = 0; q < n * n; q++) out[q] = 0.f;
      return 0;
    }
    if (k != i) {
      for (int j = i; j < n; j++) std::swap(A[i * n + j], A[k * n + j]);
      for (int j = 0; j < n; j++) std::swap(b[i * n + j], b[k * n + j]);
    }
    float d = -1 / A[i * n + i];
    for (int j = i + 1; j < n; j++) {
      float alpha = A[j * n + i] * d;
      for (int q = i + 1; q < n; q++) A[j * n + q] += alpha * A[i * n + q];
      for (int q = 0; q < n; q++) b[j * n + q] += alpha * b[i * n + q];
    }
  }
  for (int i = n - 1; i >= 0; i--)
    for (int j = 0; j < n; j++) {
      float s = b[i * n + j];
      for (int q = i + 1; q < n; q++) s -= A[i * n + q] * b[q * n + j];
      b[i * n + j] = s / A[i * n + i];
    }
  return 1;
}

// ------------------------------------------------------------------ K matrices
KMats make_kmats(const Config& c) {
  KMats m;
  float K[9] = {c.fx, 0, c.cx, 0, c.fy, c.cy, 0, 0, 1};
  std::memcpy(m.K, K, sizeof(K));
  // cv::Mat::inv for 3x3 CV_32F: determinant in double (det3 macro), cofactors in f32, scaled by 1/det in double.
#define S(i, j) K[(i) * 3 + (j)]
  double d = S(0, 0) * ((double)S(1, 1) * S(2, 2) - (double)S(1, 2) * S(2, 1)) -
             S(0, 1) * ((double)S(1, 0) * S(2, 2) - (double)S(1, 2) * S(2, 0)) +
             S(0, 2) * ((double)S(1, 0) * S(2, 1) - (double)S(1, 1) * S(2, 0));
  if (d != 0.) {
    d = 1. / d;
    float t[9];
    t[0] = (float)((S(1, 1) * S(2, 2) - S(1, 2) * S(2, 1)) * d);
    t[1] = (float)((S(0, 2) * S(2, 1) - S(0, 1) * S(2, 2)) * d);
    t[2] = (float)((S(0, 1) * S(1, 2) - S(0, 2) * S(1, 1)) * d);
    t[3] = (float)((S(1, 2) * S(2, 0) - S(1, 0) * S(2, 2)) * d);
    t[4] = (float)((S(0, 0) * S(2, 2) - S(0, 2) * S(2, 0)) * d);
    t[5] = (float)((S(0, 2) * S(1, 0) - S(0, 0) * S(1, 2)) * d);
    t[6] = (float)((S(1, 0) * S(2, 1) - S(1, 1) * S(2, 0)) * d);
    t[7] = (float)((S(0, 1) * S(2, 0) - S(0, 0) * S(2, 1)) * d);
    t[8] = (float)((S(0, 0) * S(1, 1) - S(0, 1) * S(1, 0)) * d);
    std::memcpy(m.Kinv, t, sizeof(t));
  } else {
    for (int i = 0; i < 9; i++) m.Kinv[i] = 0;
  }
#undef S
  m.fx_inv = m.Kinv[0];
  m.cx_inv = m.Kinv[2];
  m.fy_inv = m.Kinv[4];
  m.cy_inv = m.Kinv[5];
  return m;
}

// ------------------------------------------------------------------ image side
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

// cv::pyrDown on CV_8UC1 (OpenCV 3.0.0 pyramids.cpp pyrDown_<FixPtCast<uchar,8>>): separable
// [1 4 6 4 1], integer accumulation, BORDER_REFLECT_101, dst = ((w+1)/2, (h+1)/2), (sum+128)>>8.
void pyr_down_u8(const PlaneU8& src, PlaneU8& dst) {
  int sw = src.w, sh = src.h;
  int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  dst = PlaneU8(dw, dh);
  std::vector<int> hrow((size_t)dw * sh);
  for (int y = 0; y < sh; y++) {
    const uint8_t* s = src.row(y);
    for (int x = 0; x < dw; x++) {
      int c = 2 * x;
      int v = s[reflect101(c - 2, sw)] + 4 * s[reflect101(c - 1, sw)] + 6 * s[reflect101(c, sw)] +
              4 * s[reflect101(c + 1, sw)] + s[reflect101(c + 2, sw)];
      hrow[(size_t)y * dw + x] = v;
    }
  }
  for (int y = 0; y < dh; y++) {
    int c = 2 * y;
    const int* r0 = &hrow[(size_t)reflect101(c - 2, sh) * dw];
    const int* r1 = &hrow[(size_t)reflect101(c - 1, sh) * dw];
    const int* r2 = &hrow[(size_t)reflect101(c, sh) * dw];
    const int* r3 = &hrow[(size_t)reflect101(c + 1, sh) * dw];
    const int* r4 = &hrow[(size_t)reflect101(c + 2, sh) * dw];
    uint8_t* d = dst.row(y);
    for (int x = 0; x < dw; x++) {
      int v = r0[x] + 4 * r1[x] + 6 * r2[x] + 4 * r3[x] + r4[x];
      d[x] = (uint8_t)((v + 128) >> 8);
    }
  }
}

// Frame.cpp:185-285. rows/cols are the *iterated* sizes (height/2^l), which can be one less than
// the stored image size for odd dimensions (Q13); the stored stride is img.w.
void calculate_gradient(const PlaneU8& img, int rows, int cols, PlaneF& gx, PlaneF& gy) {
  gx = PlaneF(cols, rows, 0.f);
  gy = PlaneF(cols, rows, 0.f);
  auto I = [&](int y, int x) -> float { return (float)img.at(y, x); };
  for (int y = 0; y < rows; y++) {
    for (int x = 0; x < cols; x++) {
      float dx, dy;
      if (x == 0) dx = I(y, x + 1) - I(y, x);
      else if (x == cols - 1) dx = I(y, x) - I(y, x - 1);
      else dx = 0.5f * (I(y, x + 1) - I(y, x - 1));
      if (y == 0) dy = I(y + 1, x) - I(y, x);
      else if (y == rows - 1) dy = I(y, x) - I(y - 1, x);
      else dy = 0.5f * (I(y + 1, x) - I(y - 1, x));
      gx.at(y, x) = dx;
      gy.at(y, x) = dy;
    }
  }
}

// Frame.cpp:618-674
void build_max_gradients(const PlaneF& gx, const PlaneF& gy, PlaneF& out, int* n_substantial) {
  int w = gx.w, h = gx.h;
  out = PlaneF(w, h, 0.f);
  for (size_t i = 0; i < out.d.size(); i++) {
    float a = gx.d[i] * gx.d[i];
    float b = gy.d[i] * gy.d[i];
    out.d[i] = std::sqrt(a + b);
  }
  PlaneF tmp(w, h, 0.f);
  for (int y = 1; y < h - 1; y++)
    for (int x = 0; x < w; x++) {
      float g1 = std::max(out.at(y, x), out.at(y - 1, x));
      tmp.at(y, x) = std::max(g1, out.at(y + 1, x));
    }
  int n = 0;
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      float g1 = std::max(tmp.at(y, x - 1), tmp.at(y, x));
      float g = std::max(g1, tmp.at(y, x + 1));
      out.at(y, x) = g;
      if (g >= 5.0f) n++;  // MIN_ABS_GRAD_DECREASE
    }
  if (n_substantial) *n_substantial = n;
}

// Shared structure of the two reference taps (Frame.h:181-279 and :283-394): four taps with the
// reference's exact per-tap bounds expressions. fetch(y,x) returns the value as float.
template <class Fetch>
static inline float bilinear_ref(Fetch fetch, int curRows, int curCols, float x1, float y1, int check, bool* all_oob) {
  const int nCols = curCols - 1, nRows = curRows - 1;
  int countOOB = 0;
  float wy = y1 - std::floor(y1);
  float wx = x1 - std::floor(x1);
  float p1, p2, y, x;
  // case 1: row floor(y)
  y = std::floor(y1);
  x = std::floor(x1);
  if ((x < 0) || (x > nCols) || (y < 0) || (y > nRows)) { p1 = 0; countOOB++; }
  else p1 = fetch((int)y, (int)x);
  x = x1;
  if ((x < 0) || (x > nCols) || (y < 0) || (y > nRows)) { p2 = 0; countOOB++; }
  else p2 = fetch((int)y, (int)std::ceil(x));
  float top = ((1 - wx) * p1) + (wx * p2);
  // case 2: row ceil(y) (bounds tested on the unfloored y)
  y = y1;
  x = std::floor(x1);
  if ((x < 0) || (x > nCols) || (y < 0) || (y > nRows)) { p1 = 0; countOOB++; }
  else p1 = fetch((int)std::ceil(y), (int)x);
  x = x1;
  if ((x < 0) || (x > nCols) || (y < 0) || (y > nRows)) { p2 = 0; countOOB++; }
  else p2 = fetch((int)std::ceil(y), (int)std::ceil(x));
  if (all_oob) *all_oob = (countOOB == 4);
  if (countOOB == 4 && check == 1) return -1.0f;
  float btm = ((1 - wx) * p1) + (wx * p2);
  return ((1 - wy) * top) + (wy * btm);
}

float tap_u8(const PlaneU8& img, int curRows, int curCols, float x1, float y1, int check) {
  if (std::isnan(x1) || std::isnan(y1)) return check ? -1.0f : 0.0f;  // reference: UB; treated as out of bounds
  return bilinear_ref([&](int y, int x) { return (float)img.at(y, x); }, curRows, curCols, x1, y1, check, nullptr);
}

float tap_f32(const PlaneF& img, int curRows, int curCols, float x1, float y1) {
  if (std::isnan(x1) || std::isnan(y1)) return 0.0f;
  return bilinear_ref([&](int y, int x) { return img.at(y, x); }, curRows, curCols, x1, y1, 0, nullptr);
}

// ------------------------------------------------------------------ Frame
void Frame::init(const Config& c, const uint8_t* gray, int id) {
  cfg = c;
  frameId = id;
  width = c.width;
  height = c.height;
  pyrLevel = 0;
  currentCols = width;
  currentRows = height;
  for (int i = 0; i < 6; i++) poseWrtOrigin[i] = poseWrtWorld[i] = 0.f;
  rescaleFactor = 1.0f;
  image_pyramid.assign(c.levels, PlaneU8());
  image_pyramid[0] = PlaneU8(width, height);
  std::memcpy(image_pyramid[0].d.data(), gray, (size_t)width * height);
  constructImagePyramids();
  calculateGradient();
  buildMaxGradients();
  depth_pyramid.clear();
  weight_pyramid.clear();
  for (int l = 0; l < c.levels; l++) {
    depth_pyramid.push_back(PlaneF(width >> l, height >> l, 0.f));
    weight_pyramid.push_back(PlaneF(width >> l, height >> l, 0.f));
    numWeightsAdded[l] = 0;
  }
}

void Frame::constructImagePyramids() {
  for (int l = 1; l < cfg.levels; l++) pyr_down_u8(image_pyramid[l - 1], image_pyramid[l]);
}

void Frame::calculateGradient() { calculate_gradient(image_pyramid[pyrLevel], currentRows, currentCols, gradientx, gradienty); }

void Frame::calculateNonZeroDepthPts() {
  const PlaneF& d = depth_pyramid[pyrLevel];
  mask = PlaneU8(d.w, d.h, 0);
  int n = 0;
  for (size_t i = 0; i < d.d.size(); i++) {
    bool v = d.d[i] > 0.0f;
    mask.d[i] = v ? 255 : 0;
    n += v;
  }
  no_nonZeroDepthPts = n;
}

void Frame::updationOnPyrChange(int level, bool isPrevious) {
  pyrLevel = level;
  currentRows = (int)(height / std::pow(2, level));
  currentCols = (int)(width / std::pow(2, level));
  if (isPrevious) calculateNonZeroDepthPts();
  calculateGradient();
}

void Frame::buildMaxGradients() { build_max_gradients(gradientx, gradienty, maxAbsGradient, &no_points_substantial_grad); }

void Frame::finaliseWeights() {
  for (int l = cfg.levels - 1; l >= 0; l--)
    if (numWeightsAdded[l] > 0) {
      // cv: Mat / int -> MatExpr a*(1/n) -> convertTo(32f->32f): work type f32, scale = (float)(1.0/n)
      float s = (float)(1.0 / (double)numWeightsAdded[l]);
      for (auto& v : weight_pyramid[l].d) v = v * s + 0.0f;
    }
}

void Frame::calculateSE3poseOtherWrtThis(const Frame& other) {
  float rel[6];
  concatenate_origin_pose(other.poseWrtOrigin, poseWrtOrigin, rel);
  se3_exp(rel, SE3poseOtherWrtThis);
  se3_inverse_f32(SE3poseOtherWrtThis, SE3poseThisWrtOther);
  KMats km = make_kmats(cfg);
  auto kmul = [&](const float* T, float* Kr, float* Kt) {
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) {
        float s = 0;
        for (int k = 0; k < 3; k++) s += km.K[r * 3 + k] * T[k * 4 + c];
        Kr[r * 3 + c] = s;
      }
      float s = 0;
      for (int k = 0; k < 3; k++) s += km.K[r * 3 + k] * T[k * 4 + 3];
      Kt[r] = s;
    }
  };
  kmul(SE3poseThisWrtOther, K_SE3poseThisWrtOther_r, K_SE3poseThisWrtOther_t);
  kmul(SE3poseOtherWrtThis, K_SE3poseOtherWrtThis_r, K_SE3poseOtherWrtThis_t);
}

}  // namespace ellc_oracle
