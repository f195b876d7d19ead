// se(3) <-> SE(3) in closed form, evaluated in double and rounded to f32 at the interface, for the
// Gauss-Newton pose update  pose <- log(exp(delta^) * exp(pose^))  (reference: frame::concatenateRelativePose,
// Frame.cpp:503-530; the reference calls Eigen's 4x4 f32 matrix exp/log, which this replaces).
// Usable from host and device code.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define ELLC_HD __host__ __device__ __forceinline__
#else
#define ELLC_HD inline
#endif

namespace ellc {

struct Rt {          // rigid transform, rotation row-major + translation, double
  double R[9];
  double t[3];
};

// sin(t)/t, (1-cos t)/t^2, (t-sin t)/t^3 as functions of q = t^2
ELLC_HD void sinc_family(double q, double& s1, double& s2, double& s3) {
  if (q < 1e-2) {
    // alternating series in q, truncated where the next term is < 1e-19 for q < 0.01
    s1 = 1.0 - q * (1.0 / 6.0) * (1.0 - q * (1.0 / 20.0) * (1.0 - q * (1.0 / 42.0) * (1.0 - q * (1.0 / 72.0) * (1.0 - q / 110.0))));
    s2 = 0.5 * (1.0 - q * (1.0 / 12.0) * (1.0 - q * (1.0 / 30.0) * (1.0 - q * (1.0 / 56.0) * (1.0 - q * (1.0 / 90.0) * (1.0 - q / 132.0)))));
    s3 = (1.0 / 6.0) * (1.0 - q * (1.0 / 20.0) * (1.0 - q * (1.0 / 42.0) * (1.0 - q * (1.0 / 72.0) * (1.0 - q * (1.0 / 110.0) * (1.0 - q / 156.0)))));
  } else {
    double t = sqrt(q);
    double st = sin(t);
    double sh = sin(0.5 * t);
    s1 = st / t;
    s2 = 2.0 * sh * sh / q;
    s3 = (t - st) / (q * t);
  }
}

// The same three functions for the per-iteration update of the tolerance mode (solve_finish_fast), whose result is rounded to f32:
// the series cut after q^3 (the next term is < 3e-14 for q < 0.01, seven orders below an f32 ulp) with reciprocal constants — a
// dependent chain of 6 f64 multiply-adds per function where the full form has 10 and an f64 division (r05: the solve prologue is on
// the critical path of every launch of a schedule).
ELLC_HD void sinc_family_short(double q, double& s1, double& s2, double& s3) {
  if (q < 1e-2) {
    s1 = 1.0 - q * (1.0 / 6.0) * (1.0 - q * (1.0 / 20.0) * (1.0 - q * (1.0 / 42.0)));
    s2 = 0.5 * (1.0 - q * (1.0 / 12.0) * (1.0 - q * (1.0 / 30.0) * (1.0 - q * (1.0 / 56.0))));
    s3 = (1.0 / 6.0) * (1.0 - q * (1.0 / 20.0) * (1.0 - q * (1.0 / 42.0) * (1.0 - q * (1.0 / 72.0))));
  } else {
    sinc_family(q, s1, s2, s3);
  }
}

// One entry of exp: R[r][k] and the product V[r][k] * v[k] (so that t[r] = (p_r0 + p_r1) + p_r2), written so that each
// value is produced by exactly the operations exp_se3 uses for it — the solve kernel spreads the nine entries over
// nine lanes (r = lane / 3, k = lane % 3) instead of evaluating all of them on one.
template <bool SHORT = false>
ELLC_HD void exp_se3_entry(double a, double b, double c, double vx, double vy, double vz, int r, int k, double& Rrk, double& Vv) {
  const double q = a * a + b * b + c * c;
  double s1, s2, s3;
  if (SHORT) sinc_family_short(q, s1, s2, s3);
  else sinc_family(q, s1, s2, s3);
  const double wr = (r == 0) ? a : ((r == 1) ? b : c);
  const double wk = (k == 0) ? a : ((k == 1) ? b : c);
  const bool diag = (r == k);
  const int idx = 3 - r - k;                          // the remaining axis of an off-diagonal entry
  const double wi = (idx == 0) ? a : ((idx == 1) ? b : c);
  const bool pos = ((k - r + 3) % 3) == 2;            // sign of the skew part: (0,2) (1,0) (2,1) positive
  const double sw = pos ? wi : -wi;
  const double mm = wr * wk;
  const double m = diag ? (mm - q) : mm;              // [w]x^2 = w w^T - q I
  const double TR = diag ? 1.0 : (s1 * sw);
  Rrk = TR + s2 * m;
  const double TV = diag ? 1.0 : (s2 * sw);
  const double Vrk = TV + s3 * m;
  const double vk = (k == 0) ? vx : ((k == 1) ? vy : vz);
  Vv = Vrk * vk;
}

// exp: twist [w v] -> (R, t):  R = I + s1 [w]x + s2 [w]x^2,  t = V v with V = I + s2 [w]x + s3 [w]x^2
ELLC_HD void exp_se3(const double xi[6], Rt& o) {
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int r = 0; r < 3; r++) {
    double p[3];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 3; k++) exp_se3_entry(xi[0], xi[1], xi[2], xi[3], xi[4], xi[5], r, k, o.R[r * 3 + k], p[k]);
    o.t[r] = (p[0] + p[1]) + p[2];
  }
}

// log: (R, t) -> twist (principal branch, angle in [0, pi])
ELLC_HD void log_se3(const Rt& m, double xi[6]) {
  const double* R = m.R;
  const double cs = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  const double hx = 0.5 * (R[7] - R[5]), hy = 0.5 * (R[2] - R[6]), hz = 0.5 * (R[3] - R[1]);
  const double s2 = hx * hx + hy * hy + hz * hz;   // sin^2 of the rotation angle
  double a, b, c;
  if (cs > 0.9 && s2 < 1e-2) {
    // small rotation (the Gauss-Newton regime): angle/sin = asin(s)/s as a series in s^2 — no sqrt, no atan2.
    // Truncation < 2e-16 relative for s^2 < 0.01.
    const double k = 1.0 + s2 * (1.0 / 6.0 + s2 * (3.0 / 40.0 + s2 * (15.0 / 336.0 + s2 * (105.0 / 3456.0 + s2 * (945.0 / 42240.0 +
                     s2 * (10395.0 / 599040.0 + s2 * (135135.0 / 9676800.0)))))));
    a = hx * k; b = hy * k; c = hz * k;
  } else if (cs > -0.99) {
    const double sn = sqrt(s2);
    const double ang = atan2(sn, cs);
    const double k = (sn < 1e-8) ? 1.0 : ang / sn;
    a = hx * k; b = hy * k; c = hz * k;
  } else {
    const double sn = sqrt(s2);
    const double ang = atan2(sn, cs);
    // rotation close to pi: magnitude of the axis from the diagonal, signs from the skew / symmetric parts
    const double d = 1.0 - cs;
    double ux = sqrt(fmax(0.0, (R[0] - cs) / d)), uy = sqrt(fmax(0.0, (R[4] - cs) / d)), uz = sqrt(fmax(0.0, (R[8] - cs) / d));
    if (sn > 1e-12) {
      if (hx < 0) ux = -ux;
      if (hy < 0) uy = -uy;
      if (hz < 0) uz = -uz;
    } else {
      const double pxy = R[1] + R[3], pxz = R[2] + R[6], pyz = R[5] + R[7];
      if (ux >= uy && ux >= uz) { if (pxy < 0) uy = -uy; if (pxz < 0) uz = -uz; }
      else if (uy >= uz) { if (pxy < 0) ux = -ux; if (pyz < 0) uz = -uz; }
      else { if (pxz < 0) ux = -ux; if (pyz < 0) uy = -uy; }
    }
    const double n = sqrt(ux * ux + uy * uy + uz * uz);
    const double s = (n > 0) ? ang / n : 0.0;
    a = ux * s; b = uy * s; c = uz * s;
  }
  const double q = a * a + b * b + c * c;
  // V^-1 = I - 1/2 [w]x + g [w]x^2, g = (1 - s1/(2 s2)) / q
  double g;
  if (q < 1e-2) {
    g = (1.0 / 12.0) * (1.0 + q * (1.0 / 60.0) * (1.0 + q * (1.0 / 42.0) * (1.0 + q * (1.0 / 40.0) * (1.0 + q * (10.0 / 396.0)))));
  } else {
    double s1, s2, s3;
    sinc_family(q, s1, s2, s3);
    g = (1.0 - s1 / (2.0 * s2)) / q;
  }
  const double aa = a * a - q, bb = b * b - q, cc = c * c - q, ab = a * b, ac = a * c, bc = b * c;
  const double tx = m.t[0], ty = m.t[1], tz = m.t[2];
  xi[0] = a; xi[1] = b; xi[2] = c;
  xi[3] = (1.0 + g * aa) * tx + (0.5 * c + g * ab) * ty + (-0.5 * b + g * ac) * tz;
  xi[4] = (-0.5 * c + g * ab) * tx + (1.0 + g * bb) * ty + (0.5 * a + g * bc) * tz;
  xi[5] = (0.5 * b + g * ac) * tx + (-0.5 * a + g * bc) * ty + (1.0 + g * cc) * tz;
}

// f32 interface: pose6 -> 12 floats [r11 r12 r13 t1 | r21 r22 r23 t2 | r31 r32 r33 t3]
ELLC_HD void exp_se3_f32(const float pose[6], float S[12]) {
  double xi[6];
  for (int i = 0; i < 6; i++) xi[i] = (double)pose[i];
  Rt m;
  exp_se3(xi, m);
  for (int r = 0; r < 3; r++) {
    S[r * 4 + 0] = (float)m.R[r * 3 + 0];
    S[r * 4 + 1] = (float)m.R[r * 3 + 1];
    S[r * 4 + 2] = (float)m.R[r * 3 + 2];
    S[r * 4 + 3] = (float)m.t[r];
  }
}

ELLC_HD void load_f32(const float S[12], Rt& m) {
  for (int r = 0; r < 3; r++) {
    m.R[r * 3 + 0] = (double)S[r * 4 + 0];
    m.R[r * 3 + 1] = (double)S[r * 4 + 1];
    m.R[r * 3 + 2] = (double)S[r * 4 + 2];
    m.t[r] = (double)S[r * 4 + 3];
  }
}

// C = A * B on f32-held transforms; products summed in double, entries rounded once to f32
ELLC_HD void compose_f32(const float A[12], const float B[12], float C[12]) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += (double)A[r * 4 + k] * (double)B[k * 4 + c];
      C[r * 4 + c] = (float)s;
    }
    double s = 0.0;
    for (int k = 0; k < 3; k++) s += (double)A[r * 4 + k] * (double)B[k * 4 + 3];
    s += (double)A[r * 4 + 3];  // * 1 (homogeneous row of B)
    C[r * 4 + 3] = (float)s;
  }
}

ELLC_HD void invert_f32(const float A[12], float Ai[12]) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Ai[r * 4 + c] = A[c * 4 + r];
    double s = 0.0;
    for (int k = 0; k < 3; k++) s += (double)A[k * 4 + r] * (double)A[k * 4 + 3];
    Ai[r * 4 + 3] = (float)(-s);
  }
}

ELLC_HD void log_se3_f32(const float S[12], float pose[6]) {
  Rt m;
  load_f32(S, m);
  double xi[6];
  log_se3(m, xi);
  for (int i = 0; i < 6; i++) pose[i] = (float)xi[i];
}

// dest = log(exp(a) * exp(b))      (frame::concatenateRelativePose, Frame.cpp:503-530)
ELLC_HD void concat_relative_f32(const float a[6], const float b[6], float dest[6]) {
  float A[12], B[12], C[12];
  exp_se3_f32(a, A);
  exp_se3_f32(b, B);
  compose_f32(A, B, C);
  log_se3_f32(C, dest);
}

// dest = log(exp(a) * exp(b)^-1)   (frame::concatenateOriginPose, Frame.cpp:534-562)
ELLC_HD void concat_origin_f32(const float a[6], const float b[6], float dest[6]) {
  float A[12], B[12], Bi[12], C[12];
  exp_se3_f32(a, A);
  exp_se3_f32(b, B);
  invert_f32(B, Bi);
  compose_f32(A, Bi, C);
  log_se3_f32(C, dest);
}

}  // namespace ellc
