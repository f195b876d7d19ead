// ELLC ORACLE (test infrastructure) — semi-dense depth map: propagate / observe (line stereo) /
// fill holes / regularise / rescale / export. Follows DepthPropagation.cpp of the reference.
#include "ellc_oracle.hpp"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <limits>

namespace ellc_oracle {

// constants: ExternVariable.h (line numbers in comments)
static const float MIN_ABS_GRAD_CREATE = 1.0f;      // :81
static const float MIN_ABS_GRAD_DECREASE = 5.0f;    // :82
static const int MIN_BLACKLIST = -1;                // :83
static const float MAX_DIFF_CONSTANT = 40.0f * 40.0f;      // :85
static const float MAX_DIFF_GRAD_MULT = 0.5f * 0.5f;       // :86
static const float VAR_RANDOM_INIT_INITIAL = 0.125f;       // :88
static const float MIN_EPL_GRAD_SQUARED = 2.0f * 2.0f;     // :92
static const float MIN_EPL_LENGTH_SQUARED = 1.0f * 1.0f;   // :93
static const float MIN_EPL_ANGLE_SQUARED = 0.3f * 0.3f;    // :94
static const float MIN_DEPTH = 0.05f;                      // :98
static const float MAX_EPL_LENGTH_CROP = 30.0f;            // :101
static const float MIN_EPL_LENGTH_CROP = 3.0f;             // :102
static const float GRADIENT_SAMPLE_DIST = 1.0f;            // :105
static const float SAMPLE_POINT_TO_BORDER = 7.0f;          // :108
static const float MAX_ERROR_STEREO = 1300.0f;             // :111
static const float MIN_DISTANCE_ERROR_STEREO = 1.5f;       // :112
static const float STEREO_EPL_VAR_FAC = 2.0f;              // :115
static const float DIVISION_EPS = 1e-10f;                  // :117
static const int CAMERA_PIXEL_NOISE = 4 * 4;               // :120
static const int VALIDITY_COUNTER_INITIAL_OBSERVE = 5;     // :122
static const float SUCC_VAR_INC_FAC = 1.01f;               // :124
static const float FAIL_VAR_INC_FAC = 1.1f;                // :125
static const float MAX_VAR = 0.5f * 0.5f;                  // :126
static const float DIFF_FAC_OBSERVE = 1.0f * 1.0f;         // :130
static const float DIFF_FAC_PROP_MERGE = 1.0f * 1.0f;      // :131
static const float VALIDITY_COUNTER_MAX = 5.0f;            // :133
static const float VALIDITY_COUNTER_MAX_VARIABLE = 250.0f; // :134
static const float VALIDITY_COUNTER_DEC = 5.0f;            // :135
static const float VALIDITY_COUNTER_INC = 5.0f;            // :136
static const float VAL_SUM_MIN_FOR_CREATE = 30.0f;         // :141
static const float VAL_SUM_MIN_FOR_UNBLACKLIST = 100.0f;   // :142
static const float VAL_SUM_MIN_FOR_KEEP = 24.0f;           // :143
static const float REG_DIST_VAR = 0.075f * 0.075f * 1.0f * 1.0f;  // :145
static const float DIFF_FAC_SMOOTHING = 1.0f * 1.0f;       // :146

void DepthMap::init(const Config& c) {
  cfg = c;
  km = make_kmats(c);
  W = c.width;
  H = c.height;
  current.assign((size_t)W * H, Hyp());
  other.assign((size_t)W * H, Hyp());
  validityIntegralBuffer.assign((size_t)W * H, 0);  // DepthPropagation.cpp:27-28
  pyr.deptharr.resize(c.levels);
  pyr.depthvararr.resize(c.levels);
  for (int l = 0; l < c.levels; l++) {
    pyr.deptharr[l].assign((size_t)(W >> l) * (H >> l), 0.f);
    pyr.depthvararr[l].assign((size_t)(W >> l) * (H >> l), 0.f);
  }
}

static inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
// Eigen's Vector3f::dot (DepthPropagation.cpp:835-848) is a fixed-size REDUX, which Eigen 3.2 unrolls as a binary tree (Redux.h,
// redux_novec_unroller: func(first half, second half), HalfLength = Length / 2): for three terms e0 + (e1 + e2) — NOT the left-to-right
// sum of a coefficient-based matrix product (dot3 above). Found by the second source (tests/second_source_depth.py, r06); through the
// cancellation in (dot0 - oldX dot2) / nominator the two orders differ by up to ~1e-6 relative in the new inverse depth.
static inline float dot3_redux(const float* a, const float* b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

// DepthPropagation.cpp:1003-1157
void DepthMap::propagateDepth(Frame* nk) {
  for (auto& h : other) { h.isValid = 0; h.blacklisted = 0; }  // :1009-1014
  nk->calculateSE3poseOtherWrtThis(*keyFrame);                // :1020
  const float* T = nk->SE3poseThisWrtOther;                   // new <- old
  const float R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  const float t[3] = {T[3], T[7], T[11]};
  const float fxi = km.fx_inv, fyi = km.fy_inv, cxi = km.cx_inv, cyi = km.cy_inv;
  const PlaneU8& srcImg = keyFrame->image_pyramid[0];
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const Hyp* source = &current[(size_t)x + (size_t)y * W];
      if (!source->isValid) continue;
      float kp[3] = {x * fxi + cxi, y * fyi + cyi, 1.0f};
      float pn[3];
      for (int r = 0; r < 3; r++) {
        float s = (R[r * 3 + 0] * kp[0] + R[r * 3 + 1] * kp[1]) + R[r * 3 + 2] * kp[2];
        pn[r] = s / source->invDepthSmoothed + t[r];
      }
      float new_idepth = 1.0f / pn[2];
      float u_new = pn[0] * new_idepth * cfg.fx + cfg.cx;
      float v_new = pn[1] * new_idepth * cfg.fy + cfg.cy;
      if (!(u_new > 2.1f && v_new > 2.1f && u_new < W - 3.1f && v_new < H - 3.1f)) continue;  // :1059
      int newIDX = (int)(u_new + 0.5f) + ((int)(v_new + 0.5f)) * W;
      float destAbsGrad = nk->maxAbsGradient.at(y, x);  // source coordinates (Q14)
      float sourceColor = srcImg.at(y, x);
      float destColor = nk->getInterpolatedElement(u_new, v_new);
      float residual = destColor - sourceColor;
      if (residual * residual / (MAX_DIFF_CONSTANT + MAX_DIFF_GRAD_MULT * destAbsGrad * destAbsGrad) > 1.0f ||
          destAbsGrad < MIN_ABS_GRAD_DECREASE)
        continue;
      Hyp* targetBest = &other[newIDX];
      float r4 = new_idepth / source->invDepthSmoothed;
      r4 *= r4;
      r4 *= r4;
      float new_var = r4 * source->invDepth;  // sic (Q14)
      if (targetBest->isValid) {
        float diff = targetBest->invDepth - new_idepth;
        if (DIFF_FAC_PROP_MERGE * diff * diff > new_var + targetBest->variance) {
          if (new_idepth < targetBest->invDepth) continue;
          else targetBest->isValid = 0;
        }
      }
      if (!targetBest->isValid) {
        targetBest->invDepth = new_idepth;
        targetBest->variance = new_var;
        targetBest->varianceSmoothed = -1;
        targetBest->invDepthSmoothed = -1;
        targetBest->validity_counter = source->validity_counter;
        targetBest->isValid = 1;
        targetBest->blacklisted = 0;
      } else {
        float w = new_var / (targetBest->variance + new_var);
        float merged_new_idepth = w * targetBest->invDepth + (1.0f - w) * new_idepth;
        int merged_validity = source->validity_counter + targetBest->validity_counter;
        if (merged_validity > VALIDITY_COUNTER_MAX + (VALIDITY_COUNTER_MAX_VARIABLE))
          merged_validity = (int)(VALIDITY_COUNTER_MAX + (VALIDITY_COUNTER_MAX_VARIABLE));
        float temp_var = targetBest->variance;
        targetBest->invDepth = merged_new_idepth;
        targetBest->variance = 1.0f / (1.0f / temp_var + 1.0f / new_var);
        targetBest->validity_counter = merged_validity;
        targetBest->isValid = 1;
        targetBest->blacklisted = 0;
        targetBest->invDepthSmoothed = -1.0f;
        targetBest->varianceSmoothed = -1.0f;
      }
    }
  std::swap(current, other);  // :1154
}

// DepthPropagation.cpp:311-384
bool DepthMap::makeAndCheckEPL(int x, int y, float* pepx, float* pepy) {
  const float* To = currentFrame->SE3poseOtherWrtThis;
  const float t0 = To[3], t1 = To[7], t2 = To[11];
  float epx = -cfg.fx * t0 + t2 * (x - cfg.cx);
  float epy = -cfg.fy * t1 + t2 * (y - cfg.cy);
  if (std::isnan(epx + epy)) return false;
  float eplLengthSquared = epx * epx + epy * epy;
  if (eplLengthSquared < MIN_EPL_LENGTH_SQUARED) return false;
  const PlaneU8& img = keyFrame->image_pyramid[0];
  float gx = img.at(y, x + 1) - img.at(y, x - 1);
  float gy = img.at(y + 1, x) - img.at(y - 1, x);
  float eplGradSquared = gx * epx + gy * epy;
  eplGradSquared = eplGradSquared * eplGradSquared / eplLengthSquared;
  if (eplGradSquared < MIN_EPL_GRAD_SQUARED) return false;
  if (eplGradSquared / (gx * gx + gy * gy) < MIN_EPL_ANGLE_SQUARED) return false;
  float fac = GRADIENT_SAMPLE_DIST / std::sqrt(eplLengthSquared);
  *pepx = epx * fac;
  *pepy = epy * fac;
  return true;
}

// DepthPropagation.cpp:397-885
float DepthMap::doLineStereo(float u, float v, float epxn, float epyn, float min_idepth, float prior_idepth,
                             float max_idepth, float& result_idepth, float& result_var, float& result_eplLength) {
  const float* Kr = currentFrame->K_SE3poseThisWrtOther_r;
  const float* Kt = currentFrame->K_SE3poseThisWrtOther_t;
  const float* Tt = currentFrame->SE3poseThisWrtOther;
  const float Rr[9] = {Tt[0], Tt[1], Tt[2], Tt[4], Tt[5], Tt[6], Tt[8], Tt[9], Tt[10]};
  const float tt[3] = {Tt[3], Tt[7], Tt[11]};
  const float NaN = std::numeric_limits<float>::quiet_NaN();

  float KinvP[3] = {km.fx_inv * u + km.cx_inv, km.fy_inv * v + km.cy_inv, 1.0f};
  float pInf[3] = {dot3(Kr, KinvP), dot3(Kr + 3, KinvP), dot3(Kr + 6, KinvP)};
  float pReal[3] = {pInf[0] / prior_idepth + Kt[0], pInf[1] / prior_idepth + Kt[1], pInf[2] / prior_idepth + Kt[2]};
  float rescaleFactor = pReal[2] * prior_idepth;

  float firstX = u - 2 * epxn * rescaleFactor;
  float firstY = v - 2 * epyn * rescaleFactor;
  float lastX = u + 2 * epxn * rescaleFactor;
  float lastY = v + 2 * epyn * rescaleFactor;
  if (firstX <= 0 || firstX >= W - 2 || firstY <= 0 || firstY >= H - 2 || lastX <= 0 || lastX >= W - 2 || lastY <= 0 ||
      lastY >= H - 2)
    return -1;
  if (!(rescaleFactor > 0.7f && rescaleFactor < 1.4f)) return -1;

  float realVal_p1 = keyFrame->getInterpolatedElement(u + epxn * rescaleFactor, v + epyn * rescaleFactor);
  float realVal_m1 = keyFrame->getInterpolatedElement(u - epxn * rescaleFactor, v - epyn * rescaleFactor);
  float realVal = keyFrame->getInterpolatedElement(u, v);
  float realVal_m2 = keyFrame->getInterpolatedElement(u - 2 * epxn * rescaleFactor, v - 2 * epyn * rescaleFactor);
  float realVal_p2 = keyFrame->getInterpolatedElement(u + 2 * epxn * rescaleFactor, v + 2 * epyn * rescaleFactor);

  float pClose[3] = {pInf[0] + Kt[0] * max_idepth, pInf[1] + Kt[1] * max_idepth, pInf[2] + Kt[2] * max_idepth};
  if (pClose[2] < 0.001f) {
    max_idepth = (0.001f - pInf[2]) / Kt[2];
    for (int i = 0; i < 3; i++) pClose[i] = pInf[i] + Kt[i] * max_idepth;
  }
  { float z = pClose[2]; for (int i = 0; i < 3; i++) pClose[i] = pClose[i] / z; }
  float pFar[3] = {pInf[0] + Kt[0] * min_idepth, pInf[1] + Kt[1] * min_idepth, pInf[2] + Kt[2] * min_idepth};
  if (pFar[2] < 0.001f || max_idepth < min_idepth) return -1;
  { float z = pFar[2]; for (int i = 0; i < 3; i++) pFar[i] = pFar[i] / z; }
  if (std::isnan((float)(pFar[0] + pClose[0]))) return -4;

  float incx = pClose[0] - pFar[0];
  float incy = pClose[1] - pFar[1];
  float eplLength = std::sqrt(incx * incx + incy * incy);
  if ((!eplLength) > 0 || std::isinf(eplLength)) return -4;  // sic: (!eplLength) > 0
  if (eplLength > MAX_EPL_LENGTH_CROP) {
    pClose[0] = pFar[0] + incx * MAX_EPL_LENGTH_CROP / eplLength;
    pClose[1] = pFar[1] + incy * MAX_EPL_LENGTH_CROP / eplLength;
  }
  incx *= GRADIENT_SAMPLE_DIST / eplLength;
  incy *= GRADIENT_SAMPLE_DIST / eplLength;
  pFar[0] -= incx; pFar[1] -= incy;
  pClose[0] += incx; pClose[1] += incy;
  if (eplLength < MIN_EPL_LENGTH_CROP) {
    float pad = (MIN_EPL_LENGTH_CROP - (eplLength)) / 2.0f;
    pFar[0] -= incx * pad; pFar[1] -= incy * pad;
    pClose[0] += incx * pad; pClose[1] += incy * pad;
  }
  const float B = SAMPLE_POINT_TO_BORDER;
  if (pFar[0] <= B || pFar[0] >= W - B || pFar[1] <= B || pFar[1] >= H - B) return -1;
  if (pClose[0] <= B || pClose[0] >= W - B || pClose[1] <= B || pClose[1] >= H - B) {
    if (pClose[0] <= B) {
      float toAdd = (B - pClose[0]) / incx;
      pClose[0] += toAdd * incx; pClose[1] += toAdd * incy;
    } else if (pClose[0] >= W - B) {
      float toAdd = (W - B - pClose[0]) / incx;
      pClose[0] += toAdd * incx; pClose[1] += toAdd * incy;
    }
    if (pClose[1] <= B) {
      float toAdd = (B - pClose[1]) / incy;
      pClose[0] += toAdd * incx; pClose[1] += toAdd * incy;
    } else if (pClose[1] >= H - B) {
      float toAdd = (H - B - pClose[1]) / incy;
      pClose[0] += toAdd * incx; pClose[1] += toAdd * incy;
    }
    float fincx = pClose[0] - pFar[0];
    float fincy = pClose[1] - pFar[1];
    float newEplLength = std::sqrt(fincx * fincx + fincy * fincy);
    if (pClose[0] <= B || pClose[0] >= W - B || pClose[1] <= B || pClose[1] >= H - B || newEplLength < 8.0f) return -1;
  }

  float cpx = pFar[0], cpy = pFar[1];
  float val_cp_m2 = currentFrame->getInterpolatedElement(cpx - 2.0f * incx, cpy - 2.0f * incy);
  float val_cp_m1 = currentFrame->getInterpolatedElement(cpx - incx, cpy - incy);
  float val_cp = currentFrame->getInterpolatedElement(cpx, cpy);
  float val_cp_p1 = currentFrame->getInterpolatedElement(cpx + incx, cpy + incy);
  float val_cp_p2;

  int loopCounter = 0;
  float best_match_x = -1, best_match_y = -1;
  float best_match_err = std::numeric_limits<float>::infinity();         // float = 1e50
  float second_best_match_err = std::numeric_limits<float>::infinity();
  float best_match_errPre = NaN, best_match_errPost = NaN, best_match_DiffErrPre = NaN, best_match_DiffErrPost = NaN;
  bool bestWasLastLoop = false;
  float eeLast = -1;
  float e1A = NaN, e1B = NaN, e2A = NaN, e2B = NaN, e3A = NaN, e3B = NaN, e4A = NaN, e4B = NaN, e5A = NaN, e5B = NaN;
  int loopCBest = -1, loopCSecond = -1;
  while (((incx < 0) == (cpx > pClose[0]) && (incy < 0) == (cpy > pClose[1])) || loopCounter == 0) {
    val_cp_p2 = currentFrame->getInterpolatedElement(cpx + 2 * incx, cpy + 2 * incy);
    float ee = 0;
    if (loopCounter % 2 == 0) {
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
    loopCounter++;
  }
  if (best_match_err > 4.0f * (float)MAX_ERROR_STEREO) return -3;
  if (std::abs(loopCBest - loopCSecond) > 1.0f && MIN_DISTANCE_ERROR_STEREO * best_match_err > second_best_match_err) return -2;

  bool didSubpixel = false;
  {
    float gradPre_pre = -(best_match_errPre - best_match_DiffErrPre);
    float gradPre_this = +(best_match_err - best_match_DiffErrPre);
    float gradPost_this = -(best_match_err - best_match_DiffErrPost);
    float gradPost_post = +(best_match_errPost - best_match_DiffErrPost);
    bool interpPost = false, interpPre = false;
    if (best_match_errPre < 0 || best_match_errPost < 0) {
    } else if ((gradPre_pre < 0) ^ (gradPre_this < 0)) {
      if ((gradPost_post < 0) ^ (gradPost_this < 0)) {
      } else interpPre = true;
    } else if ((gradPost_post < 0) ^ (gradPost_this < 0)) {
      interpPost = true;
    }
    if (interpPre) {
      float d = gradPre_this / (gradPre_this - gradPre_pre);
      best_match_x -= d * incx;
      best_match_y -= d * incy;
      best_match_err = best_match_err - 2 * d * gradPre_this - (gradPre_pre - gradPre_this) * d * d;
      didSubpixel = true;
    } else if (interpPost) {
      float d = gradPost_this / (gradPost_this - gradPost_post);
      best_match_x += d * incx;
      best_match_y += d * incy;
      best_match_err = best_match_err + 2 * d * gradPost_this + (gradPost_post - gradPost_this) * d * d;
      didSubpixel = true;
    }
  }
  float sampleDist = GRADIENT_SAMPLE_DIST * rescaleFactor;
  float gradAlongLine = 0;
  float tmp = realVal_p2 - realVal_p1; gradAlongLine += tmp * tmp;
  tmp = realVal_p1 - realVal; gradAlongLine += tmp * tmp;
  tmp = realVal - realVal_m1; gradAlongLine += tmp * tmp;
  tmp = realVal_m1 - realVal_m2; gradAlongLine += tmp * tmp;
  gradAlongLine /= sampleDist * sampleDist;
  if (best_match_err > (float)MAX_ERROR_STEREO + sqrtf(gradAlongLine) * 20) return -3;

  float idnew_best_match, alpha;
  if (incx * incx > incy * incy) {
    float oldX = km.fx_inv * best_match_x + km.cx_inv;
    float nominator = (oldX * tt[2] - tt[0]);
    float dot0 = dot3_redux(KinvP, Rr);
    float dot2 = dot3_redux(KinvP, Rr + 6);
    idnew_best_match = (dot0 - oldX * dot2) / nominator;
    alpha = incx * km.fx_inv * (dot0 * tt[2] - dot2 * tt[0]) / (nominator * nominator);
  } else {
    float oldY = km.fy_inv * best_match_y + km.cy_inv;
    float nominator = (oldY * tt[2] - tt[1]);
    float dot1 = dot3_redux(KinvP, Rr + 3);
    float dot2 = dot3_redux(KinvP, Rr + 6);
    idnew_best_match = (dot1 - oldY * dot2) / nominator;
    alpha = incy * km.fx_inv * (dot1 * tt[2] - dot2 * tt[1]) / (nominator * nominator);  // FX_INV (Q19)
  }
  if (idnew_best_match < 0) return -2;
  float photoDispError = 4.0f * CAMERA_PIXEL_NOISE / (gradAlongLine + DIVISION_EPS);
  float trackingErrorFac = 0.25f * 1.0f;
  float g0 = keyFrame->getInterpolatedGradX(u, v);
  float g1 = keyFrame->getInterpolatedGradY(u, v);
  float geoDispError = (g0 * epxn + g1 * epyn) + DIVISION_EPS;
  geoDispError = trackingErrorFac * trackingErrorFac * (g0 * g0 + g1 * g1) / (geoDispError * geoDispError);
  result_var = alpha * alpha * ((didSubpixel ? 0.05f : 0.5f) * sampleDist * sampleDist + geoDispError + photoDispError);
  result_idepth = idnew_best_match;
  result_eplLength = eplLength;
  return best_match_err;
}

// DepthPropagation.cpp:267-308
int DepthMap::observeDepthCreate(int x, int y, int idx) {
  Hyp* target = &current[idx];
  float epx, epy;
  if (!makeAndCheckEPL(x, y, &epx, &epy)) return -1;
  float result_idepth = 0, result_var = 0, result_eplLength = 0;
  float error = doLineStereo((float)x, (float)y, epx, epy, 0.0f, 1.0f, 1.0f / MIN_DEPTH, result_idepth, result_var, result_eplLength);
  if (error == -3 || error == -2) target->blacklisted--;
  if (error < 0 || result_var > MAX_VAR) return -2;
  result_idepth = unzero(result_idepth);
  target->invDepth = result_idepth;
  target->variance = result_var;
  target->invDepthSmoothed = -1;
  target->varianceSmoothed = -1;
  target->validity_counter = VALIDITY_COUNTER_INITIAL_OBSERVE;
  target->isValid = 1;
  target->blacklisted = 0;
  return 1;
}

// DepthPropagation.cpp:888-999
int DepthMap::observeDepthUpdate(int x, int y, int idx) {
  Hyp* target = &current[idx];
  float epx, epy;
  if (!makeAndCheckEPL(x, y, &epx, &epy)) return -5;
  float sv = std::sqrt(target->varianceSmoothed);
  float min_idepth = target->invDepthSmoothed - sv * STEREO_EPL_VAR_FAC;
  float max_idepth = target->invDepthSmoothed + sv * STEREO_EPL_VAR_FAC;
  if (min_idepth < 0) min_idepth = 0;
  if (max_idepth > 1 / MIN_DEPTH) max_idepth = 1 / MIN_DEPTH;
  float result_idepth = 0, result_var = 0, result_eplLength = 0;
  float error = doLineStereo((float)x, (float)y, epx, epy, min_idepth, target->invDepthSmoothed, max_idepth, result_idepth,
                             result_var, result_eplLength);
  float diff = result_idepth - target->invDepthSmoothed;
  if (error == -1) return -1;
  else if (error == -2) {
    target->validity_counter -= VALIDITY_COUNTER_DEC;
    if (target->validity_counter < 0) target->validity_counter = 0;
    target->variance *= FAIL_VAR_INC_FAC;
    if (target->variance > MAX_VAR) {
      target->isValid = 0;
      target->blacklisted--;
    }
    return -2;
  } else if (error == -3) return -3;
  else if (error == -4) return -4;
  else if (DIFF_FAC_OBSERVE * diff * diff > result_var + target->varianceSmoothed) {
    target->variance *= FAIL_VAR_INC_FAC;
    if (target->variance > MAX_VAR) target->isValid = 0;
    return -6;
  } else {
    float id_var = target->variance * SUCC_VAR_INC_FAC;
    float w = result_var / (result_var + id_var);
    float new_idepth = (1 - w) * result_idepth + w * target->invDepth;
    target->invDepth = unzero(new_idepth);
    id_var = id_var * w;
    if (id_var < target->variance) target->variance = id_var;
    target->validity_counter += VALIDITY_COUNTER_INC;
    float absGrad = keyFrame->maxAbsGradient.at(y, x);
    if (target->validity_counter > VALIDITY_COUNTER_MAX + absGrad * (VALIDITY_COUNTER_MAX_VARIABLE) / 255.0f)
      target->validity_counter = (int)(VALIDITY_COUNTER_MAX + absGrad * (VALIDITY_COUNTER_MAX_VARIABLE) / 255.0f);
    return 1;
  }
}

// DepthPropagation.cpp:191-263
void DepthMap::observeDepthRow(int ymin, int ymax) {
  for (int y = ymin; y < ymax; y++)
    for (int x = 3; x < W - 3; x++) {
      int idx = x + y * W;
      Hyp* target = &current[idx];
      bool hasHypothesis = target->isValid;
      float mg = keyFrame->maxAbsGradient.at(y, x);
      if (hasHypothesis && mg < MIN_ABS_GRAD_DECREASE) {
        target->isValid = 0;
        continue;
      }
      if (mg < MIN_ABS_GRAD_CREATE || target->blacklisted < MIN_BLACKLIST) continue;
      if (!hasHypothesis) observeDepthCreate(x, y, idx);
      else observeDepthUpdate(x, y, idx);
    }
}

// DepthPropagation.cpp:1932-1958 (bands only split independent pixels)
void DepthMap::observeDepthRowParallel() {
  currentFrame->calculateSE3poseOtherWrtThis(*keyFrame);
  observeDepthRow(3, H - 3);
}

// DepthPropagation.cpp:1403-1432
void DepthMap::buildValIntegralBuffer() {
  for (int y = 3; y < H - 3; y++) {
    int sum = 0;
    for (int x = 0; x < W; x++) {
      const Hyp& s = current[(size_t)x + (size_t)y * W];
      if (s.isValid) sum += s.validity_counter;
      validityIntegralBuffer[(size_t)x + (size_t)y * W] = sum;
    }
  }
}

// DepthPropagation.cpp:1317-1400
void DepthMap::fillDepthHoles() {
  buildValIntegralBuffer();
  other = current;  // memcpy :1322
  for (int y = 3; y < H - 3; y++)
    for (int x = 3; x < W - 2; x++) {
      int idx = x + y * W;
      const Hyp* dest = &other[idx];
      if (dest->isValid) continue;
      if (keyFrame->maxAbsGradient.at(y, x) < MIN_ABS_GRAD_DECREASE) continue;
      const int* io = &validityIntegralBuffer[idx];
      int val = io[2 + 2 * W] - io[2 - 3 * W] - io[-3 + 2 * W] + io[-3 - 3 * W];
      if ((dest->blacklisted >= MIN_BLACKLIST && val > VAL_SUM_MIN_FOR_CREATE) || val > VAL_SUM_MIN_FOR_UNBLACKLIST) {
        float sumIdepthObs = 0, sumIVarObs = 0;
        for (int yy = y - 2; yy < y + 3; yy++)
          for (int xx = x - 2; xx < x + 3; xx++) {
            const Hyp* s = &other[(size_t)xx + (size_t)yy * W];
            if (!s->isValid) continue;
            sumIdepthObs += s->invDepth / s->variance;
            sumIVarObs += 1.0f / s->variance;
          }
        float idepthObs = sumIdepthObs / sumIVarObs;
        idepthObs = unzero(idepthObs);
        Hyp& c = current[idx];
        c.invDepth = idepthObs;
        c.variance = VAR_RANDOM_INIT_INITIAL;
        c.validity_counter = 0;
        c.isValid = 1;
        c.blacklisted = 0;
        c.invDepthSmoothed = -1;
        c.varianceSmoothed = -1;
      }
    }
}

// DepthPropagation.cpp:1436-1543
void DepthMap::regularizeDepthMap(bool removeOcclusions) {
  other = current;  // memcpy :1438
  const int validityTH = (int)VAL_SUM_MIN_FOR_KEEP;
  const int R = 2;
  const float regDistVar = REG_DIST_VAR;
  for (int y = 3; y < H - 3; y++)
    for (int x = R; x < W - R; x++) {
      Hyp* dest = &current[(size_t)x + (size_t)y * W];
      const Hyp* destRead = &other[(size_t)x + (size_t)y * W];
      if (!destRead->isValid) continue;
      float sum = 0, val_sum = 0, sumIvar = 0;
      int numOccluding = 0, numNotOccluding = 0;
      for (int dx = -R; dx <= R; dx++)
        for (int dy = -R; dy <= R; dy++) {
          const Hyp* source = destRead + dx + dy * W;
          if (!source->isValid) continue;
          float diff = source->invDepth - destRead->invDepth;
          if (DIFF_FAC_SMOOTHING * diff * diff > source->variance + destRead->variance) {
            if (removeOcclusions)
              if (source->invDepth > destRead->invDepth) numOccluding++;
            continue;
          }
          val_sum += source->validity_counter;
          if (removeOcclusions) numNotOccluding++;
          float distFac = (float)(dx * dx + dy * dy) * regDistVar;
          float ivar = 1.0f / (source->variance + distFac);
          sum += source->invDepth * ivar;
          sumIvar += ivar;
        }
      if (val_sum < validityTH) {
        dest->isValid = 0;
        dest->blacklisted--;
        continue;
      }
      if (removeOcclusions)
        if (numOccluding > numNotOccluding) {
          dest->isValid = 0;
          continue;
        }
      sum = sum / sumIvar;
      sum = unzero(sum);
      dest->invDepthSmoothed = sum;
      dest->varianceSmoothed = 1.0f / sumIvar;
    }
}

// DepthPropagation.cpp:1546-1587 (calculate_on_current branch)
void DepthMap::makeInvDepthOne() {
  float sumIdepth = 0, numIdepth = 0;
  for (const Hyp& s : current) {
    if (!s.isValid) continue;
    sumIdepth += s.invDepthSmoothed;
    numIdepth++;
  }
  float rescaleFactor = numIdepth / sumIdepth;
  depthScale = rescaleFactor;
  keyFrame->rescaleFactor = rescaleFactor;
  global_depth_scale *= depthScale;
  float rescaleFactor2 = rescaleFactor * rescaleFactor;
  for (Hyp& s : current) {
    if (!s.isValid) continue;
    s.invDepth *= rescaleFactor;
    s.invDepthSmoothed *= rescaleFactor;
    s.variance *= rescaleFactor2;
    s.varianceSmoothed *= rescaleFactor2;
  }
}

// DepthPropagation.cpp:1254-1315
void DepthMap::updateDepthImage() {
  PlaneF& depth = keyFrame->depth_pyramid[0];  // keyFrame->depth aliases depth_pyramid[0] (:1728)
  float* deptharr = pyr.deptharr[0].data();
  float* vararr = pyr.depthvararr[0].data();
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      Hyp* pt = &current[(size_t)x + (size_t)y * W];
      if (y < 3 || y >= H - 3 || x < 3 || x >= W - 3) pt->isValid = 0;
      size_t i = (size_t)x + (size_t)y * W;
      if (pt->isValid && (pt->invDepthSmoothed >= -0.05f)) {
        depth.at(y, x) = (1 / (pt->invDepthSmoothed));
        deptharr[i] = (1 / (pt->invDepthSmoothed));
        vararr[i] = pt->varianceSmoothed;
      } else {
        depth.at(y, x) = 0.0f;
        deptharr[i] = -1.0f;
        vararr[i] = -1.0f;
      }
    }
  buildInvVarDepth();
  mapDepthArr2Mat();
}

// DepthPropagation.cpp:1637-1719
void DepthMap::buildInvVarDepth() {
  for (int i = 1; i < cfg.levels; i++) {
    int width = W >> i, height = H >> i;
    int sw = 2 * width;
    // NOTE: the reference indexes the source with stride 2*width, which equals the source width only
    // when (W >> (i-1)) is even; kept as written.
    const float* vs = pyr.depthvararr[i - 1].data();
    float* vd = pyr.depthvararr[i].data();
    const float* ds = pyr.deptharr[i - 1].data();
    float* dd = pyr.deptharr[i].data();
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++) {
        int idx = 2 * (x + y * sw);
        int idxDest = x + y * width;
        float idepthSumsSum = 0, ivarSumsSum = 0;
        int num = 0;
        const int offs[4] = {0, 1, sw, sw + 1};
        for (int q = 0; q < 4; q++) {
          float var = vs[idx + offs[q]];
          if (var > 0) {
            float ivar = 1.0f / var;
            ivarSumsSum += ivar;
            idepthSumsSum += ivar * 1.0f / ds[idx + offs[q]];
            num++;
          }
        }
        if (num > 0) {
          dd[idxDest] = ivarSumsSum / idepthSumsSum;
          vd[idxDest] = num / ivarSumsSum;
        } else {
          dd[idxDest] = 0.0f;
          vd[idxDest] = -1.0f;
        }
      }
  }
}

// DepthPropagation.cpp:1722-1746
void DepthMap::mapDepthArr2Mat() {
  for (int i = 1; i < cfg.levels; i++) {
    PlaneF& p = keyFrame->depth_pyramid[i];
    const float* s = pyr.deptharr[i].data();
    for (int y = 0; y < p.h; y++)
      for (int x = 0; x < p.w; x++) p.at(y, x) = s[x + y * p.w];
  }
}

// DepthPropagation.cpp:1804-1830
float DepthMap::calculate_no_of_Seeds() const {
  float count = 0;
  for (const Hyp& h : current) count += float(h.isValid);
  return count / (W * H) * 100;
}

// DepthPropagation.cpp:1758-1794
void DepthMap::createKeyFrame(Frame* nk) {
  nk->calculateSE3poseOtherWrtThis(*keyFrame);
  propagateDepth(nk);
  keyFrame = nk;
  regularizeDepthMap(true);
  doRegularization(false);
  makeInvDepthOne();
  updateDepthImage();
  for (int i = 0; i < 6; i++) keyFrame->poseWrtOrigin[i] = 0.0f;
}

}  // namespace ellc_oracle
