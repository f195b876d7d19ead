// ELLC ORACLE (test infrastructure) — pixel-wise Gauss-Newton (FCA + constant-weight ICA) and its driver.
// Follows PixelWisePyramid.cpp and ImageFunc.cpp of the reference; see ellc_oracle.hpp.
#include "ellc_oracle.hpp"
#include <cmath>
#include <cstring>
#include <thread>
#include <algorithm>

#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>

namespace ellc_oracle {

// CPU-baseline variant only (thread_mode 2): T worker threads that stay alive between iterations and run the row bands of
// every iteration, instead of the reference's create/join of NUM_POSE_THREADS threads per iteration
// (PixelWisePyramid.cpp:424-436). Same bands, same sums; only the thread management differs.
namespace {
class BandPool {
 public:
  explicit BandPool(int T) : T_(T) {
    for (int t = 0; t < T; t++) workers_.emplace_back([this, t] { loop(t); });
  }
  ~BandPool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  void run(const std::function<void(int)>& f) {
    std::unique_lock<std::mutex> l(m_);
    job_ = &f;
    left_ = T_;
    gen_++;
    cv_.notify_all();
    done_.wait(l, [this] { return left_ == 0; });
    job_ = nullptr;
  }
  static BandPool& get(int T) {
    static std::mutex pm;
    static std::map<int, std::unique_ptr<BandPool>> pools;
    std::lock_guard<std::mutex> l(pm);
    auto& p = pools[T];
    if (!p) p.reset(new BandPool(T));
    return *p;
  }

 private:
  void loop(int t) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int)>* job;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        job = job_;
      }
      (*job)(t);
      {
        std::lock_guard<std::mutex> l(m_);
        if (--left_ == 0) done_.notify_one();
      }
    }
  }
  int T_;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* job_ = nullptr;
  int left_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};
}  // namespace

PixelWisePyramid::PixelWisePyramid(Frame* prev, Frame* cur, float* pose_, const DepthPyr* dm)
    : prev_frame(prev), current_frame(cur), depthMap(dm), pose(pose_) {
  pyrlevel = prev->pyrLevel;
  nRows = prev->currentRows;
  nCols = prev->currentCols;
  display_weightimg = PlaneF(nCols, nRows, 0.f);
  display_iterationres = PlaneF(nCols, nRows, 0.f);
  savedWarpedPointsX = PlaneF(nCols, nRows, 0.f);
  savedWarpedPointsY = PlaneF(nCols, nRows, 0.f);
  std::memset(hessian, 0, sizeof(hessian));
  std::memset(sd_param, 0, sizeof(sd_param));
  std::memset(hessianInv, 0, sizeof(hessianInv));
  std::memset(deltapose, 0, sizeof(deltapose));
}

namespace {
// Per-pixel 1x6 steepest-descent row, PixelWisePyramid.cpp:296-320 (and :637-662). pow(.,2) and
// pow(.,-1) take (float,int) and therefore evaluate in double (Q2); everything else is f32.
inline void jacobian_row(float gradx, float grady, int x, int y, float Z, const Intrin& k, float J[6]) {
  const float fx = k.fx, fy = k.fy, cx = k.cx, cy = k.cy;
  const float u = -cx + x;
  const float v = -cy + y;
  float jb[6], jt[6];
  jb[0] = (float)(grady * (-(fy + (std::pow((double)v, 2) / fy))));
  jt[0] = gradx * (-(v * u) / fy);
  jb[1] = grady * ((v * u) / fx);
  jt[1] = (float)(gradx * (fx + (std::pow((double)u, 2) / fx)));
  jb[2] = grady * ((fy * u) / fx);
  jt[2] = gradx * (-(fx * v / fy));
  const double invZ = 1.0 / (double)Z;  // pow(depth,-1) in double
  jb[3] = 0;
  jt[3] = (float)(gradx * (fx * invZ));
  jb[4] = (float)(grady * (fy * invZ));
  jt[4] = 0;
  jb[5] = (float)(grady * (-v * invZ));
  jt[5] = (float)(gradx * (-u * invZ));
  for (int i = 0; i < 6; i++) J[i] = jt[i] + jb[i];
}

struct WarpOut { float px, py, pz, wx, wy; };
// PixelWisePyramid.cpp:236-262 (both branches of the SE3_vec[1]==0 test evaluate the same f32 expression)
inline WarpOut warp_point(int x, int y, float Z, const Intrin& k, const float* S) {
  float X = (x - k.cx) * Z / k.fx;
  float Y = (y - k.cy) * Z / k.fy;
  WarpOut o;
  o.px = (S[0] * X) + (S[1] * Y) + (S[2] * Z) + (S[3]);
  o.py = (S[4] * X) + (S[5] * Y) + (S[6] * Z) + (S[7]);
  o.pz = (S[8] * X) + (S[9] * Y) + (S[10] * Z) + (S[11]);
  o.pz = unzero(o.pz);
  o.wx = ((o.px / o.pz) * k.fx) + k.cx;
  o.wy = ((o.py / o.pz) * k.fy) + k.cy;
  return o;
}
}  // namespace

// PixelWisePyramid.cpp:58-413 — one row band. H/b are the band's f32 partial sums (raster order);
// Hd/bd the same terms summed in double (SUM_F64 mode, diagnostics only).
void PixelWisePyramid::calculatePixelWise(int ymin, int ymax, float H_out[36], float b_out[6], double Hd_out[36], double bd_out[6]) {
  const Intrin k = get_intrinsic(prev_frame->cfg, prev_frame->pyrLevel);
  // The band's sums live on this thread's stack and are stored once at the end (same additions in the same order). The
  // reference gives every band separately allocated cv::Mat accumulators (PixelWisePyramid.cpp:370-404); accumulating
  // through pointers into one contiguous vector shared by the bands would put several bands' sums into one cache line
  // (false sharing) and make the multithreaded CPU baseline slower than one thread.
  float H[36], b[6];
  double Hd[36], bd[6];
  for (int i = 0; i < 36; i++) { H[i] = 0; Hd[i] = 0; }
  for (int i = 0; i < 6; i++) { b[i] = 0; bd[i] = 0; }
  float SE3[16];
  se3_exp(pose, SE3);
  const float* S = SE3;  // r11 r12 r13 t1 | r21 .. t2 | r31 .. t3
  const float tx = SE3[3], ty = SE3[7], tz = SE3[11];
  const PlaneU8& img = current_frame->image_pyramid[pyrlevel];
  const PlaneU8& pimg = prev_frame->image_pyramid[pyrlevel];
  const PlaneF& depth = prev_frame->depth_pyramid[pyrlevel];
  const std::vector<float>& var = depthMap->depthvararr[prev_frame->pyrLevel];
  (void)img;
  for (int y = ymin; y < ymax; y++) {
    for (int x = 0; x < nCols; x++) {
      if (prev_frame->mask.at(y, x) == 0) {
        display_iterationres.at(y, x) = 0;
        display_weightimg.at(y, x) = 0;
        savedWarpedPointsX.at(y, x) = -2.0f;
        savedWarpedPointsY.at(y, x) = -2.0f;
        if (dbg) {
          dbg->residual.at(y, x) = 0; dbg->weight.at(y, x) = 0; dbg->warped.at(y, x) = 0;
          dbg->warpedX.at(y, x) = -2.0f; dbg->warpedY.at(y, x) = -2.0f;
          for (int i = 0; i < 6; i++) dbg->J[i].at(y, x) = 0;
        }
        continue;
      }
      const int idx = x + nCols * y;
      const float Z = depth.at(y, x);
      WarpOut w = warp_point(x, y, Z, k, S);
      float warpedintensity = current_frame->getInterpolatedElement(w.wx, w.wy, 1);
      if (warpedintensity == -1) {
        savedWarpedPointsX.at(y, x) = -1.0f;
        savedWarpedPointsY.at(y, x) = -1.0f;
      } else {
        savedWarpedPointsX.at(y, x) = w.wx;
        savedWarpedPointsY.at(y, x) = w.wy;
      }
      float gradx = current_frame->getInterpolatedGradX(w.wx, w.wy);
      float grady = current_frame->getInterpolatedGradY(w.wx, w.wy);
      float J[6];
      jacobian_row(gradx, grady, x, y, Z, k, J);
      float residual;
      if (warpedintensity == -1) residual = 0.0f;
      else residual = warpedintensity - float(pimg.at(y, x));
      display_iterationres.at(y, x) = residual;
      float res_weight;
      if (warpedintensity == -1) res_weight = 0;
      else {
        float px = w.px, py = w.py, pz = w.pz;
        float d = 1.0f / Z;
        float rp = residual;
        float gx = k.fx * gradx;
        float gy = k.fy * grady;
        float s = 1.0f * var[idx];
        float g0 = (tx * pz - tz * px) / (pz * pz * d);
        float g1 = (ty * pz - tz * py) / (pz * pz * d);
        float drpdd = gx * g0 + gy * g1;
        float w_p = 1.0f / (16.0f + s * drpdd * drpdd);  // CAMERA_PIXEL_NOISE_2
        float weighted_rp = std::fabs(rp * sqrtf(w_p));
        float wh = std::fabs(weighted_rp < (3.0f / 2) ? 1 : (3.0f / 2) / weighted_rp);  // HUBER_D/2
        res_weight = wh * w_p;
      }
      display_weightimg.at(y, x) = res_weight;
      if (dbg) {
        dbg->residual.at(y, x) = residual; dbg->weight.at(y, x) = res_weight;
        dbg->warped.at(y, x) = (warpedintensity == -1) ? 0.f : warpedintensity;
        dbg->warpedX.at(y, x) = savedWarpedPointsX.at(y, x); dbg->warpedY.at(y, x) = savedWarpedPointsY.at(y, x);
        for (int i = 0; i < 6; i++) dbg->J[i].at(y, x) = J[i];
      }
      // :364-374  wJ = J^T.mul(w) (f32), H += wJ*J (K=1 product rounded to f32, then f32 add),
      //           b += J.mul(r*w)
      float wJ[6];
      for (int i = 0; i < 6; i++) wJ[i] = J[i] * res_weight;
      const float rw = residual * res_weight;
      for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) {
          float t = wJ[i] * J[j];
          H[i * 6 + j] += t;
          Hd[i * 6 + j] += (double)t;
        }
        float t = J[i] * rw;
        b[i] += t;
        bd[i] += (double)t;
      }
    }
  }
  std::memcpy(H_out, H, sizeof(H));
  std::memcpy(b_out, b, sizeof(b));
  std::memcpy(Hd_out, Hd, sizeof(Hd));
  std::memcpy(bd_out, bd, sizeof(bd));
}

// PixelWisePyramid.cpp:416-455
void PixelWisePyramid::calculatePixelWiseParallel() {
  const int T = n_threads;
  std::vector<float> Hs((size_t)T * 36), bs((size_t)T * 6);
  std::vector<double> Hds((size_t)T * 36), bds((size_t)T * 6);
  const int y_increment = nRows / T;
  auto band = [&](int t) {
    int y0 = t * y_increment, y1 = (t == T - 1) ? nRows : (t + 1) * y_increment;
    calculatePixelWise(y0, y1, &Hs[(size_t)t * 36], &bs[(size_t)t * 6], &Hds[(size_t)t * 36], &bds[(size_t)t * 6]);
  };
  if (thread_mode == 2) {
    BandPool::get(T).run(band);
  } else if (thread_mode == 1) {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(band, t);
    for (auto& t : th) t.join();
  } else {
    for (int t = 0; t < T; t++) band(t);
  }
  // hessian = h1 + h2 + h3 (left to right, f32)
  for (int i = 0; i < 36; i++) {
    float s = Hs[i];
    double sd = Hds[i];
    for (int t = 1; t < T; t++) { s = s + Hs[(size_t)t * 36 + i]; sd += Hds[(size_t)t * 36 + i]; }
    hessian[i] = s;
    hessian_d[i] = sd;
  }
  for (int i = 0; i < 6; i++) {
    float s = bs[i];
    double sd = bds[i];
    for (int t = 1; t < T; t++) { s = s + bs[(size_t)t * 6 + i]; sd += bds[(size_t)t * 6 + i]; }
    sd_param[i] = s;
    sd_param_d[i] = sd;
  }
  if (sum_mode == SUM_F64) {
    for (int i = 0; i < 36; i++) hessian[i] = (float)hessian_d[i];
    for (int i = 0; i < 6; i++) sd_param[i] = (float)sd_param_d[i];
  }
  lu_inverse_f32(hessian, 6, hessianInv);
  updatePose();
}

// PixelWisePyramid.cpp:460-491
void PixelWisePyramid::updatePose() {
  // cv::gemm f32 (6x6 * 6x1) accumulates in double and rounds once; then negated.
  for (int i = 0; i < 6; i++) {
    double s = 0;
    for (int kk = 0; kk < 6; kk++) s += (double)sd_param[kk] * (double)hessianInv[i * 6 + kk];
    deltapose[i] = -(float)s;
  }
  static const float wgt[6] = {100000.0f, 100000.0f, 100000.0f, 10000.0f, 10000.0f, 10000.0f};  // ExternVariable.h:76
  float weighted_pose = std::fabs(deltapose[0] * wgt[0]) + std::fabs(deltapose[1] * wgt[1]) + std::fabs(deltapose[2] * wgt[2]) +
                        std::fabs(deltapose[3] * wgt[3]) + std::fabs(deltapose[4] * wgt[4]) + std::fabs(deltapose[5] * wgt[5]);
  weightedPose = weighted_pose;
  concatenate_relative_pose(deltapose, pose, pose);
}

// PixelWisePyramid.cpp:544-549 (useAverageWeights == true is the only live branch, ImageFunc.cpp:285)
void PixelWisePyramid::saveWeights(bool useAverageWeights) {
  if (!useAverageWeights) return;
  PlaneF& wp = prev_frame->weight_pyramid[pyrlevel];
  for (int y = 0; y < nRows; y++)
    for (int x = 0; x < nCols; x++) wp.at(y, x) = wp.at(y, x) + display_weightimg.at(y, x);
  prev_frame->numWeightsAdded[pyrlevel]++;
}

// PixelWisePyramid.cpp:561-680
void PixelWisePyramid::precomputePixelWiseInvCompositional(int ymin, int ymax) {
  const Intrin k = get_intrinsic(prev_frame->cfg, prev_frame->pyrLevel);
  const size_t N = (size_t)nRows * nCols;
  const PlaneF& depth = prev_frame->depth_pyramid[pyrlevel];
  const PlaneF& wgt = prev_frame->weight_pyramid[pyrlevel];
  for (int y = ymin; y < ymax; y++)
    for (int x = 0; x < nCols; x++) {
      const size_t idx = (size_t)x + (size_t)nCols * y;
      if (prev_frame->mask.at(y, x) == 0) {
        for (int i = 0; i < 6; i++) { steepestDescent[i * N + idx] = 0; weightedSteepestDescent[i * N + idx] = 0; }
        continue;
      }
      float J[6];
      jacobian_row(prev_frame->gradientx.at(y, x), prev_frame->gradienty.at(y, x), x, y, depth.at(y, x), k, J);
      for (int i = 0; i < 6; i++) {
        steepestDescent[i * N + idx] = J[i];
        weightedSteepestDescent[i * N + idx] = J[i] * wgt.at(y, x);
      }
    }
}

// PixelWisePyramid.cpp:687-913
void PixelWisePyramid::iteratePixelWiseInvCompositional(int ymin, int ymax, float b_out[6], double bd_out[6]) {
  const Intrin k = get_intrinsic(prev_frame->cfg, prev_frame->pyrLevel);
  const size_t N = (size_t)nRows * nCols;
  float b[6];      // band sums on this thread's stack, stored once (see calculatePixelWise)
  double bd[6];
  for (int i = 0; i < 6; i++) { b[i] = 0; bd[i] = 0; }
  float SE3[16];
  se3_exp(pose, SE3);
  const PlaneU8& pimg = prev_frame->image_pyramid[pyrlevel];
  const PlaneF& depth = prev_frame->depth_pyramid[pyrlevel];
  const PlaneF& wgt = prev_frame->weight_pyramid[pyrlevel];
  for (int y = ymin; y < ymax; y++)
    for (int x = 0; x < nCols; x++) {
      if (prev_frame->mask.at(y, x) == 0) {
        display_iterationres.at(y, x) = 0;
        display_weightimg.at(y, x) = 0;
        if (dbg) { dbg->residual.at(y, x) = 0; dbg->weight.at(y, x) = 0; }
        continue;
      }
      const size_t idx = (size_t)x + (size_t)nCols * y;
      WarpOut w = warp_point(x, y, depth.at(y, x), k, SE3);
      float warpedintensity = current_frame->getInterpolatedElement(w.wx, w.wy, 1);
      float residual = (warpedintensity == -1) ? 0.0f : warpedintensity - float(pimg.at(y, x));
      display_iterationres.at(y, x) = residual;
      display_weightimg.at(y, x) = wgt.at(y, x);
      if (dbg) { dbg->residual.at(y, x) = residual; dbg->weight.at(y, x) = wgt.at(y, x); }
      const float rw = residual * wgt.at(y, x);
      for (int i = 0; i < 6; i++) {
        float t = steepestDescent[i * N + idx] * rw;
        b[i] += t;
        bd[i] += (double)t;
      }
    }
  std::memcpy(b_out, b, sizeof(b));
  std::memcpy(bd_out, bd, sizeof(bd));
}

// PixelWisePyramid.cpp:917-974 (FLAG_DO_PARALLEL_CONST_WEIGHT_POSE_EST branch: 3 bands precompute, 2 uneven bands iterate)
void PixelWisePyramid::calculatePixelWiseParallelInvCompositional(int iter) {
  const int y_increment = nRows / 3;  // NUM_CONST_WT_POSE_EST_THREADS
  const size_t N = (size_t)nRows * nCols;
  if (iter == 0) {
    steepestDescent.assign(6 * N, 0.f);
    weightedSteepestDescent.assign(6 * N, 0.f);
    precomputePixelWiseInvCompositional(0, y_increment);
    precomputePixelWiseInvCompositional(y_increment, 2 * y_increment);
    precomputePixelWiseInvCompositional(2 * y_increment, nRows);
    // hessian = WSD * SD^T : cv::gemm f32 with double accumulation, rounded once (:938)
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double s = 0;
        const float* a = &weightedSteepestDescent[i * N];
        const float* c = &steepestDescent[j * N];
        for (size_t q = 0; q < N; q++) s += (double)a[q] * (double)c[q];
        hessian[i * 6 + j] = (float)s;
        hessian_d[i * 6 + j] = s;
      }
    lu_inverse_f32(hessian, 6, hessianInv);
  }
  float b1[6], b2[6];
  double bd1[6], bd2[6];
  auto f1 = [&]() { iteratePixelWiseInvCompositional(0, y_increment, b1, bd1); };
  auto f2 = [&]() { iteratePixelWiseInvCompositional(y_increment, nRows, b2, bd2); };
  if (thread_mode == 2) {
    BandPool::get(2).run([&](int t) { if (t == 0) f1(); else f2(); });
  } else if (thread_mode == 1) {
    std::thread t1(f1), t2(f2);
    t1.join();
    t2.join();
  } else {
    f1();
    f2();
  }
  for (int i = 0; i < 6; i++) {
    sd_param[i] = b1[i] + b2[i];
    sd_param_d[i] = bd1[i] + bd2[i];
    if (sum_mode == SUM_F64) sd_param[i] = (float)sd_param_d[i];
  }
  updatePose();
}

// ImageFunc.cpp:92-138, 150-307
AlignResult GetImagePoseEstimate(Frame* prev_frame, Frame* current_frame, const DepthPyr* dm, Frame* tminus1,
                                 const float* init_rel_pose, bool fromLoopClosure, bool save_weights, SumMode mode,
                                 int thread_mode, int n_threads) {
  const Config& cfg = prev_frame->cfg;
  AlignResult res;
  std::memset(&res, 0, sizeof(res));
  float pose[6] = {0, 0, 0, 0, 0, 0};
  if (init_rel_pose) std::memcpy(pose, init_rel_pose, sizeof(pose));
  else concatenate_origin_pose(tminus1->poseWrtWorld, prev_frame->poseWrtWorld, pose);  // :106
  for (int level = cfg.levels - 1; level >= 0; level--) {
    prev_frame->updationOnPyrChange(level);            // :158
    current_frame->updationOnPyrChange(level, false);  // :159
    PixelWisePyramid wp(prev_frame, current_frame, pose, dm);
    wp.sum_mode = mode;
    wp.thread_mode = thread_mode;
    wp.n_threads = n_threads;
    int iter;
    int executed = 0;
    for (iter = 0; iter < cfg.max_iter[level]; ++iter) {
      if (fromLoopClosure) wp.calculatePixelWiseParallelInvCompositional(iter);  // :243
      else wp.calculatePixelWiseParallel();                                       // :247
      executed++;
      res.last_weighted = wp.weightedPose;
      if (cfg.early_exit && wp.weightedPose < 1.0f) iter = cfg.max_iter[level] - 1;  // :251-252
      if (save_weights && !fromLoopClosure && iter == cfg.max_iter[level] - 1) wp.saveWeights(true);  // :280-288
    }
    res.iters[level] = executed;
  }
  // :305-307
  concatenate_relative_pose(pose, prev_frame->poseWrtOrigin, current_frame->poseWrtOrigin);
  concatenate_relative_pose(pose, prev_frame->poseWrtWorld, current_frame->poseWrtWorld);
  std::memcpy(res.pose, pose, sizeof(pose));
  return res;
}

}  // namespace ellc_oracle
