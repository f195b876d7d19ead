// ELLC ORACLE (test infrastructure) — flat C entry points so tests/ and bench.py's cpu_baseline leg can
// drive the restatement through ctypes. Nothing in the product path links this file.
#include "ellc_oracle.hpp"
#include <malloc.h>
#include <cstring>
#include <chrono>
#include <thread>

using namespace ellc_oracle;

extern "C" {

struct orc_config {
  int width, height, levels;
  float fx, fy, cx, cy;
  int max_iter[kMaxLevels];
  int early_exit;
  int num_pose_threads;
};

static Config to_cfg(const orc_config* c) {
  Config k;
  k.width = c->width; k.height = c->height; k.levels = c->levels;
  k.fx = c->fx; k.fy = c->fy; k.cx = c->cx; k.cy = c->cy;
  for (int i = 0; i < kMaxLevels; i++) k.max_iter[i] = c->max_iter[i];
  k.early_exit = c->early_exit;
  k.num_pose_threads = c->num_pose_threads > 0 ? c->num_pose_threads : 3;
  return k;
}

// ---- algebra
void orc_se3_exp(const float* pose, float* T) { se3_exp(pose, T); }
void orc_se3_log(const float* T, float* pose) { se3_log(T, pose); }
void orc_se3_exp_d(const double* pose, double* T) { se3_exp_d(pose, T); }
void orc_se3_log_d(const double* T, double* pose) { se3_log_d(T, pose); }
void orc_concat_relative(const float* a, const float* b, float* out) { concatenate_relative_pose(a, b, out); }
void orc_concat_origin(const float* a, const float* b, float* out) { concatenate_origin_pose(a, b, out); }
void orc_inv_lie_pose(const float* a, float* out) { inv_lie_pose(a, out); }
int orc_lu_inverse(const float* A, int n, float* out) { return lu_inverse_f32(A, n, out); }
void orc_get_intrinsic(const orc_config* c, int level, float* out4) {
  Intrin k = get_intrinsic(to_cfg(c), level);
  out4[0] = k.fx; out4[1] = k.fy; out4[2] = k.cx; out4[3] = k.cy;
}
void orc_kmats(const orc_config* c, float* K9, float* Kinv9) {
  KMats m = make_kmats(to_cfg(c));
  std::memcpy(K9, m.K, sizeof(m.K));
  std::memcpy(Kinv9, m.Kinv, sizeof(m.Kinv));
}

// ---- image side (free functions)
void orc_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) {
  PlaneU8 s(w, h), d;
  std::memcpy(s.d.data(), src, (size_t)w * h);
  pyr_down_u8(s, d);
  std::memcpy(dst, d.d.data(), d.d.size());
}
void orc_gradient(const uint8_t* img, int stored_w, int stored_h, int rows, int cols, float* gx, float* gy) {
  PlaneU8 s(stored_w, stored_h);
  std::memcpy(s.d.data(), img, (size_t)stored_w * stored_h);
  PlaneF a, b;
  calculate_gradient(s, rows, cols, a, b);
  std::memcpy(gx, a.d.data(), a.d.size() * 4);
  std::memcpy(gy, b.d.data(), b.d.size() * 4);
}
void orc_max_gradients(const float* gx, const float* gy, int w, int h, float* out, int* n_substantial) {
  PlaneF a(w, h), b(w, h), o;
  std::memcpy(a.d.data(), gx, (size_t)w * h * 4);
  std::memcpy(b.d.data(), gy, (size_t)w * h * 4);
  build_max_gradients(a, b, o, n_substantial);
  std::memcpy(out, o.d.data(), o.d.size() * 4);
}
void orc_tap_u8(const uint8_t* img, int stored_w, int stored_h, int rows, int cols, const float* xs, const float* ys, int n,
                int check, float* out) {
  PlaneU8 s(stored_w, stored_h);
  std::memcpy(s.d.data(), img, (size_t)stored_w * stored_h);
  for (int i = 0; i < n; i++) out[i] = tap_u8(s, rows, cols, xs[i], ys[i], check);
}
void orc_tap_f32(const float* img, int w, int h, const float* xs, const float* ys, int n, float* out) {
  PlaneF s(w, h);
  std::memcpy(s.d.data(), img, (size_t)w * h * 4);
  for (int i = 0; i < n; i++) out[i] = tap_f32(s, h, w, xs[i], ys[i]);
}

// ---- frame handle
Frame* orc_frame_create(const orc_config* c, const uint8_t* gray, int id) {
  Frame* f = new Frame();
  f->init(to_cfg(c), gray, id);
  return f;
}
void orc_frame_destroy(Frame* f) { delete f; }
void orc_frame_set_early_exit(Frame* f, int e) { f->cfg.early_exit = e; }
void orc_frame_set_max_iter(Frame* f, const int* mi) { for (int i = 0; i < kMaxLevels; i++) f->cfg.max_iter[i] = mi[i]; }
void orc_frame_level_dims(Frame* f, int level, int* stored_w, int* stored_h, int* cols, int* rows) {
  *stored_w = f->image_pyramid[level].w; *stored_h = f->image_pyramid[level].h;
  *cols = f->width >> level; *rows = f->height >> level;
}
void orc_frame_get_image(Frame* f, int level, uint8_t* out) {
  std::memcpy(out, f->image_pyramid[level].d.data(), f->image_pyramid[level].d.size());
}
void orc_frame_set_depth(Frame* f, int level, const float* depth) {
  std::memcpy(f->depth_pyramid[level].d.data(), depth, f->depth_pyramid[level].d.size() * 4);
}
void orc_frame_get_depth(Frame* f, int level, float* depth) {
  std::memcpy(depth, f->depth_pyramid[level].d.data(), f->depth_pyramid[level].d.size() * 4);
}
void orc_frame_set_weights(Frame* f, int level, const float* w, int count) {
  std::memcpy(f->weight_pyramid[level].d.data(), w, f->weight_pyramid[level].d.size() * 4);
  f->numWeightsAdded[level] = count;
}
void orc_frame_get_weights(Frame* f, int level, float* w, int* count) {
  std::memcpy(w, f->weight_pyramid[level].d.data(), f->weight_pyramid[level].d.size() * 4);
  if (count) *count = f->numWeightsAdded[level];
}
void orc_frame_finalise_weights(Frame* f) { f->finaliseWeights(); }
void orc_frame_get_max_gradient(Frame* f, float* out, int* n) {
  std::memcpy(out, f->maxAbsGradient.d.data(), f->maxAbsGradient.d.size() * 4);
  if (n) *n = f->no_points_substantial_grad;
}
void orc_frame_set_pose(Frame* f, const float* origin6, const float* world6) {
  if (origin6) std::memcpy(f->poseWrtOrigin, origin6, 24);
  if (world6) std::memcpy(f->poseWrtWorld, world6, 24);
}
void orc_frame_get_pose(Frame* f, float* origin6, float* world6) {
  std::memcpy(origin6, f->poseWrtOrigin, 24);
  std::memcpy(world6, f->poseWrtWorld, 24);
}
float orc_frame_get_rescale(Frame* f) { return f->rescaleFactor; }
void orc_frame_update_level(Frame* f, int level, int is_prev) { f->updationOnPyrChange(level, is_prev != 0); }
void orc_frame_get_gradient(Frame* f, float* gx, float* gy) {
  std::memcpy(gx, f->gradientx.d.data(), f->gradientx.d.size() * 4);
  std::memcpy(gy, f->gradienty.d.data(), f->gradienty.d.size() * 4);
}
int orc_frame_get_mask(Frame* f, uint8_t* mask) {
  std::memcpy(mask, f->mask.d.data(), f->mask.d.size());
  return f->no_nonZeroDepthPts;
}

// ---- depth variance pyramid
DepthPyr* orc_depthpyr_create(const orc_config* c) {
  DepthPyr* p = new DepthPyr();
  p->deptharr.resize(c->levels);
  p->depthvararr.resize(c->levels);
  for (int l = 0; l < c->levels; l++) {
    p->deptharr[l].assign((size_t)(c->width >> l) * (c->height >> l), 0.f);
    p->depthvararr[l].assign((size_t)(c->width >> l) * (c->height >> l), 0.f);
  }
  return p;
}
void orc_depthpyr_destroy(DepthPyr* p) { delete p; }
void orc_depthpyr_set_var(DepthPyr* p, int level, const float* var) {
  std::memcpy(p->depthvararr[level].data(), var, p->depthvararr[level].size() * 4);
}

// ---- one GN step at the frame's current level (kf and cur must have been put on `level`)
//   mode 0: FCA (calculatePixelWiseParallel); mode 1: ICA, iter index given (precompute when iter == 0)
struct orc_gn_handle {
  PixelWisePyramid* p;
  float pose[6];
  GNDebugPlanes dbg;
};
orc_gn_handle* orc_gn_begin(Frame* kf, Frame* cur, DepthPyr* dp, int level, const float* pose, int sum_mode, int n_threads,
                            int want_planes) {
  kf->updationOnPyrChange(level, true);
  cur->updationOnPyrChange(level, false);
  orc_gn_handle* h = new orc_gn_handle();
  std::memcpy(h->pose, pose, 24);
  h->p = new PixelWisePyramid(kf, cur, h->pose, dp);
  h->p->sum_mode = (SumMode)sum_mode;
  h->p->n_threads = n_threads;
  if (want_planes) {
    int w = h->p->nCols, hh = h->p->nRows;
    h->dbg.residual = PlaneF(w, hh); h->dbg.weight = PlaneF(w, hh); h->dbg.warpedX = PlaneF(w, hh);
    h->dbg.warpedY = PlaneF(w, hh); h->dbg.warped = PlaneF(w, hh);
    h->dbg.J.assign(6, PlaneF(w, hh));
    h->p->dbg = &h->dbg;
  }
  return h;
}
void orc_gn_step(orc_gn_handle* h, int mode, int iter, float* H36, float* b6, float* Hinv36, float* delta6, float* pose_out6,
                 float* weighted, double* H36d, double* b6d) {
  if (mode == 0) h->p->calculatePixelWiseParallel();
  else h->p->calculatePixelWiseParallelInvCompositional(iter);
  if (H36) std::memcpy(H36, h->p->hessian, 144);
  if (b6) std::memcpy(b6, h->p->sd_param, 24);
  if (Hinv36) std::memcpy(Hinv36, h->p->hessianInv, 144);
  if (delta6) std::memcpy(delta6, h->p->deltapose, 24);
  if (pose_out6) std::memcpy(pose_out6, h->pose, 24);
  if (weighted) *weighted = h->p->weightedPose;
  if (H36d) std::memcpy(H36d, h->p->hessian_d, 288);
  if (b6d) std::memcpy(b6d, h->p->sd_param_d, 48);
}
void orc_gn_planes(orc_gn_handle* h, float* residual, float* weight, float* warpedX, float* warpedY, float* J6) {
  size_t n = h->dbg.residual.d.size();
  if (residual) std::memcpy(residual, h->dbg.residual.d.data(), n * 4);
  if (weight) std::memcpy(weight, h->dbg.weight.d.data(), n * 4);
  if (warpedX) std::memcpy(warpedX, h->dbg.warpedX.d.data(), n * 4);
  if (warpedY) std::memcpy(warpedY, h->dbg.warpedY.d.data(), n * 4);
  if (J6) for (int i = 0; i < 6; i++) std::memcpy(J6 + i * n, h->dbg.J[i].d.data(), n * 4);
}
void orc_gn_sd(orc_gn_handle* h, float* sd6n, float* wsd6n) {
  if (sd6n) std::memcpy(sd6n, h->p->steepestDescent.data(), h->p->steepestDescent.size() * 4);
  if (wsd6n) std::memcpy(wsd6n, h->p->weightedSteepestDescent.data(), h->p->weightedSteepestDescent.size() * 4);
}
void orc_gn_save_weights(orc_gn_handle* h) { h->p->saveWeights(true); }
void orc_gn_end(orc_gn_handle* h) { delete h->p; delete h; }

// ---- full alignment (GetImagePoseEstimate). flags: bit0 fromLoopClosure (ICA), bit1 save weights,
// bit2 spawn real threads per iteration (CPU baseline behaviour), bit3 persistent worker pool instead; sum_mode as above.
void orc_align(Frame* kf, Frame* cur, DepthPyr* dp, const float* init_pose, int flags, int sum_mode, int n_threads,
               float* pose6, int* iters, float* last_weighted) {
  AlignResult r = GetImagePoseEstimate(kf, cur, dp, cur, init_pose, (flags & 1) != 0, (flags & 2) != 0, (SumMode)sum_mode,
                                       (flags & 8) ? 2 : ((flags & 4) ? 1 : 0), n_threads);
  std::memcpy(pose6, r.pose, 24);
  if (iters) for (int l = 0; l < kf->cfg.levels; l++) iters[l] = r.iters[l];
  if (last_weighted) *last_weighted = r.last_weighted;
}

// CPU baseline: repeat the alignment `reps` times, return seconds and the number of GN iterations executed.
double orc_align_timed(Frame* kf, Frame* cur, DepthPyr* dp, const float* init_pose, int flags, int n_threads, int reps,
                       long long* gn_iterations) {
  long long its = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) {
    AlignResult a = GetImagePoseEstimate(kf, cur, dp, cur, init_pose, (flags & 1) != 0, false, SUM_F32_BANDS,
                                         (flags & 8) ? 2 : ((flags & 4) ? 1 : 0), n_threads);
    for (int l = 0; l < kf->cfg.levels; l++) its += a.iters[l];
  }
  auto t1 = std::chrono::steady_clock::now();
  if (gn_iterations) *gn_iterations = its;
  return std::chrono::duration<double>(t1 - t0).count();
}
// CPU baseline, batch-parallel: n_align independent alignments (each with its OWN frame / pyramid objects: GetImagePoseEstimate
// changes the frames' level state) spread over n_outer host threads, alignment i on thread i % n_outer, every alignment repeated
// `reps` times; inside an alignment the row bands run as `flags` says (bit2: three threads created / joined per iteration as the
// reference does; otherwise one after the other on the alignment's thread). The reference itself runs a loop-closure batch's
// alignments one after the other (GlobalOptimize.cpp:480-610): this is the strongest fair use of the host, not the reference's.
double orc_align_batch_timed(Frame** kfs, Frame** curs, DepthPyr** dps, int n_align, int n_outer, int flags, int n_threads, int reps,
                             long long* gn_iterations) {
  // (the alignment allocates its working planes per level as the reference does; as mmap / munmap of fresh pages those calls
  // serialise the threads on the process's address-space lock — keep freed blocks in the heap instead)
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  std::vector<long long> its((size_t)n_outer, 0);
  std::vector<std::thread> th;
  auto t0 = std::chrono::steady_clock::now();
  for (int j = 0; j < n_outer; j++)
    th.emplace_back([&, j]() {
      const float zero[6] = {0, 0, 0, 0, 0, 0};
      long long mine = 0;
      for (int i = j; i < n_align; i += n_outer)
        for (int r = 0; r < reps; r++) {
          AlignResult a = GetImagePoseEstimate(kfs[i], curs[i], dps[i], curs[i], zero, (flags & 1) != 0, false, SUM_F32_BANDS, (flags & 4) ? 1 : 0, n_threads);
          for (int l = 0; l < kfs[i]->cfg.levels; l++) mine += a.iters[l];
        }
      its[(size_t)j] = mine;
    });
  for (auto& t : th) t.join();
  auto t1 = std::chrono::steady_clock::now();
  long long tot = 0;
  for (long long v : its) tot += v;
  if (gn_iterations) *gn_iterations = tot;
  return std::chrono::duration<double>(t1 - t0).count();
}
int orc_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

// ---- depth map handle (SoA in / out)
DepthMap* orc_dm_create(const orc_config* c) {
  DepthMap* d = new DepthMap();
  d->init(to_cfg(c));
  return d;
}
void orc_dm_destroy(DepthMap* d) { delete d; }
void orc_dm_set_state(DepthMap* d, const float* id, const float* ids, const float* var, const float* vars, const int* validity,
                      const int* blacklisted, const uint8_t* valid) {
  size_t n = d->current.size();
  for (size_t i = 0; i < n; i++) {
    Hyp& h = d->current[i];
    h.invDepth = id[i]; h.invDepthSmoothed = ids[i]; h.variance = var[i]; h.varianceSmoothed = vars[i];
    h.validity_counter = validity[i]; h.blacklisted = blacklisted[i]; h.isValid = valid[i] ? 1 : 0;
  }
}
void orc_dm_get_state(DepthMap* d, float* id, float* ids, float* var, float* vars, int* validity, int* blacklisted,
                      uint8_t* valid) {
  size_t n = d->current.size();
  for (size_t i = 0; i < n; i++) {
    const Hyp& h = d->current[i];
    id[i] = h.invDepth; ids[i] = h.invDepthSmoothed; var[i] = h.variance; vars[i] = h.varianceSmoothed;
    validity[i] = h.validity_counter; blacklisted[i] = h.blacklisted; valid[i] = h.isValid;
  }
}
void orc_dm_set_keyframe(DepthMap* d, Frame* kf) { d->keyFrame = kf; }
void orc_dm_set_current(DepthMap* d, Frame* f) { d->currentFrame = f; }
void orc_dm_propagate(DepthMap* d, Frame* nk) { d->propagateDepth(nk); }
void orc_dm_observe(DepthMap* d) { d->observeDepthRowParallel(); }
void orc_dm_fill_holes(DepthMap* d) { d->fillDepthHoles(); }
void orc_dm_regularize(DepthMap* d, int removeOcclusions) { d->regularizeDepthMap(removeOcclusions != 0); }
float orc_dm_make_inv_depth_one(DepthMap* d) { d->makeInvDepthOne(); return d->depthScale; }
void orc_dm_update_depth_image(DepthMap* d) { d->updateDepthImage(); }
void orc_dm_create_keyframe(DepthMap* d, Frame* nk) { d->createKeyFrame(nk); }
float orc_dm_seeds(DepthMap* d) { return d->calculate_no_of_Seeds(); }
void orc_dm_get_pyr(DepthMap* d, int level, float* depth, float* var) {
  std::memcpy(depth, d->pyr.deptharr[level].data(), d->pyr.deptharr[level].size() * 4);
  std::memcpy(var, d->pyr.depthvararr[level].data(), d->pyr.depthvararr[level].size() * 4);
}
void orc_dm_get_integral(DepthMap* d, int* out) { std::memcpy(out, d->validityIntegralBuffer.data(), d->validityIntegralBuffer.size() * 4); }
DepthPyr* orc_dm_pyr(DepthMap* d) { return &d->pyr; }
// level-0 arrays in the reference's array convention (depth -1 / var -1 where invalid), then A25/A26
void orc_dm_set_pyr0(DepthMap* d, const float* deptharr0, const float* vararr0) {
  std::memcpy(d->pyr.deptharr[0].data(), deptharr0, d->pyr.deptharr[0].size() * 4);
  std::memcpy(d->pyr.depthvararr[0].data(), vararr0, d->pyr.depthvararr[0].size() * 4);
}
void orc_dm_build_inv_var_depth(DepthMap* d) { d->buildInvVarDepth(); }
void orc_dm_map_depth_to_keyframe(DepthMap* d) { d->mapDepthArr2Mat(); }
// one line-stereo probe (for unit tests): returns the error code / best error
float orc_dm_line_stereo(DepthMap* d, float u, float v, float epxn, float epyn, float min_id, float prior, float max_id,
                         float* out3) {
  float a = 0, b = 0, c = 0;
  float e = d->doLineStereo(u, v, epxn, epyn, min_id, prior, max_id, a, b, c);
  out3[0] = a; out3[1] = b; out3[2] = c;
  return e;
}
int orc_dm_check_epl(DepthMap* d, int x, int y, float* ep2) { return d->makeAndCheckEPL(x, y, ep2, ep2 + 1) ? 1 : 0; }
// inputs of the depth side's second source (tests/second_source_depth.py): the frame-to-frame matrices of Frame.cpp:376-413 and
// the inverse-intrinsics constants of EigenInitialization.cpp:20-34 (third-party arithmetic — Eigen exp / inverse, cv::Mat::inv —
// that the second source takes as given)
void orc_frame_calc_se3(Frame* f, Frame* other) { f->calculateSE3poseOtherWrtThis(*other); }
void orc_frame_get_se3(Frame* f, float* out44) {
  std::memcpy(out44, f->SE3poseOtherWrtThis, 64);
  std::memcpy(out44 + 16, f->SE3poseThisWrtOther, 64);
  std::memcpy(out44 + 32, f->K_SE3poseThisWrtOther_r, 36);
  std::memcpy(out44 + 41, f->K_SE3poseThisWrtOther_t, 12);
}
void orc_kmats_inv(Frame* f, float* out4) {
  KMats km = make_kmats(f->cfg);
  out4[0] = km.fx_inv; out4[1] = km.fy_inv; out4[2] = km.cx_inv; out4[3] = km.cy_inv;
}

}  // extern "C"
