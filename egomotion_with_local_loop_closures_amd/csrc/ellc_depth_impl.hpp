// Semi-dense depth map entry points of the C ABI (class depthMap, DepthPropagation.cpp).
// Included at the end of ellc_hip.hip (the library is one translation unit).
#pragma once
#include "ellc_context.hpp"
#include "ellc_kernels_depth.hpp"
#include <cstring>

using namespace ellc;

namespace {

dim3 grid2(int w, int h, dim3 blk) { return dim3((w + blk.x - 1) / blk.x, (h + blk.y - 1) / blk.y); }

// the matrices new <- old of a propagation (frame::calculateSE3poseOtherWrtThis, Frame.cpp:376-413): This-w.r.t.-Other of the new keyframe
struct RelMats { float two[12]; };
RelMats relative_matrices(const ellc_ctx* c, const float* pose) {
  RelMats m;
  const float zero[6] = {0, 0, 0, 0, 0, 0};
  float rel[6], otw[12];
  concat_origin_f32(zero, pose, rel);
  exp_se3_f32(rel, otw);
  invert_f32(otw, m.two);
  (void)c;
  return m;
}

void swap_maps(ellc_ctx* c) { std::swap(c->dm_cur, c->dm_oth); }

ellc_status need_map(ellc_ctx* c) {
  if (!c) return ELLC_ERR_BAD_ARG;
  if (!c->dm_ready) return fail(c, ELLC_ERR_NOT_READY, "depth map: call ellc_depth_set_state and ellc_depth_set_keyframe first");
  return ELLC_OK;
}

ellc_status do_regularize(ellc_ctx* c, int removeOcclusions, const int* gate = nullptr) {
  const int W = c->cfg.width, H = c->cfg.height;
  const int tiles_x = (W + DM_TX - 1) / DM_TX, tiles = tiles_x * ((H + DM_TY - 1) / DM_TY);
  // in place, except for the validity flags: they go to the other map's plane, which becomes this map's
  hipLaunchKernelGGL(dm_regularize, dim3(8 * ((tiles + 7) / 8)), dim3(DM_TX * DM_TY), 0, c->stream, c->dm_cur, c->dm_oth.isValid, W, H, removeOcclusions,
                     tiles_x, tiles, gate);
  ELLC_HIP(c, hipGetLastError());
  std::swap(c->dm_cur.isValid, c->dm_oth.isValid);
  return ELLC_OK;
}

ellc_status do_fill_holes(ellc_ctx* c, const int* gate = nullptr) {
  const int W = c->cfg.width, H = c->cfg.height;
  const int tiles_x = (W + DM_TX - 1) / DM_TX, tiles = tiles_x * ((H + DM_TY - 1) / DM_TY);
  hipLaunchKernelGGL(dm_fill_holes, dim3(8 * ((tiles + 7) / 8)), dim3(DM_TX * DM_TY), 0, c->stream, c->dm_cur, c->dm_oth, c->kf_maxgrad[c->dm_kf_slot], W, H,
                     tiles_x, tiles, gate);
  ELLC_HIP(c, hipGetLastError());
  swap_maps(c);
  return ELLC_OK;
}

// makeInvDepthOne (:1546-1587), enqueue only: the factor stays on the device (the second stage of the sum is redone by every
// block of the rescale: 256 partials, one launch less); do_rescale_finish fetches it
ellc_status do_rescale_enqueue(ellc_ctx* c) {
  const int n = c->cfg.width * c->cfg.height;
  const int nb = 256;
  hipLaunchKernelGGL(dm_sum_stage1, dim3(nb), dim3(256), 0, c->stream, c->dm_cur, n, c->red_scratch);
  hipLaunchKernelGGL(dm_rescale, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->dm_cur, n, c->red_scratch, nb, (float*)(c->red_scratch + 2 * nb + 2));
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}
ellc_status do_rescale_finish(ellc_ctx* c, float* factor_out) {
  const float* factor_d = (const float*)(c->red_scratch + 2 * 256 + 2);
  float f = 0;
  ELLC_HIP(c, hipMemcpyAsync(&f, factor_d, 4, hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  c->dm_depth_scale = f;
  c->dm_global_scale *= f;   // util::GLOABL_DEPTH_SCALE, kept per context
  if (factor_out) *factor_out = f;
  return ELLC_OK;
}
ellc_status do_rescale(ellc_ctx* c, float* factor_out) {
  const ellc_status s = do_rescale_enqueue(c);
  return s != ELLC_OK ? s : do_rescale_finish(c, factor_out);
}

// the tiles of dm_reg_fill_reg, whose per-tile sums dm_export_pyramid<true> finishes (every block re-reads them: bounded)
#define DM_MAX_SUM_PARTS 4096
static int depth_tiles(const ellc_ctx* c) { return ((c->cfg.width + DM_TX - 1) / DM_TX) * ((c->cfg.height + DM_TY - 1) / DM_TY); }
static int export_pyramid_steps(const ellc_ctx* c) {   // levels the export's own launch produces (0: the per-level kernels)
  const int W = c->cfg.width, H = c->cfg.height;
  int steps = 0;
  while (steps < 3 && steps + 1 < c->L && ((W >> steps) & 1) == 0 && ((H >> steps) & 1) == 0) steps++;
  return (steps > 0 && (W % 32) == 0 && (H % 32) == 0) ? steps : 0;   // tiles of 32 x 32 must align with every level's 2 x 2 cells
}
static bool rescale_in_export(const ellc_ctx* c) { return export_pyramid_steps(c) > 0 && depth_tiles(c) <= DM_MAX_SUM_PARTS; }

// with_rescale: makeInvDepthOne in the export's launch, from the per-tile sums the one-launch regularise / fill / regularise left
ellc_status do_update_depth_image(ellc_ctx* c, bool with_rescale = false) {
  const int W = c->cfg.width, H = c->cfg.height;
  const KfLevelDev& k = c->kf_tab_h[c->dm_kf_slot];
  invalidate_records(c, c->dm_kf_slot);   // before the first write (a failure half-way must not leave a valid tag)
  // levels that halve exactly go into the export's own launch (dm_export_pyramid); the rest take the per-level kernel
  int steps = export_pyramid_steps(c);
  if (with_rescale && !(steps > 0 && depth_tiles(c) <= DM_MAX_SUM_PARTS)) return fail(c, ELLC_ERR_BAD_ARG, "rescale in the export needs the tiled export");
  if (steps > 0) {
    ExportPyrArgs ea;
    ea.W = W; ea.H = H; ea.steps = steps;
    for (int l = 0; l < 4; l++) {
      const KfLevelDev& kl = c->kf_tab_h[(size_t)std::min(l, c->L - 1) * c->cfg.max_keyframes + c->dm_kf_slot];
      ea.depth[l] = kl.depth;
      ea.var[l] = kl.var;
    }
    float* factor_d = (float*)(c->red_scratch + 2 * 256 + 2);
    if (with_rescale)
      hipLaunchKernelGGL(dm_export_pyramid<true>, dim3(W / 32, H / 32), dim3(256), 0, c->stream, c->dm_cur, ea, c->sum_parts, depth_tiles(c), factor_d);
    else
      hipLaunchKernelGGL(dm_export_pyramid<false>, dim3(W / 32, H / 32), dim3(256), 0, c->stream, c->dm_cur, ea, (const double*)nullptr, 0, (float*)nullptr);
  } else {
    dim3 blk(32, 8);
    hipLaunchKernelGGL(dm_export_level0, grid2(W, H, blk), blk, 0, c->stream, c->dm_cur, k.depth, k.var, W, H);
  }
  ELLC_HIP(c, hipGetLastError());
  ellc_status s = build_depth_pyramid_from(c, c->dm_kf_slot, steps + 1);   // buildInvVarDepth + mapDepthArr2Mat: the remaining levels
  if (s != ELLC_OK) return s;
  c->kf_has_depth[c->dm_kf_slot] = 1;
  c->kf_dense[c->dm_kf_slot] = 0;   // the map's export is semi-dense
  return enqueue_eager_lists(c, c->dm_kf_slot);   // (a tracking context: the next alignment's lists, off its critical path)
}

// regularizeDepthMap(removeOcclusions) + fillDepthHoles + regularizeDepthMap(false) as createKeyFrame runs them (:1775-1777) in ONE
// launch (dm_reg_fill_reg): the result goes to the other copy of the map, which becomes the map
ellc_status do_reg_fill_reg(ellc_ctx* c, int removeOcclusions, bool with_sums = false) {
  const int W = c->cfg.width, H = c->cfg.height;
  const int tiles_x = (W + DM_TX - 1) / DM_TX, tiles = tiles_x * ((H + DM_TY - 1) / DM_TY);
  hipLaunchKernelGGL(dm_reg_fill_reg, dim3(8 * ((tiles + 7) / 8)), dim3(DM_TX * DM_TY), 0, c->stream, c->dm_cur, c->dm_oth, c->kf_maxgrad[c->dm_kf_slot], W, H,
                     removeOcclusions, tiles_x, tiles, with_sums ? c->sum_parts : (double*)nullptr);
  ELLC_HIP(c, hipGetLastError());
  swap_maps(c);
  return ELLC_OK;
}

// doRegularization (:1627-1635): fillDepthHoles + regularizeDepthMap in one launch (dm_fill_reg), into the other copy of the map
ellc_status do_fill_regularize(ellc_ctx* c, int removeOcclusions, const int* gate = nullptr) {
  const int W = c->cfg.width, H = c->cfg.height;
  const int tiles_x = (W + DM_TX - 1) / DM_TX, tiles = tiles_x * ((H + DM_TY - 1) / DM_TY);
  ExportPyrArgs none;
  for (int l = 0; l < 4; l++) { none.depth[l] = nullptr; none.var[l] = nullptr; }
  none.W = W; none.H = H; none.steps = 0;
  hipLaunchKernelGGL(dm_fill_reg<false>, dim3(8 * ((tiles + 7) / 8)), dim3(DM_TX * DM_TY), 0, c->stream, c->dm_cur, c->dm_oth, c->kf_maxgrad[c->dm_kf_slot], W, H,
                     removeOcclusions, tiles_x, tiles, gate, none);
  ELLC_HIP(c, hipGetLastError());
  swap_maps(c);
  return ELLC_OK;
}

// doRegularization(false) + updateDepthImage of a tracked frame as ONE launch (dm_fill_reg<true>) when the image tiles into 32 x 8
// blocks that halve exactly for the pyramid levels the export's own launch would produce; otherwise the stages one after the other
ellc_status do_fill_regularize_and_update_depth_image(ellc_ctx* c, const int* gate) {
  const int W = c->cfg.width, H = c->cfg.height;
  int steps = 0;
  while (steps < 3 && steps + 1 < c->L && ((W >> steps) & 1) == 0 && ((H >> steps) & 1) == 0) steps++;
  if (!(steps > 0 && (W % DM_TX) == 0 && (H % DM_TY) == 0 && (DM_TX >> steps) >= 1 && (DM_TY >> steps) >= 1)) {
    ellc_status s = do_fill_regularize(c, 0, gate);
    return s != ELLC_OK ? s : do_update_depth_image(c);
  }
  invalidate_records(c, c->dm_kf_slot);   // before the first write (a failure half-way must not leave a valid tag)
  ExportPyrArgs ea;
  ea.W = W; ea.H = H; ea.steps = steps;
  for (int l = 0; l < 4; l++) {
    const KfLevelDev& kl = c->kf_tab_h[(size_t)std::min(l, c->L - 1) * c->cfg.max_keyframes + c->dm_kf_slot];
    ea.depth[l] = kl.depth;
    ea.var[l] = kl.var;
  }
  const int tiles_x = W / DM_TX, tiles = tiles_x * (H / DM_TY);
  hipLaunchKernelGGL(dm_fill_reg<true>, dim3(8 * ((tiles + 7) / 8)), dim3(DM_TX * DM_TY), 0, c->stream, c->dm_cur, c->dm_oth, c->kf_maxgrad[c->dm_kf_slot], W, H, 0,
                     tiles_x, tiles, gate, ea);
  ELLC_HIP(c, hipGetLastError());
  swap_maps(c);
  ellc_status s = build_depth_pyramid_from(c, c->dm_kf_slot, steps + 1);   // the remaining levels
  if (s != ELLC_OK) return s;
  c->kf_has_depth[c->dm_kf_slot] = 1;
  c->kf_dense[c->dm_kf_slot] = 0;   // the map's export is semi-dense
  return enqueue_eager_lists(c, c->dm_kf_slot);   // (a tracking context: the next alignment's lists, off its critical path)
}

ellc_status do_propagate(ellc_ctx* c, int new_kf_slot, const float* pose_new_wrt_old) {
  if (new_kf_slot < 0 || new_kf_slot >= c->cfg.max_keyframes || !pose_new_wrt_old) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (!c->kf_has_image[new_kf_slot]) return fail(c, ELLC_ERR_NOT_READY, "new keyframe slot has no image");
  if (!c->kf_maxgrad_valid[new_kf_slot]) {
    ellc_status s = build_maxgrad(c, true, new_kf_slot);
    if (s != ELLC_OK) return s;
  }
  const int W = c->cfg.width, H = c->cfg.height, n = W * H;
  const RelMats m = relative_matrices(c, pose_new_wrt_old);   // new <- old
  PropArgs a;
  a.src = c->dm_cur;
  a.dst = c->dm_oth;
  a.oldImg = c->kf_tab_h[c->dm_kf_slot].img;
  a.newImg = c->kf_tab_h[new_kf_slot].img;
  a.newMaxGrad = c->kf_maxgrad[new_kf_slot];
  a.W = W; a.H = H; a.sw = c->geom_h[0].sw;
  for (int r = 0; r < 3; r++) {
    for (int q = 0; q < 3; q++) a.R[r * 3 + q] = m.two[r * 4 + q];
    a.t[r] = m.two[r * 4 + 3];
  }
  a.fx = c->cfg.fx; a.fy = c->cfg.fy; a.cx = c->cfg.cx; a.cy = c->cfg.cy;
  a.fxi = c->Kinv[0]; a.cxi = c->Kinv[2]; a.fyi = c->Kinv[4]; a.cyi = c->Kinv[5];
  a.tgt = c->pr_tgt; a.nid = c->pr_id; a.nvar = c->pr_var; a.nval = c->pr_val; a.cnt = c->pr_cnt; a.slots = c->pr_slots;
  dim3 blk(32, 8);
  hipLaunchKernelGGL(dm_prop_project, grid2(W, H, blk), blk, 0, c->stream, a);
  hipLaunchKernelGGL(dm_prop_fold, dim3((n + 255) / 256), dim3(256), 0, c->stream, a, n);
  ELLC_HIP(c, hipGetLastError());
  swap_maps(c);   // std::swap(currentDepthHypothesis, otherDepthHypothesis) (:1154)
  return ELLC_OK;
}

}  // namespace

extern "C" {

ellc_status ellc_depth_set_state(ellc_ctx* c, const ellc_hypotheses* h) {
  ELLC_ENTER(c);
  if (!c || !h || !h->invDepth || !h->invDepthSmoothed || !h->variance || !h->varianceSmoothed || !h->validity_counter || !h->blacklisted || !h->isValid)
    return fail(c, ELLC_ERR_BAD_ARG, "ellc_depth_set_state: null array");
  const size_t n = (size_t)c->cfg.width * c->cfg.height;
  const DepthSoA& d = c->dm_cur;
  ELLC_HIP(c, hipMemcpyAsync(d.invDepth, h->invDepth, n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(d.invDepthSmoothed, h->invDepthSmoothed, n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(d.variance, h->variance, n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(d.varianceSmoothed, h->varianceSmoothed, n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(d.validity, h->validity_counter, n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(d.blacklisted, h->blacklisted, n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(d.isValid, h->isValid, n, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  c->dm_ready = (c->dm_kf_slot >= 0);
  return ELLC_OK;
}

ellc_status ellc_depth_get_state(ellc_ctx* c, const ellc_hypotheses* h) {
  ELLC_ENTER(c);
  if (!c || !h) return fail(c, ELLC_ERR_BAD_ARG, "ellc_depth_get_state: null");
  const size_t n = (size_t)c->cfg.width * c->cfg.height;
  const DepthSoA& d = c->dm_cur;
  if (h->invDepth) ELLC_HIP(c, hipMemcpyAsync(h->invDepth, d.invDepth, n * 4, hipMemcpyDeviceToHost, c->stream));
  if (h->invDepthSmoothed) ELLC_HIP(c, hipMemcpyAsync(h->invDepthSmoothed, d.invDepthSmoothed, n * 4, hipMemcpyDeviceToHost, c->stream));
  if (h->variance) ELLC_HIP(c, hipMemcpyAsync(h->variance, d.variance, n * 4, hipMemcpyDeviceToHost, c->stream));
  if (h->varianceSmoothed) ELLC_HIP(c, hipMemcpyAsync(h->varianceSmoothed, d.varianceSmoothed, n * 4, hipMemcpyDeviceToHost, c->stream));
  if (h->validity_counter) ELLC_HIP(c, hipMemcpyAsync(h->validity_counter, d.validity, n * 4, hipMemcpyDeviceToHost, c->stream));
  if (h->blacklisted) ELLC_HIP(c, hipMemcpyAsync(h->blacklisted, d.blacklisted, n * 4, hipMemcpyDeviceToHost, c->stream));
  if (h->isValid) ELLC_HIP(c, hipMemcpyAsync(h->isValid, d.isValid, n, hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  return ELLC_OK;
}

ellc_status ellc_depth_set_keyframe(ellc_ctx* c, int kf_slot) {
  ELLC_ENTER(c);
  if (!c || kf_slot < 0 || kf_slot >= c->cfg.max_keyframes) return fail(c, ELLC_ERR_BAD_ARG, "bad keyframe slot");
  if (!c->kf_has_image[kf_slot]) return fail(c, ELLC_ERR_NOT_READY, "keyframe slot has no image");
  c->dm_kf_slot = kf_slot;
  c->dm_ready = true;
  return ELLC_OK;
}

ellc_status ellc_depth_propagate(ellc_ctx* c, int new_kf_slot, const float* pose_new_wrt_old) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  return do_propagate(c, new_kf_slot, pose_new_wrt_old);
}

static ObsArgs observe_args(const ellc_ctx* c, int frame_slot) {
  ObsArgs a;
  a.s = c->dm_cur;
  a.kfImg = c->kf_tab_h[c->dm_kf_slot].img;
  a.curImg = c->fr_tab_h[frame_slot].img;
  a.kfMaxGrad = c->kf_maxgrad[c->dm_kf_slot];
  a.W = c->cfg.width; a.H = c->cfg.height; a.sw = c->geom_h[0].sw;
  a.fx = c->cfg.fx; a.fy = c->cfg.fy; a.cx = c->cfg.cx; a.cy = c->cfg.cy;
  a.fxi = c->Kinv[0]; a.cxi = c->Kinv[2]; a.fyi = c->Kinv[4]; a.cyi = c->Kinv[5];
  a.mats = nullptr;
  a.gate = nullptr;
  a.list = c->obs_list;
  a.list_ep = c->obs_list_ep;
  a.ctr = c->obs_ctr + (c->obs_parity ? 2 * DM_OBS_REGIONS : 0);          // this call's counters ...
  a.ctr_next = c->obs_ctr + (c->obs_parity ? 0 : 2 * DM_OBS_REGIONS);     // ... and the next call's (launch_observe alternates)
  {   // a region holds every pixel of the select blocks that append to it (block b -> region b mod DM_OBS_REGIONS)
    const int blocks = ((a.W + 31) / 32) * ((a.H + 7) / 8);
    a.region_cap = ((blocks + DM_OBS_REGIONS - 1) / DM_OBS_REGIONS) * 256;
  }
  return a;
}

// observeDepthRow: candidate selection (a 32 x 8 tile per block), then the line stereo over the work list — a grid for the most
// candidates there can be, the blocks past the list's end leave at once
static void launch_observe(ellc_ctx* c, const ObsArgs& a, bool dev) {
  c->obs_parity ^= 1;   // (observe_args built `a` for the set this call uses; the next call takes the other)
  const dim3 tiles((a.W + 31) / 32, (a.H + 7) / 8), blk(256);
  const int most = std::max(0, a.W - 6) * std::max(0, a.H - 6);
  const dim3 walk(std::max(1, (most + 255) / 256 + 1));   // a wave per chunk of 64 entries of one kind: at most two partial chunks more than most / 64
  RideWeights rw;
  if (dev) {
    dim3 grid = tiles;
    if (c->track_ride_weights) {   // the tracking call's saved weights (enqueue_schedule_persist left them to this launch)
      c->track_ride_weights = false;
      rw.per_level = 128;
      rw.n = rw.per_level * c->L;
      rw.kf_tab = c->kf_tab_d; rw.kf_slot = c->kf_slot_d; rw.geom = c->geom_d; rw.state = c->state_d;
      rw.max_kf = c->cfg.max_keyframes; rw.fast_records = c->fast ? 1 : 0;
      grid.y += (rw.n + tiles.x - 1) / tiles.x;
    }
    hipLaunchKernelGGL(dm_observe_select<true>, grid, blk, 0, c->stream, a, rw);
    hipLaunchKernelGGL(dm_observe_walk<true>, walk, blk, 0, c->stream, a);
  } else {
    hipLaunchKernelGGL(dm_observe_select<false>, tiles, blk, 0, c->stream, a, rw);
    hipLaunchKernelGGL(dm_observe_walk<false>, walk, blk, 0, c->stream, a);
  }
}

static ellc_status do_observe(ellc_ctx* c, int frame_slot, const float* pose_frame_wrt_kf) {
  ObsArgs a = observe_args(c, frame_slot);
  ObsMats m;
  build_obs_mats(c->Kmat, pose_frame_wrt_kf, m);   // observeDepthRowParallel :1935
  for (int i = 0; i < 3; i++) { a.otw_t[i] = m.otw_t[i]; a.Kt[i] = m.Kt[i]; a.tt[i] = m.tt[i]; }
  for (int i = 0; i < 9; i++) { a.Kr[i] = m.Kr[i]; a.Rr[i] = m.Rr[i]; }
  launch_observe(c, a, false);
  ELLC_HIP(c, hipGetLastError());
  return mark_frame_use(c, frame_slot);
}

ellc_status ellc_depth_observe(ellc_ctx* c, int frame_slot, const float* pose_frame_wrt_kf) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  if (frame_slot < 0 || frame_slot >= c->cfg.max_frames || !pose_frame_wrt_kf) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (!c->fr_has_image[frame_slot]) return fail(c, ELLC_ERR_NOT_READY, "frame slot has no image");
  return do_observe(c, frame_slot, pose_frame_wrt_kf);
}

// main.cpp:330 + :499-502 for a frame that does not switch the keyframe, as ONE device sequence: the alignment against the depth
// map's keyframe, then — without the pose travelling to the host and back — observeDepthRowParallel, doRegularization and
// updateDepthImage with the matrices built on the device from the alignment's pose in its finish kernel (track_setup_wave). The host fetches
// the pose while the depth stages run. seeds_percent: calculate_no_of_Seeds (:1804-1830) of the map BEFORE this observation,
// which main.cpp writes beside the pose (main.cpp:368-373).
ellc_status ellc_track_frame(ellc_ctx* c, int frame_slot, const float* init_pose, int save_weights, float* out_pose, int* out_iters,
                             float* out_weighted, float* seeds_percent) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  if (frame_slot < 0 || frame_slot >= c->cfg.max_frames) return fail(c, ELLC_ERR_BAD_ARG, "bad frame slot");
  if (!c->fr_has_image[frame_slot]) return fail(c, ELLC_ERR_NOT_READY, "frame slot has no image");
  if (c->n_inflight > 0) return fail(c, ELLC_ERR_NOT_READY, "ellc_track_frame: fetch the enqueued batches first");
  const int n = c->cfg.width * c->cfg.height;
  if (!c->track_h) {   // host-visible record: [0] valid hypotheses before the observation
    ELLC_HIP(c, hipHostMalloc((void**)&c->track_h, 64, hipHostMallocDefault));
    std::memset(c->track_h, 0, 64);   // ([1], the number of the last count, starts at 0 as the device's counter does)
    c->host_allocs.push_back(c->track_h);
    void* da = nullptr;
    ELLC_HIP(c, hipHostGetDevicePointer(&da, c->track_h, 0));
    c->track_dev_alias = (int*)da;
  }
  const int kf = c->dm_kf_slot;
  // the seeds figure: the count of the valid hypotheses rides along in the alignment's staging launch when that sequence is launched
  // kernel by kernel (stage_in_args); in front of a captured sequence it is a launch of its own
  const bool rides = launches_directly(c, ELLC_MODE_FCA, 1);
  if (!rides)
    hipLaunchKernelGGL(dm_count_valid_block, dim3(std::max(1, ((n >> 4) + 1023) / 1024)), dim3(1024), 0, c->stream, c->dm_cur.isValid, n, c->seed_acc,
                       c->track_dev_alias);
  c->done_deferred = false;
  c->track_ride_weights = false;
  c->track_call = true;   // this alignment's finish kernel builds the observation's matrices and sets the gate
  c->track_count_valid = c->dm_cur.isValid;
  c->track_count_n = rides ? n : 0;
  s = ellc_align_enqueue(c, 1, &kf, &frame_slot, init_pose, ELLC_MODE_FCA, save_weights);   // one batch: it runs on the main stream
  c->track_call = false;
  const bool counted = !rides || c->track_count_n == 0;
  c->track_count_n = 0;
  if (s != ELLC_OK) return s;
  if (!counted) return fail(c, ELLC_ERR_HIP, "ellc_track_frame: the staging launch did not take the count along");
  c->track_counts++;   // (one count per call, whichever launch carried it)
  const int set = c->inflight[0] / ellc_ctx::MAX_COALESCE;
  ellc_ctx::BatchSet& bs = c->batch_set[set];
  // (launch_group may have left the group's `done` event to this call: it is recorded behind the depth stages, on every way out)
  struct DoneGuard {
    ellc_ctx* c; hipEvent_t ev; bool armed;
    ~DoneGuard() { if (armed) (void)hipEventRecord(ev, c->stream); }
  } done_guard{c, bs.done, c->done_deferred};
  c->done_deferred = false;
  if (!bs.launched || bs.stream_idx != 0) return fail(c, ELLC_ERR_HIP, "ellc_track_frame: the alignment did not take the main stream");
  ObsArgs a = observe_args(c, frame_slot);
  a.mats = (ObsMats*)c->track_mats_d;
  a.gate = c->track_gate_d;
  launch_observe(c, a, true);
  ELLC_HIP(c, hipGetLastError());
  // doRegularization(false) :1627-1635 and updateDepthImage (an unchanged map exports the same planes): one launch
  if ((s = do_fill_regularize_and_update_depth_image(c, c->track_gate_d)) != ELLC_OK) return s;
  // the frame slot's "last read" mark goes behind the whole chain, not behind the observation that reads it: an event record in the
  // middle of the chain held the next launch back ~10 us, and the next upload into this slot is a frame away either way
  if ((s = mark_frame_use(c, frame_slot)) != ELLC_OK) return s;
  if (done_guard.armed) {
    done_guard.armed = false;
    ELLC_HIP(c, hipEventRecord(bs.done, c->stream));
  }
  // the pose: waits for the alignment only (its result record, or its event, recorded in front of the depth stages)
  if (wait_batch_results(c, bs) != hipSuccess) return fail(c, ELLC_ERR_HIP, "ellc_track_frame: the alignment failed on the device");
  const bool continued = bs.adaptive && bs.result_h[0].pad == 1;   // the state-driven schedule needs its continuation: the gate stayed closed
  float pose[6];
  s = ellc_align_fetch(c, 1, pose, out_iters, out_weighted);   // (runs the continuation when one is needed)
  if (s != ELLC_OK) return s;
  if (out_pose) std::memcpy(out_pose, pose, sizeof(pose));
  {   // the count of this call (dm_count_valid_body numbers its counts): normally there long before the alignment has ended
    const volatile int* th = (const volatile int*)c->track_h;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; (int)((unsigned)th[1] - (unsigned)c->track_counts) < 0; spin++) {   // (not behind: a call that failed after its count leaves the device ahead)
      if ((spin & 1023u) == 1023u) {
        std::this_thread::yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) return fail(c, ELLC_ERR_HIP, "ellc_track_frame: the count of the valid hypotheses did not arrive");
      }
    }
  }
  if (seeds_percent) *seeds_percent = (float)c->track_h[0] / (float)n * 100;
  if (continued) {   // rare: the depth stages again, the usual way
    float pwo[6];
    const float zero[6] = {0, 0, 0, 0, 0, 0};
    concat_relative_f32(pose, zero, pwo);
    if ((s = do_observe(c, frame_slot, pwo)) != ELLC_OK) return s;
    return do_fill_regularize_and_update_depth_image(c, nullptr);
  }
  return ELLC_OK;
}

ellc_status ellc_depth_fill_holes(ellc_ctx* c) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  return do_fill_holes(c);
}

ellc_status ellc_depth_regularize(ellc_ctx* c, int remove_occlusions) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  return do_regularize(c, remove_occlusions);
}

ellc_status ellc_depth_make_inv_depth_one(ellc_ctx* c, float* rescale_factor) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  return do_rescale(c, rescale_factor);
}

ellc_status ellc_depth_update_depth_image(ellc_ctx* c) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  return do_update_depth_image(c);
}

ellc_status ellc_depth_do_regularization(ellc_ctx* c, int remove_occlusions) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  if (!c->kf_maxgrad_valid[c->dm_kf_slot] && (s = build_maxgrad(c, true, c->dm_kf_slot)) != ELLC_OK) return s;
  return do_fill_regularize(c, remove_occlusions ? 1 : 0);
}

ellc_status ellc_depth_regularize_fill_regularize(ellc_ctx* c, int remove_occlusions) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  if (!c->kf_maxgrad_valid[c->dm_kf_slot] && (s = build_maxgrad(c, true, c->dm_kf_slot)) != ELLC_OK) return s;
  return do_reg_fill_reg(c, remove_occlusions ? 1 : 0);
}

ellc_status ellc_depth_create_keyframe(ellc_ctx* c, int new_kf_slot, const float* pose_new_wrt_old, float* rescale_factor) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  if ((s = do_propagate(c, new_kf_slot, pose_new_wrt_old)) != ELLC_OK) return s;   // :1769 (on failure the map is unchanged)
  const int old_slot = c->dm_kf_slot;
  c->dm_kf_slot = new_kf_slot;                                                     // :1772
  // :1775 regularizeDepthMap(true), :1777 doRegularization(false) = fill + regularise: one launch, which also leaves the per-tile
  // sums of :1779 makeInvDepthOne; the export's launch (:1781 updateDepthImage) finishes that sum and rescales first
  const bool merged = rescale_in_export(c);
  if ((s = do_reg_fill_reg(c, 1, merged)) == ELLC_OK &&
      (merged || (s = do_rescale_enqueue(c)) == ELLC_OK) &&
      (s = do_update_depth_image(c, merged)) == ELLC_OK)
    s = do_rescale_finish(c, rescale_factor);   // the one host wait of the whole sequence: the factor the caller is handed
  if (s != ELLC_OK) {   // a device error part-way: the map no longer matches either keyframe
    c->dm_kf_slot = old_slot;
    c->dm_ready = false;
  }
  return s;
}

#ifdef ELLC_DIAG_ABI
// measurement hook (bench.py): `reps` enqueues of one depth-map stage between two HIP events on the context's stream.
// stage 0: regularizeDepthMap(false), 1: fillDepthHoles, 2: observeDepthRow against frame_slot / pose, 3: updateDepthImage
// (export + depth / variance pyramid), 4: createKeyFrame's regularise + fill + regularise in one launch, 5: the tracked frame's
// fill + regularise + updateDepthImage in one launch. The map keeps evolving from call to call, as it does from frame to frame.
ellc_status ellc_profile_depth_stage(ellc_ctx* c, int stage, int frame_slot, const float* pose_frame_wrt_kf, int reps, float* avg_ms) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  if (reps < 1 || stage < 0 || stage > 5) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  auto once = [&]() -> ellc_status {
    switch (stage) {
      case 0: return do_regularize(c, 0);
      case 1: return do_fill_holes(c);
      case 2: return do_observe(c, frame_slot, pose_frame_wrt_kf);
      case 4: return do_reg_fill_reg(c, 1);
      case 5: return do_fill_regularize_and_update_depth_image(c, nullptr);
      default: return do_update_depth_image(c);
    }
  };
  if ((s = once()) != ELLC_OK) return s;
  ELLC_HIP(c, hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < reps; i++)
    if ((s = once()) != ELLC_OK) return s;
  ELLC_HIP(c, hipEventRecord(c->ev1, c->stream));
  ELLC_HIP(c, hipEventSynchronize(c->ev1));
  float ms = 0;
  ELLC_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if (avg_ms) *avg_ms = ms / reps;
  return ELLC_OK;
}
#endif   // ELLC_DIAG_ABI

ellc_status ellc_depth_seeds(ellc_ctx* c, float* percent) {
  ELLC_ENTER(c);
  ellc_status s = need_map(c);
  if (s != ELLC_OK) return s;
  const int n = c->cfg.width * c->cfg.height;
  ELLC_HIP(c, hipMemsetAsync(c->pr_remaining, 0, 4, c->stream));
  hipLaunchKernelGGL(dm_count_valid, dim3(std::min(128, (n + 255) / 256)), dim3(256), 0, c->stream, c->dm_cur, n, c->pr_remaining);
  ELLC_HIP(c, hipGetLastError());
  int cnt = 0;
  ELLC_HIP(c, hipMemcpyAsync(&cnt, c->pr_remaining, 4, hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  if (percent) *percent = (float)cnt / (float)n * 100;   // count/(W*H)*100 (:1829)
  return ELLC_OK;
}

}  // extern "C"
