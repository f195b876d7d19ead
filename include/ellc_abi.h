/*
 * ellc_abi.h — C ABI of libellc_hip.so, the MI355X (gfx950) implementation of ELLC's per-keyframe
 * data-parallel hot path: pyramidal direct image alignment (Gauss-Newton on se(3)) and the semi-dense
 * inverse-depth map update.
 *
 * The reference has no plugin/FFI layer: the seam is the C++ object API that main.cpp and
 * GlobalOptimize.cpp call (frame, PixelWisePyramid, depthMap, GetImagePoseEstimate).  Each entry point
 * below names the reference member(s) it replaces (file:line under the reference's src/).  The C++
 * facade with the reference's own class/method names lives in ellc_facade.hpp and calls only this ABI.
 *
 * Conventions
 *   - every function returns an ellc_status (0 = ok, negative = error); numeric degeneracy is not an
 *     error (singular 6x6 H => zero update, as cv::Mat::inv returns zeros, PixelWisePyramid.cpp:451).
 *   - all pointers are HOST pointers unless the parameter name ends in _dev.
 *   - pose = 6 f32 [wx wy wz vx vy vz] (rotation first, PixelWisePyramid.cpp:153).
 *   - images are u8 row-major W x H; depth / variance / weights are f32 row-major.
 *   - a context owns its HIP streams (one, plus one per extra batch in flight: ellc_align_enqueue) and behaves as one
 *     in-order queue; calls on one context are serialised by the caller, calls on different contexts are independent
 *     and may come from different threads (the reference's loop-closure thread, GlobalOptimize.cpp:241).
 */
#ifndef ELLC_ABI_H
#define ELLC_ABI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELLC_MAX_LEVELS 8
#define ELLC_ABI_VERSION 10   /* r06: measurement hooks and self-tests moved out (ellc_abi_diag.h) */

typedef enum {
  ELLC_OK = 0,
  ELLC_ERR_BAD_ARG = -1,
  ELLC_ERR_HIP = -2,
  ELLC_ERR_NOT_READY = -3,   /* slot never uploaded, depth map not initialised, ... */
  ELLC_ERR_NO_DEVICE = -4,
  ELLC_ERR_CAPACITY = -5
} ellc_status;

typedef enum {
  ELLC_MODE_FCA = 0,  /* forward-compositional, per-iteration weights: calculatePixelWiseParallel, PixelWisePyramid.cpp:416-455 */
  ELLC_MODE_ICA = 1   /* constant saved weights, template gradient: calculatePixelWiseParallelInvCompositional, :917-974 */
} ellc_mode;

typedef enum {
  ELLC_ARITH_EXACT = 0, /* every per-pixel value is computed with the reference's expression order in IEEE f32 (f64 where its
                         * pow() promotes): bit-identical to the CPU path per pixel; only the order of the 27 sums differs */
  ELLC_ARITH_FAST = 1   /* tolerance mode: fused multiply-adds, hardware reciprocal / rsqrt (1 ulp) for the divisions and the
                         * sqrt, f32 for the double-promoted Jacobian terms, the 6x6 system solved by L D L^T in double, the
                         * pose carried as exp(pose) through the schedule. Per-pixel values agree with EXACT to a few 1e-7
                         * relative, the final pose to <= 1e-5 (tests/test_gpu_fast.py); about half the instructions per pixel.
                         * Requires width, height <= 4096. */
} ellc_arith;

/* Run-time form of the compile-time constants in ExternVariable.h:39-59 and main.cpp:34. */
typedef struct {
  int width, height;            /* ORIG_COLS, ORIG_ROWS */
  int levels;                   /* MAX_PYRAMID_LEVEL (reference: 4) */
  float fx, fy, cx, cy;         /* ORIG_FX, ORIG_FY, ORIG_CX, ORIG_CY */
  int max_iter[ELLC_MAX_LEVELS];/* util::MAX_ITER, index = pyramid level (0 = finest) */
  int early_exit;               /* 1: stop a level when weightedPose < 1 (ImageFunc.cpp:251-252) */
  int max_keyframes;            /* keyframe (reference-side) slots resident in HBM */
  int max_frames;               /* current-frame slots resident in HBM */
  int max_batch;                /* largest B accepted by ellc_align */
  int device;                   /* HIP device ordinal */
  int concurrent_batches;       /* 1..16: how many batches the caller keeps in flight (ellc_align_enqueue); > 1 sizes the
                                 * fine-level grids for sharing the device. Fixed per context, so a batch's result does
                                 * not depend on what else happens to be in flight (the grid fixes the summation order) */
  int arith;                    /* ELLC_ARITH_EXACT (default) or ELLC_ARITH_FAST: arithmetic of the Gauss-Newton pixel pass and solve */
  int coalesce;                 /* 1 (default) .. 4: full batches (B = max_batch) enqueued one after the other are launched side by
                                 * side, up to this many per launch sequence, and 4 x coalesce batches may be in flight. Fixed per
                                 * context: a full batch's grids are those of a full group whether it runs alone or not, so its
                                 * result does not depend on what it was launched with */
  int cache_records;            /* 0 (default): the compact pixel lists of a batch's keyframes are rebuilt by every call, as the
                                 * reference recomputes its masks per alignment (ImageFunc.cpp:158). 1: they are kept with the
                                 * keyframe slot and rebuilt only after the slot's image, depth or weights have changed (every
                                 * entry point that writes them marks the slot); batches that only READ a slot's lists then also
                                 * run concurrently instead of one after the other. Results are identical either way */
  int grid_batch;               /* 0 (default): the launch grids — which fix the order of the 27 sums, i.e. the last bits of a result —
                                 * are chosen for the size of each call's batch. N > 0: for a batch of N, whatever a call's B is.
                                 * A rank that aligns its block of a sharded loop-closure batch (ellc_shard_range) sets the same N on
                                 * every rank (e.g. max_batch): every alignment's result is then bit-identical to what ONE context
                                 * with the same N computes for the whole batch, for every world size (such a context always runs
                                 * the level-bound schedule, also for one or two alignments) */
} ellc_config;

typedef struct ellc_ctx ellc_ctx;

/* ---- lifetime ------------------------------------------------------------------------------------ */
int ellc_abi_version(void);
int ellc_device_count(void);                           /* HIP devices visible to the process (0: none); a launcher maps LOCAL_RANK to cfg.device with it */
void ellc_default_config(ellc_config* cfg, int width, int height, int levels);
ellc_status ellc_ctx_create(const ellc_config* cfg, ellc_ctx** out);
ellc_status ellc_ctx_destroy(ellc_ctx* ctx);
const char* ellc_last_error(const ellc_ctx* ctx);   /* human-readable text of the last failure */
ellc_status ellc_sync(ellc_ctx* ctx);               /* hipStreamSynchronize on the context stream */
/* Diagnostic counters since the context was created (v7; no reference counterpart: they make the library's two ways of waiting for a
 * batch observable to tests). out[i], i < n: ELLC_CTR_* below, 0 beyond ELLC_CTR_COUNT. */
enum {
  ELLC_CTR_POLLED = 0,        /* batches whose result records the host polled successfully (pinned memory, no event wait) */
  ELLC_CTR_POLL_TIMEOUT = 1,  /* polls that ran out (ellc_ctx_set_poll_timeout_us, default 2 ms) and fell back to the event */
  ELLC_CTR_EVENT_WAIT = 2,    /* batches waited for through their event (never polled, or after a timeout) */
  ELLC_CTR_CONTINUATION = 3,  /* continuation graphs of the state-driven early-exit schedule */
  ELLC_CTR_COUNT = 4
};
ellc_status ellc_ctx_counters(ellc_ctx* ctx, long long* out, int n);
ellc_status ellc_ctx_set_poll_timeout_us(ellc_ctx* ctx, int microseconds);
/* cfg.grid_batch for the calls that follow (v8; 0: grids follow each call's B again). The loop-closure batch of
 * globalOptimize::findMatchParallel (GlobalOptimize.cpp:454-646, call :566) has 1 .. 43 candidates, usually a handful: every rank
 * of a sharded run sets the same N — the smallest of {4, 8, 16, 43} that holds the WHOLE batch — before it aligns its block, so the
 * bits stay independent of the world size (ellc_shard_range) while a batch of three candidates no longer runs on grids sized for
 * 43. Batches in flight keep the grids they were launched with. */
ellc_status ellc_ctx_set_grid_batch(ellc_ctx* ctx, int n);
/* The list-free path for dense maps (v10). A keyframe whose level-0 depth plane was uploaded with at least nine tenths of its
 * pixels valid (ellc_keyframe_set_depth counts them) is aligned — FCA, no saved weights, every keyframe of the batch dense — by
 * kernels that read its planes directly instead of compact lists (gn_fca_dense / gn_fca_dense4 in the tolerance mode, gn_fca_dense_x
 * in the exact mode: there every per-pixel value is the list path's bit for bit). The two paths chunk the pixels differently (plane
 * index / list index), so their SUMS differ in the last bits: within the mode's tolerance (pose <= 1e-5 vs the CPU path either
 * way; measured <= 2e-6 between the paths), but NOT bit-identical — and a batch flips to the list path as soon as one of its
 * keyframes is not dense. A caller that compares runs bit for bit (world-size invariance, regression hashes) pins the choice:
 * mode 0 = automatic (default), 1 = always the lists. Batches in flight keep the path they were launched with. */
ellc_status ellc_ctx_set_dense_maps(ellc_ctx* ctx, int mode);
/* How the state-driven schedule (early exit on, one or two alignments per call: the tracking call, main.cpp:330) reaches the device
 * (r05, ABI v9). 1 (default): as ONE resident launch whenever the call finds the context's pipeline empty (with other batches of
 * the context in flight: as mode 0, so that their launches keep overlapping) — the level's blocks stay on the device for the whole
 * schedule and hand their partial sums to each other through tagged records in device memory (gn_fca_persist); a launch whose blocks do not all
 * become resident within ~50 ms (a device shared with more such launches than it holds) is abandoned and the schedule finished with
 * ordinary launches, with the same results as mode 0 from the iteration it had reached. 0: one launch per iteration (as up to
 * ABI v8). 2: test hook — every resident launch is abandoned at its first hand-over. Takes effect with the next call; ends the
 * context's adaptive hint. */
ellc_status ellc_ctx_set_persistent_schedule(ellc_ctx* ctx, int mode);
void* ellc_stream(ellc_ctx* ctx);                   /* the context's hipStream_t (for event timing by callers) */

/* ---- frame side: frame::frame / constructImagePyramids / calculateGradient / buildMaxGradients
 *      (Frame.cpp:78-124, 170-182, 185-285, 618-674) -------------------------------------------------- */
/* Upload a W x H u8 image into a current-frame slot; builds the u8 pyramid on device (cv::pyrDown). */
ellc_status ellc_frame_upload(ellc_ctx* ctx, int slot, const uint8_t* image);
/* Frame ingest as a device pre-pass (frame::frame(VideoCapture), Frame.cpp:45-75, after the decode): BGR 8-bit frame of
 * orig_w x orig_h (= 4 x the context size: DIM_FACTOR, ExternVariable.h:41-43) -> cvtColor BGR2GRAY -> undistort with the
 * 5-coefficient model {k1,k2,p1,p2,k3} and the alpha=0 getOptimalNewCameraMatrix (FLAG_DO_UNDISTORTION, :58-72) -> resize by
 * 1/4 (INTER_LINEAR) -> level 0 of frame slot `slot` + pyramid. fx..cy are the FULL-size intrinsics the reference builds
 * cam_K from (ORIG_FX*INTRINSIC_FACTOR ..., :61). configure builds the fixed-point map once; new_camera4 (optional)
 * receives {fx,fy,cx,cy} of the new camera matrix. The probes (optional, tests) receive the grey value of source pixel
 * (4x+1, 4y+1) [W*H] and the four undistorted source pixels of every output pixel [W*H*4]. */
ellc_status ellc_ingest_configure(ellc_ctx* ctx, int orig_w, int orig_h, float fx, float fy, float cx, float cy,
                                  const float* dist5, int do_undistort, float* new_camera4);
ellc_status ellc_frame_ingest_bgr(ellc_ctx* ctx, int slot, const uint8_t* bgr, uint8_t* gray_probe, uint8_t* undistorted_probe);
/* Same for a keyframe slot (the reference-side / template frame of an alignment). Also builds
 * maxAbsGradient (level 0). */
ellc_status ellc_keyframe_upload(ellc_ctx* ctx, int slot, const uint8_t* image);
/* Copy a frame slot's pyramid into a keyframe slot on device (depthMap::createKeyFrame makes the tracked
 * frame the next keyframe, DepthPropagation.cpp:1772). */
ellc_status ellc_keyframe_from_frame(ellc_ctx* ctx, int kf_slot, int frame_slot);
/* Read back pyramid level `level` of a slot (is_keyframe selects the slot table). out holds
 * stored_w*stored_h bytes; stored sizes follow pyrDown's ceil rule. Any size pointer may be NULL. */
ellc_status ellc_get_image_level(ellc_ctx* ctx, int is_keyframe, int slot, int level, uint8_t* out,
                                 int* stored_w, int* stored_h, int* cols, int* rows);
/* frame::calculateGradient at `level` (Frame.cpp:185-285): gx, gy each rows*cols f32. */
ellc_status ellc_get_gradient(ellc_ctx* ctx, int is_keyframe, int slot, int level, float* gx, float* gy);
/* frame::buildMaxGradients (Frame.cpp:618-674): W*H f32 and the count of pixels >= MIN_ABS_GRAD_DECREASE. */
ellc_status ellc_get_max_gradient(ellc_ctx* ctx, int is_keyframe, int slot, float* out, int* n_substantial);

/* ---- keyframe depth / variance / weights pyramids -------------------------------------------------- */
/* keyFrame->depth (0 = no hypothesis) and depthMap::depthvararrpyr0 (-1 = none), W*H f32 each; builds
 * levels 1.. on device as depthMap::buildInvVarDepth + mapDepthArr2Mat do (DepthPropagation.cpp:1637-1746). */
ellc_status ellc_keyframe_set_depth(ellc_ctx* ctx, int slot, const float* depth0, const float* var0);
/* Explicit per-level override: frame::depth_pyramid[level] and depthMap::depthvararrptr[level]. */
ellc_status ellc_keyframe_set_depth_level(ellc_ctx* ctx, int slot, int level, const float* depth, const float* var);
ellc_status ellc_keyframe_get_depth_level(ellc_ctx* ctx, int slot, int level, float* depth, float* var);
/* frame::weight_pyramid[level] / numWeightsAdded[level] (Frame.h:57,73). */
ellc_status ellc_keyframe_set_weights(ellc_ctx* ctx, int slot, int level, const float* weights, int num_added);
ellc_status ellc_keyframe_get_weights(ellc_ctx* ctx, int slot, int level, float* weights, int* num_added);
/* frame::finaliseWeights (Frame.cpp:678-695). */
ellc_status ellc_keyframe_finalise_weights(ellc_ctx* ctx, int slot);

/* ---- alignment: GetImagePoseEstimate (ImageFunc.cpp:49-315) ----------------------------------------
 * B independent alignments: alignment b aligns frame slot frame_slots[b] to keyframe slot kf_slots[b],
 * starting from init_pose[b] (the relative pose the reference derives at ImageFunc.cpp:106), visiting
 * levels (levels-1 .. 0) with up to max_iter[level] Gauss-Newton iterations each.
 *   mode            ELLC_MODE_FCA or ELLC_MODE_ICA (fromLoopClosure with FLAG_DO_CONST_WEIGHT_POSE_ESTIMATION)
 *   save_weights    1: add the last executed iteration's weights of every level into the keyframe's
 *                   weight_pyramid (PixelWisePyramid::saveWeights(true), :544-549; ImageFunc.cpp:280-288). The weights
 *                   are accumulated per keyframe slot, so with save_weights every alignment of the batch must use a
 *                   different keyframe slot (ELLC_ERR_BAD_ARG otherwise): the reference saves weights on the tracking
 *                   path only, one alignment at a time (ImageFunc.cpp:280)
 *   out_pose        B*6 f32 relative poses
 *   out_iters       B*levels ints: iterations executed per level (may be NULL)
 *   out_weighted    B f32: weightedPose of the last executed iteration (may be NULL)
 */
ellc_status ellc_align(ellc_ctx* ctx, int B, const int* kf_slots, const int* frame_slots, const float* init_pose,
                       int mode, int save_weights, float* out_pose, int* out_iters, float* out_weighted);
/* Asynchronous form: enqueue only. Up to THREE batches may be in flight; each runs on a stream of its own with its own
 * staging, state and result records, so batches in flight execute CONCURRENTLY on the device (the latency-bound coarse
 * iterations of one batch overlap the fine iterations of another) unless they share a keyframe slot, in which case the
 * later one is ordered after the earlier one. A fourth enqueue returns ELLC_ERR_NOT_READY.
 *   With cfg.coalesce = c > 1 the unit that runs is a GROUP: full batches (B = max_batch, same mode, no saved weights)
 * enqueued one after the other are staged side by side and launched as ONE sequence over all their alignments once c of
 * them have arrived — or as soon as the oldest of them is fetched, or any other entry point is called. A launch over 2 x 32
 * alignments costs little more than one over 32, so a caller that keeps the queue full gets more alignments per second
 * (r02: +20...35 %); up to 4 x c batches may be in flight (three groups run concurrently, a fourth is queued behind the
 * oldest). Results are per batch and do not depend on the grouping.
 *   ellc_align_fetch waits for
 * the OLDEST batch in flight only (an event) and returns its results (B must be that batch's size: ELLC_ERR_BAD_ARG
 * otherwise, the batch stays in flight); with nothing in flight it returns
 * ELLC_ERR_NOT_READY. Every other entry point is ordered after the batches in flight and before the batches enqueued
 * later, exactly as if the context had a single in-order queue: an upload into a slot a batch in flight reads takes
 * effect behind that batch.
 *   With cfg.early_exit on, an FCA call of one or two alignments (the tracking call) runs a schedule whose launch sequence is
 * as long as alignments usually need; when one needs more, ellc_align_fetch replays the remainder before it returns. Only
 * the host can start that remainder, so while such a batch is in flight a batch sharing one of its keyframe slots, and
 * every non-batch entry point, first WAITS for it on the host (results and ordering are unchanged). */
ellc_status ellc_align_enqueue(ellc_ctx* ctx, int B, const int* kf_slots, const int* frame_slots, const float* init_pose,
                               int mode, int save_weights);
ellc_status ellc_align_fetch(ellc_ctx* ctx, int B, float* out_pose, int* out_iters, float* out_weighted);

/* One Gauss-Newton iteration at one level, for parity tests (PixelWisePyramid::calculatePixelWiseParallel
 * or ...InvCompositional(iter)): returns H (6x6 row-major), b, delta = -Hinv*b, the updated pose
 * log(exp(delta)*exp(pose)) and weightedPose (:460-491). iter == 0 in ICA mode runs the precompute.
 * planes (optional, may be NULL): 10 f32 planes rows*cols each in this order: residual, weight,
 * warpedX, warpedY, J0..J5 (display_iterationres, display_weightimg, savedWarpedPointsX/Y, steepest descent). */
ellc_status ellc_gn_iterate(ellc_ctx* ctx, int kf_slot, int frame_slot, int level, int mode, int iter, const float* pose,
                            float* H36, float* b6, float* delta6, float* new_pose6, float* weighted, float* planes);

/* The display planes PixelWisePyramid fills beside the residual and the weights in a pass at `level` with `pose`
 * (PixelWisePyramid.cpp:209-225, :275-284; the reference shows display_warpedimg, ImageFunc.cpp:277): where the keyframe has a
 * depth, display_templateimg = the current image (u8), display_2bewarpedimg = the keyframe image (u8), display_origres = their
 * difference before warping, display_warpedimg = the current image interpolated at the warped point (0 outside); 0 where
 * masked. rows*cols elements each; any pointer may be NULL. (display_iterationres, display_weightimg and the saved warped
 * points are planes 0-3 of ellc_gn_iterate.) */
ellc_status ellc_gn_display_planes(ellc_ctx* ctx, int kf_slot, int frame_slot, int level, const float* pose, uint8_t* templateimg,
                                   uint8_t* tobewarpedimg, float* warpedimg, float* origres);

/* se(3) helpers used by callers (frame::concatenateRelativePose Frame.cpp:503-530,
 * frame::concatenateOriginPose :534-562); host-side, no device work. */
void ellc_concatenate_relative_pose(const float* src_1wrt2, const float* src_2wrt3, float* dest_1wrt3);
void ellc_concatenate_origin_pose(const float* src_1wrt0, const float* src_2wrt0, float* dest_1wrt2);
void ellc_se3_exp(const float* pose6, float* T16);
void ellc_se3_log(const float* T16, float* pose6);

/* ---- loop-closure candidate selection support (globalOptimize, GlobalOptimize.cpp) -------------------------
 * calculateImageHistogram (:40-100): 256-bin histogram of the level-0 image, counts as f32 divided by their f32 sum. */
ellc_status ellc_histogram(ellc_ctx* ctx, int is_keyframe, int slot, float* hist256);
/* compareImageHistogram (:116-122) = cv::compareHist(H1, H2, CV_COMP_KL_DIV): sum p*log(p/q) over bins with |p| > DBL_EPSILON,
 * q replaced by 1e-10 when |q| <= DBL_EPSILON. Host-side. */
double ellc_kl_divergence(const float* p, const float* q, int n);
/* pushToArray deep-copies the finished keyframe and its depth map into a ring slot (:185-223): device-to-device copy of a
 * slot's pyramids (keyframe -> keyframe also copies depth / variance / weight pyramids, maxAbsGradient and weight counts). */
ellc_status ellc_copy_slot(ellc_ctx* ctx, int dst_is_keyframe, int dst_slot, int src_is_keyframe, int src_slot);
/* The same between two contexts on one device with the same width / height / levels. The reference hands its loop-closure
 * thread deep copies of the finished keyframe and depth map (new frame(*currentframe), new depthMap(*currentDepthMap),
 * GlobalOptimize.cpp:185-186) and lets it run beside tracking (:241, joined at :161): here the loop-closure ring lives in a
 * context of its own (own streams, own batches) and this call is the deep copy. Ordered on the device: the copy runs behind
 * everything enqueued on src_ctx so far, and src_ctx's later work behind the copy; the host does not wait. The caller
 * serialises it with every other call on either context. */
ellc_status ellc_copy_slot_across(ellc_ctx* dst_ctx, int dst_is_keyframe, int dst_slot, ellc_ctx* src_ctx, int src_is_keyframe, int src_slot);

/* ---- semi-dense depth map: class depthMap (DepthPropagation.cpp) -----------------------------------
 * One depth map per context (the reference's currentDepthMap). State is SoA on device:
 * invDepth, invDepthSmoothed, variance, varianceSmoothed f32; validity_counter, blacklisted i32; isValid u8
 * (DepthHypothesis.h:14-40, live fields). */
typedef struct {
  float* invDepth; float* invDepthSmoothed; float* variance; float* varianceSmoothed;
  int32_t* validity_counter; int32_t* blacklisted; uint8_t* isValid;
} ellc_hypotheses;   /* seven host arrays of W*H elements */

ellc_status ellc_depth_set_state(ellc_ctx* ctx, const ellc_hypotheses* host);
ellc_status ellc_depth_get_state(ellc_ctx* ctx, const ellc_hypotheses* host);
/* depthMap::keyFrame = keyframe slot (main.cpp:232, 307; DepthPropagation.cpp:1772). */
ellc_status ellc_depth_set_keyframe(ellc_ctx* ctx, int kf_slot);
/* depthMap::propagateDepth(new_keyframe) (:1003-1157). new_kf_slot holds the new keyframe's image;
 * pose_new_wrt_old = new_keyframe->poseWrtOrigin (old keyframe is the origin). */
ellc_status ellc_depth_propagate(ellc_ctx* ctx, int new_kf_slot, const float* pose_new_wrt_old);
/* depthMap::observeDepthRowParallel (:1932-1958, :191-263, line stereo :397-885) against frame slot
 * `frame_slot` whose pose w.r.t. the keyframe is pose_frame_wrt_kf (frame::poseWrtOrigin). */
ellc_status ellc_depth_observe(ellc_ctx* ctx, int frame_slot, const float* pose_frame_wrt_kf);
ellc_status ellc_depth_fill_holes(ellc_ctx* ctx);                         /* fillDepthHoles :1317-1400 */
ellc_status ellc_depth_regularize(ellc_ctx* ctx, int remove_occlusions);  /* regularizeDepthMap :1436-1543 */
ellc_status ellc_depth_make_inv_depth_one(ellc_ctx* ctx, float* rescale_factor); /* makeInvDepthOne :1546-1587 */
/* depthMap::doRegularization (DepthPropagation.cpp:1627-1635) = fillDepthHoles + regularizeDepthMap(remove_occlusions) in ONE
 * launch; the state afterwards is that of ellc_depth_fill_holes followed by ellc_depth_regularize(remove_occlusions), bit for
 * bit (v7). */
ellc_status ellc_depth_do_regularization(ellc_ctx* ctx, int remove_occlusions);
/* regularizeDepthMap(remove_occlusions) + fillDepthHoles + regularizeDepthMap(false), the three stencil stages in the order
 * createKeyFrame runs them (DepthPropagation.cpp:1775-1777), in ONE launch; the state afterwards is that of the three calls
 * ellc_depth_regularize(remove_occlusions), ellc_depth_fill_holes, ellc_depth_regularize(0), bit for bit (v7). */
ellc_status ellc_depth_regularize_fill_regularize(ellc_ctx* ctx, int remove_occlusions);
/* depthMap::updateDepthImage (:1254-1315): exports depth / variance pyramids into the keyframe slot
 * (what GetImagePoseEstimate then reads). */
ellc_status ellc_depth_update_depth_image(ellc_ctx* ctx);
/* depthMap::createKeyFrame(new_keyframe) (:1758-1794): propagate, regularise(occlusions), fill+regularise,
 * rescale, export; the new slot becomes the depth map's keyframe. */
ellc_status ellc_depth_create_keyframe(ellc_ctx* ctx, int new_kf_slot, const float* pose_new_wrt_old, float* rescale_factor);
ellc_status ellc_depth_seeds(ellc_ctx* ctx, float* percent);              /* calculate_no_of_Seeds :1804-1830 */
/* One tracked frame that does not switch the keyframe, main.cpp:330 + :499-502, as ONE device sequence: GetImagePoseEstimate
 * (ImageFunc.cpp:49-315; FCA) of frame slot `frame_slot` against the depth map's keyframe from init_pose, then — behind it on the
 * device, the pose never travelling to the host and back in between — observeDepthRowParallel with poseWrtOrigin =
 * concatenateRelativePose(pose, 0) (ImageFunc.cpp:305), doRegularization(false) and updateDepthImage. The call returns once the
 * POSE is there; the depth stages are still running (every later call is ordered behind them). seeds_percent (may be NULL):
 * calculate_no_of_Seeds of the map BEFORE this frame's observation (what main.cpp:368-373 writes beside the pose). Results are
 * those of ellc_align + ellc_depth_seeds + ellc_depth_observe + ellc_depth_fill_holes + ellc_depth_regularize(0) +
 * ellc_depth_update_depth_image. */
ellc_status ellc_track_frame(ellc_ctx* ctx, int frame_slot, const float* init_pose, int save_weights, float* out_pose, int* out_iters,
                             float* out_weighted, float* seeds_percent);

/* ---- multi-GPU: the loop-closure batch sharded over the ranks of one node, one gather of the results ------
 * The batch loop of globalOptimize::findMatchParallel (GlobalOptimize.cpp:480-610) runs B independent alignments; rank r of
 * `world` (one process per GPU) aligns the contiguous block ellc_shard_range gives it on its own context and the ONLY
 * exchange is one gather per batch of 8 floats per alignment: [pose(6), weightedPose, iterations executed]. A communicator
 * is independent of any ellc_ctx. Transports: RCCL (ncclAllGather over xGMI on a stream of its own; libellc_hip.so) and TCP
 * through rank 0 (host memory only: CPU tests of the sharded path, hosts without xGMI; also in libellc_comm.so, which is
 * csrc/ellc_comm.cpp alone). Up to 4 gathers may be outstanding (start / finish), so the exchange of one batch overlaps the
 * kernels of the next; ellc_gather_results = start + finish. Every rank passes the same `total`; n_local must be the size
 * of the rank's block; out8 receives `total` records in global order on every rank. */
typedef struct ellc_comm ellc_comm;
void ellc_shard_range(int total, int world, int rank, int* lo, int* hi);
ellc_status ellc_comm_unique_id(unsigned char* id128);   /* rank 0 creates it (ncclGetUniqueId); the host program hands the 128 bytes to the other ranks */
ellc_status ellc_comm_init_rccl(int device, const unsigned char* id128, int world, int rank, int max_total, ellc_comm** out);
ellc_status ellc_comm_init_tcp(const char* host_ipv4, int port, int world, int rank, int max_total, ellc_comm** out);
/* What the TRANSPORT itself reports, for the self-check of a multi-rank run: transport (1 RCCL, 2 TCP); the number of ranks and this
 * process's rank as RCCL sees them (ncclCommCount / ncclCommUserRank; TCP: the ranks that joined rank 0 / the rank announced) — both
 * init calls fail unless they equal `world` / `rank`; the PCI bus id of the communicator's device (RCCL; "" for TCP). */
ellc_status ellc_comm_info(const ellc_comm* comm, int* transport, int* world_seen, int* rank_seen, char* pci_bus_id, int pci_capacity);
ellc_status ellc_comm_destroy(ellc_comm* comm);
const char* ellc_comm_last_error(const ellc_comm* comm);
ellc_status ellc_gather_start(ellc_comm* comm, int total, const float* local8, int n_local);
ellc_status ellc_gather_finish(ellc_comm* comm, float* out8, int out_capacity);   /* out8 holds out_capacity records: ELLC_ERR_CAPACITY (the gather stays outstanding) if the oldest gather's total is larger */
ellc_status ellc_gather_results(ellc_comm* comm, int total, const float* local8, int n_local, float* out8);

/* (Measurement hooks, device self-tests and test hooks are NOT part of this interface: include/ellc_abi_diag.h declares them and only
 * libellc_hip_diag.so — the same sources built with -DELLC_DIAG_ABI — exports them. libellc_hip.so exports exactly what this header declares.) */

#ifdef __cplusplus
}
#endif
#endif /* ELLC_ABI_H */
