// ellc_facade.hpp — the reference's object API (frame / PixelWisePyramid / depthMap / GetImagePoseEstimate)
// re-expressed over the C ABI of libellc_hip.so, so a main.cpp-shaped driver keeps its structure.
// Header-only, host C++11; every numeric operation is a call into include/ellc_abi.h (no CPU fallback).
//
// Reference members mirrored (file:line under the reference's src/):
//   frame                     Frame.h:35-397, Frame.cpp:34-124 (constructor after decode/undistort/resize), :503-562, :678-695
//   PixelWisePyramid          PixelWisePyramid.h:89-101, PixelWisePyramid.cpp:14-51, :416-491, :500-552, :917-974
//   depthMap                  DepthPropagation.h:82-132, DepthPropagation.cpp:44-66, :83-184, :1254-1315, :1627-1635,
//                             :1749-1802, :1804-1830, :1932-1958
//   GetImagePoseEstimate      ImageFunc.h:31, ImageFunc.cpp:49-315
// Differences that are deliberate: image decode / undistort / VideoCapture stay with the caller (frame takes
// the grey W x H image); the process-wide globals of the reference (numberOfInstances, util::K*, GLOABL_DEPTH_SCALE)
// live in ellc::Runtime; failures throw std::runtime_error carrying ellc_last_error() instead of being ignored.
#ifndef ELLC_FACADE_HPP
#define ELLC_FACADE_HPP

#include "ellc_abi.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace ellc {

// Text checkpoint format of the reference (Frame.cpp:697-871): default ostream formatting of float (6 significant digits,
// %g style), one blank after every value; a Mat ends each row with '\n', an array is a single line. Host-only helpers.
namespace text {
inline void writeMat(const std::string& path, const float* data, int width, int height) {
  std::ofstream f(path.c_str());
  if (!f) throw std::runtime_error("saveMatAsText: cannot open " + path);
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) f << data[(size_t)y * width + x] << " ";
    f << "\n";
  }
}
inline void writeArray(const std::string& path, const float* data, size_t n) {
  std::ofstream f(path.c_str());
  if (!f) throw std::runtime_error("saveArrayAsText: cannot open " + path);
  for (size_t i = 0; i < n; i++) f << data[i] << " ";
}
// whitespace-separated values, rows and single lines alike (makeMatFromText / makeArrayFromText, Frame.cpp:738-810): the
// same `stream >> element` per entry, so entries a short file does not hold are left as they were
inline void readValues(const std::string& path, float* data, size_t n) {
  std::ifstream f(path.c_str());
  if (!f) throw std::runtime_error("makeMatFromText: cannot open " + path);
  for (size_t i = 0; i < n; i++) f >> data[i];
}
}  // namespace text

// Owner of the context and of the slot bookkeeping (keyframe slots: active + incoming; frame slots: ring).
class Runtime {
 public:
  ellc_config cfg;
  ellc_ctx* ctx = nullptr;
  int numberOfInstances = 0;      // Frame.cpp:24
  bool FLAG_DO_LOOP_CLOSURE = false;           // "LC" mode: save / average the tracking weights (ExternVariable.h:203,211)
  int BATCH_START_ID = 1;
  // text checkpoints of the active keyframe for the alternate Gauss-Newton <-> rotation-averaging loop (ToggleFlags.h:108-203)
  bool FLAG_ALTERNATE_GN_RA = false;           // file ids become frameId + BATCH_START_ID - 1 (Frame.cpp:700-704)
  bool FLAG_SAVE_MATS = false;                 // write <id>_{Depth,Depth_pyr0,DepthVarArr_pyr0}.txt at keyframe switches (ImageFunc.cpp:73-87)
  bool FLAG_REPLICATE_POSE_ESTIMATION = false; // read them back instead (ImageFunc.cpp:58-66)
  int KEYFRAME_PROPAGATE_INTERVAL = 8;         // ExternVariable.h:39
  std::string SAVED_MATS_PATH = ".";
  explicit Runtime(const ellc_config& c) : cfg(c) {
    if (cfg.max_frames < 3) cfg.max_frames = 3;
    if (cfg.max_keyframes < 2) cfg.max_keyframes = 2;
    if (cfg.max_batch < 1) cfg.max_batch = 1;
    check(ellc_ctx_create(&cfg, &ctx), "ellc_ctx_create");
  }
  ~Runtime() { if (ctx) ellc_ctx_destroy(ctx); }
  Runtime(const Runtime&) = delete;
  Runtime& operator=(const Runtime&) = delete;
  void check(int st, const char* what) const {
    if (st != ELLC_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(st) + "): " + (ctx ? ellc_last_error(ctx) : "no context"));
  }
  // Frame.cpp:45-75 after the decode: BGR -> grey -> undistort -> 1/4 resize as a device pre-pass. fx..cy are the full-size
  // intrinsics of cam_K (util::ORIG_FX*INTRINSIC_FACTOR ...), dist5 = util::distortion_parameters.
  bool ingest_configured = false;
  void configureIngest(int orig_w, int orig_h, float fx, float fy, float cx, float cy, const float* dist5, bool FLAG_DO_UNDISTORTION = true) {
    check(ellc_ingest_configure(ctx, orig_w, orig_h, fx, fy, cx, cy, dist5, FLAG_DO_UNDISTORTION ? 1 : 0, nullptr), "ellc_ingest_configure");
    ingest_configured = true;
  }
  // Multi-GPU (one process per GPU, every process tracks the whole sequence): the loop-closure batch is sharded over the
  // ranks and its results gathered once per batch (ellc_gather_results). comm == nullptr: single process.
  ellc_comm* comm = nullptr;
  int world = 1, rank = 0;
  // The loop-closure context's launch grids are fixed per batch — for the smallest of {4, 8, 16, 43} candidates that holds the WHOLE
  // batch (ellc_ctx_set_grid_batch before every batch, the same value on every rank) — so that the files a run writes do not depend
  // on how many ranks shared the batch, including one (ellc_main --world 2 equals the single process byte for byte), while the usual
  // batch of a handful of candidates no longer runs on grids sized for the ring's 43 (r03 advisor finding; r04 fixed them at 43).
  // A single process that does not need that may clear this before it constructs globalOptimize: grids then follow each call's B.
  bool lc_fixed_grids = true;
  int frame_ring = 3;             // tracking uses frame slots [0, frame_ring): current, t-1 and one spare
  int next_frame_slot() { int s = frame_cursor_; frame_cursor_ = (frame_cursor_ + 1) % frame_ring; return s; }
  int other_keyframe_slot(int current) const { return (current + 1) % 2; }   // keyframe slots 0 / 1: active and incoming
 private:
  int frame_cursor_ = 0;
};

class frame {
 public:
  Runtime* rt;
  int frameId;
  int parentKeyframeId = 0;
  bool isKeyframe = false;
  int width, height;
  int slot;                 // frame slot holding the u8 pyramid on the device
  int kf_slot = -1;         // keyframe slot once this frame has become a keyframe
  float poseWrtOrigin[6];   // w.r.t. the keyframe
  float poseWrtWorld[6];    // w.r.t. the first frame
  float rescaleFactor = 1.0f;
  int numWeightsAdded[ELLC_MAX_LEVELS];

  // Frame.cpp:34-124 from "width=image.cols" on: ids, zero poses, pyramids (constructImagePyramids on device)
  // `pixels` is the grey W x H image, or — with from_bgr — the decoded full-size BGR frame (Runtime::configureIngest first)
  frame(Runtime& r, const uint8_t* pixels, bool from_bgr = false) : rt(&r) {
    frameId = ++r.numberOfInstances;
    width = r.cfg.width;
    height = r.cfg.height;
    for (int i = 0; i < 6; i++) poseWrtOrigin[i] = poseWrtWorld[i] = 0.0f;
    for (int i = 0; i < ELLC_MAX_LEVELS; i++) numWeightsAdded[i] = 0;
    slot = r.next_frame_slot();
    if (from_bgr) r.check(ellc_frame_ingest_bgr(r.ctx, slot, pixels, nullptr, nullptr), "ellc_frame_ingest_bgr");
    else r.check(ellc_frame_upload(r.ctx, slot, pixels), "ellc_frame_upload");
  }
  void concatenateRelativePose(const float* src_1wrt2, const float* src_2wrt3, float* dest_1wrt3) const {
    ellc_concatenate_relative_pose(src_1wrt2, src_2wrt3, dest_1wrt3);
  }
  void concatenateOriginPose(const float* src_1wrt0, const float* src_2wrt0, float* dest_1wrt2) const {
    ellc_concatenate_origin_pose(src_1wrt0, src_2wrt0, dest_1wrt2);
  }
  void calculatePoseWrtOrigin(frame* prev_image, const float* poseChange) { concatenateRelativePose(poseChange, prev_image->poseWrtOrigin, poseWrtOrigin); }
  void calculatePoseWrtWorld(frame* prev_image, const float* poseChange) { concatenateRelativePose(poseChange, prev_image->poseWrtWorld, poseWrtWorld); }
  // ---- text checkpoints (Frame.cpp:697-871). `depth` and `depth_pyramid[0]` are one plane here (level 0 of the keyframe
  // slot); the variance array is depthMap::depthvararrpyr0 (DepthPropagation.cpp:1637-1746), the slot's level-0 variance.
  // Format: default ostream formatting of float (6 significant digits), one blank after every value; a Mat ends each
  // row with '\n', an array is a single line.
  std::string matFileName(const std::string& name, const std::string& dir) const {
    const int id = rt->FLAG_ALTERNATE_GN_RA ? (frameId + rt->BATCH_START_ID - 1) : frameId;
    std::stringstream ss;
    ss << dir << "/" << id << "_" << name << ".txt";
    return ss.str();
  }
  void level0(std::vector<float>& depth, std::vector<float>& var) const {
    if (kf_slot < 0) throw std::runtime_error("text checkpoint: frame is not a keyframe");
    depth.assign((size_t)width * height, 0.f);
    var.assign((size_t)width * height, 0.f);
    rt->check(ellc_keyframe_get_depth_level(rt->ctx, kf_slot, 0, depth.data(), var.data()), "ellc_keyframe_get_depth_level");
  }
  void saveMatAsText(const std::string& name, const std::string& save_mat_path) const {   // name: "Depth" | "Depth_pyr0"
    std::vector<float> d, v;
    level0(d, v);
    text::writeMat(matFileName(name, save_mat_path), d.data(), width, height);
  }
  void saveArrayAsText(const std::string& name, const std::string& save_arr_path, int pyr_level) const {   // "DepthVarArr_pyr0"
    if (pyr_level != 0) throw std::runtime_error("saveArrayAsText: only level 0 is checkpointed (ImageFunc.cpp:84)");
    std::vector<float> d, v;
    level0(d, v);
    text::writeArray(matFileName(name, save_arr_path), v.data(), v.size());
  }
  void makeMatFromText(const std::string& name, const std::string& read_txt_path) {
    std::vector<float> d, v;
    level0(d, v);
    text::readValues(matFileName(name, read_txt_path), d.data(), d.size());
    rt->check(ellc_keyframe_set_depth_level(rt->ctx, kf_slot, 0, d.data(), v.data()), "ellc_keyframe_set_depth_level");
  }
  void makeArrayFromText(const std::string& name, const std::string& read_txt_path, int pyr_level) {
    if (pyr_level != 0) throw std::runtime_error("makeArrayFromText: only level 0 is checkpointed (ImageFunc.cpp:65)");
    std::vector<float> d, v;
    level0(d, v);
    text::readValues(matFileName(name, read_txt_path), v.data(), v.size());
    rt->check(ellc_keyframe_set_depth_level(rt->ctx, kf_slot, 0, d.data(), v.data()), "ellc_keyframe_set_depth_level");
  }
  void finaliseWeights() {   // Frame.cpp:678-695
    if (kf_slot < 0) throw std::runtime_error("finaliseWeights: frame is not a keyframe");
    rt->check(ellc_keyframe_finalise_weights(rt->ctx, kf_slot), "ellc_keyframe_finalise_weights");
  }
};

class depthMap {
 public:
  Runtime* rt;
  frame* keyFrame = nullptr;
  frame* currentFrame = nullptr;
  float depthScale = 1.0f;
  explicit depthMap(Runtime& r) : rt(&r) {}

  // DepthPropagation.cpp:44-66
  void formDepthMap(frame* image_frame) {
    currentFrame = image_frame;
    currentFrame->isKeyframe = false;
    if (image_frame->frameId == 1) {
      keyFrame = image_frame;
      keyFrame->isKeyframe = true;
      keyFrame->rescaleFactor = 1.0f;
      keyFrame->kf_slot = 0;
      rt->check(ellc_keyframe_from_frame(rt->ctx, 0, image_frame->slot), "ellc_keyframe_from_frame");
      rt->check(ellc_depth_set_keyframe(rt->ctx, 0), "ellc_depth_set_keyframe");
      initializeRandomly();
    }
    currentFrame->parentKeyframeId = keyFrame->frameId;
    currentFrame->rescaleFactor = keyFrame->rescaleFactor;
  }
  // DepthPropagation.cpp:83-184 (random branch): glibc rand(), unseeded, raster order over the interior
  void initializeRandomly() {
    const int W = rt->cfg.width, H = rt->cfg.height;
    const size_t n = (size_t)W * H;
    std::vector<float> mg(n), id(n, 0.f), ids(n, 0.f), var(n, 0.f), vars(n, 0.f);
    std::vector<int32_t> val(n, 0), bl(n, 0);
    std::vector<uint8_t> ok(n, 0);
    rt->check(ellc_get_max_gradient(rt->ctx, 1, keyFrame->kf_slot, mg.data(), nullptr), "ellc_get_max_gradient");
    // The reference never seeds rand(), i.e. it consumes glibc's default sequence (seed 1) from its start. The GPU
    // runtime may draw from rand() while it initialises, so the default sequence is re-established explicitly.
    srand(1);
    for (int y = 1; y < H - 1; y++)
      for (int x = 1; x < W - 1; x++) {
        const size_t i = (size_t)x + (size_t)W * y;
        if (mg[i] > 1.0 * 1.0f) {   // MIN_ABS_GRAD_CREATE
          id[i] = 0.5f + 1.0f * ((rand() % 100001) / 100000.0f);
          var[i] = 0.125f;          // VAR_RANDOM_INIT_INITIAL
          vars[i] = 0.125f;
          ids[i] = id[i];
          val[i] = 20;
          ok[i] = 1;
        }
      }
    ellc_hypotheses h = {id.data(), ids.data(), var.data(), vars.data(), val.data(), bl.data(), ok.data()};
    rt->check(ellc_depth_set_state(rt->ctx, &h), "ellc_depth_set_state");
  }
  void updateDepthImage(bool = false) { rt->check(ellc_depth_update_depth_image(rt->ctx), "ellc_depth_update_depth_image"); }
  void doRegularization(bool removeOcclusions = false) {   // :1627-1635
    rt->check(ellc_depth_do_regularization(rt->ctx, removeOcclusions ? 1 : 0), "ellc_depth_do_regularization");   // fill + regularise: one launch
  }
  void finaliseKeyframe() { doRegularization(); updateDepthImage(); }   // :1749-1755
  void updateKeyFrame() {}   // :1796-1802: the Sim3-scaled pose it computes is overwritten before use (:1935)
  void observeDepthRowParallel() {   // :1932-1958
    rt->check(ellc_depth_observe(rt->ctx, currentFrame->slot, currentFrame->poseWrtOrigin), "ellc_depth_observe");
  }
  void createKeyFrame(frame* new_keyframe) {   // :1758-1794
    const int ns = rt->other_keyframe_slot(keyFrame->kf_slot);
    rt->check(ellc_keyframe_from_frame(rt->ctx, ns, new_keyframe->slot), "ellc_keyframe_from_frame");
    float f = 1.0f;
    rt->check(ellc_depth_create_keyframe(rt->ctx, ns, new_keyframe->poseWrtOrigin, &f), "ellc_depth_create_keyframe");
    new_keyframe->kf_slot = ns;
    keyFrame = new_keyframe;
    keyFrame->isKeyframe = true;
    keyFrame->rescaleFactor = f;
    depthScale = f;
    for (int i = 0; i < 6; i++) keyFrame->poseWrtOrigin[i] = 0.0f;
  }
  float calculate_no_of_Seeds(bool = true) {   // :1804-1830
    float p = 0;
    rt->check(ellc_depth_seeds(rt->ctx, &p), "ellc_depth_seeds");
    return p;
  }
};

// One pyramid level of the Gauss-Newton loop, one iteration per call (PixelWisePyramid.cpp:416-491, :917-974).
// The batched fast path is GetImagePoseEstimate below; this class exists for callers that drive single steps.
class PixelWisePyramid {
 public:
  Runtime* rt;
  frame* prev_frame;
  frame* current_frame;
  depthMap* currentDepthMap;
  float* pose;            // caller-owned 6-vector, updated in place (ImageFunc.cpp:183)
  int pyrlevel;
  float weightedPose = 0;
  float hessian[36], sd_param[6], deltapose[6];
  float prevPose[6];
  PixelWisePyramid(frame* prev, frame* cur, float* pose_, depthMap* dm, int level)
      : rt(prev->rt), prev_frame(prev), current_frame(cur), currentDepthMap(dm), pose(pose_), pyrlevel(level) {}
  void putPreviousPose(frame* tminus1) { tminus1->concatenateOriginPose(tminus1->poseWrtWorld, prev_frame->poseWrtWorld, prevPose); }
  void calculatePixelWiseParallel() { step(ELLC_MODE_FCA, 0); }
  void calculatePixelWiseParallelInvCompositional(int iter) { step(ELLC_MODE_ICA, iter); }
 private:
  void step(int mode, int iter) {
    float np[6];
    rt->check(ellc_gn_iterate(rt->ctx, prev_frame->kf_slot, current_frame->slot, pyrlevel, mode, iter, pose, hessian, sd_param, deltapose, np,
                              &weightedPose, nullptr), "ellc_gn_iterate");
    std::memcpy(pose, np, sizeof(np));
  }
};

// ImageFunc.cpp:49-315. The level / iteration loops run on the device (ellc_align); weights of the last executed
// iteration of every level are saved into the keyframe when the runtime is in LC mode and the call does not come
// from loop closure (ImageFunc.cpp:280-288).
namespace detail {
// ImageFunc.cpp:58-131: the text checkpoints at a keyframe switch and the initial relative pose of the alignment
inline void initial_pose_estimate(frame* prev_frame, frame* current_frame, frame* tminus1_prev_frame, float* initial_pose_estimate, bool fromLoopClosure, float* pose) {
  Runtime* rt = prev_frame->rt;
  if (prev_frame->kf_slot < 0) throw std::runtime_error("GetImagePoseEstimate: prev_frame is not a keyframe");
  const bool kf_switch = !fromLoopClosure && (current_frame->frameId % rt->KEYFRAME_PROPAGATE_INTERVAL == 0);
  if (rt->FLAG_REPLICATE_POSE_ESTIMATION && kf_switch) {   // :58-66 revive the keyframe's level-0 depth / variance
    prev_frame->makeMatFromText("Depth", rt->SAVED_MATS_PATH);
    prev_frame->makeMatFromText("Depth_pyr0", rt->SAVED_MATS_PATH);
    prev_frame->makeArrayFromText("DepthVarArr_pyr0", rt->SAVED_MATS_PATH, 0);
  }
  if (rt->FLAG_SAVE_MATS && kf_switch && (rt->FLAG_ALTERNATE_GN_RA || current_frame->frameId < 50)) {   // :73-87
    prev_frame->saveMatAsText("Depth", rt->SAVED_MATS_PATH);
    prev_frame->saveMatAsText("Depth_pyr0", rt->SAVED_MATS_PATH);
    prev_frame->saveArrayAsText("DepthVarArr_pyr0", rt->SAVED_MATS_PATH, 0);
  }
  prev_frame->concatenateOriginPose(tminus1_prev_frame->poseWrtWorld, prev_frame->poseWrtWorld, pose);   // :106
  if (initial_pose_estimate) {
    // FLAG_INITIALIZE_NONZERO_POSE (:109-131): the given world pose (so3poses7.txt) is converted to a pose w.r.t. the
    // keyframe and supplies the rotation; the translation stays the one predicted from frame t-1
    float from_file[6];
    prev_frame->concatenateOriginPose(initial_pose_estimate, prev_frame->poseWrtWorld, from_file);
    pose[0] = from_file[0];
    pose[1] = from_file[1];
    pose[2] = from_file[2];
  }
}
}  // namespace detail

inline std::vector<float> GetImagePoseEstimate(frame* prev_frame, frame* current_frame, int /*frame_num*/, depthMap* /*currDepthMap*/,
                                               frame* tminus1_prev_frame, float* initial_pose_estimate, bool fromLoopClosure = false,
                                               bool /*homo*/ = false) {
  Runtime* rt = prev_frame->rt;
  float pose[6];
  detail::initial_pose_estimate(prev_frame, current_frame, tminus1_prev_frame, initial_pose_estimate, fromLoopClosure, pose);
  const int save = (rt->FLAG_DO_LOOP_CLOSURE && !fromLoopClosure) ? 1 : 0;
  float out[6];
  rt->check(ellc_align(rt->ctx, 1, &prev_frame->kf_slot, &current_frame->slot, pose, fromLoopClosure ? ELLC_MODE_ICA : ELLC_MODE_FCA, save, out,
                       nullptr, nullptr), "ellc_align");
  if (save) for (int l = 0; l < rt->cfg.levels; l++) prev_frame->numWeightsAdded[l]++;
  current_frame->calculatePoseWrtOrigin(prev_frame, out);   // :305
  current_frame->calculatePoseWrtWorld(prev_frame, out);    // :306
  return std::vector<float>(out, out + 6);
}

// A tracked frame that does not switch the keyframe — main.cpp:330 (GetImagePoseEstimate), :368 (calculate_no_of_Seeds) and
// :499-502 (updateKeyFrame, observeDepthRowParallel, doRegularization, updateDepthImage) — as ONE device sequence
// (ellc_track_frame): the depth stages start behind the alignment's last kernel without waiting for the host. prev_frame must be
// the depth map's keyframe. *seeds_num: the seeds figure of the map before this frame's observation, as main.cpp writes it.
inline std::vector<float> TrackFrameAndObserve(frame* prev_frame, frame* current_frame, depthMap* currDepthMap, frame* tminus1_prev_frame,
                                               float* initial_pose_estimate, float* seeds_num) {
  Runtime* rt = prev_frame->rt;
  if (currDepthMap->keyFrame != prev_frame) throw std::runtime_error("TrackFrameAndObserve: prev_frame is not the depth map's keyframe");
  float pose[6];
  detail::initial_pose_estimate(prev_frame, current_frame, tminus1_prev_frame, initial_pose_estimate, false, pose);
  const int save = rt->FLAG_DO_LOOP_CLOSURE ? 1 : 0;
  float out[6], seeds = 0;
  rt->check(ellc_track_frame(rt->ctx, current_frame->slot, pose, save, out, nullptr, nullptr, &seeds), "ellc_track_frame");
  if (save) for (int l = 0; l < rt->cfg.levels; l++) prev_frame->numWeightsAdded[l]++;
  current_frame->calculatePoseWrtOrigin(prev_frame, out);   // :305
  current_frame->calculatePoseWrtWorld(prev_frame, out);    // :306
  if (seeds_num) *seeds_num = seeds;
  currDepthMap->formDepthMap(current_frame);   // main.cpp:391 (bookkeeping only beyond the first frame)
  return std::vector<float>(out, out + 6);
}

// Local loop-closure detection (GlobalOptimize.h:39-97, GlobalOptimize.cpp). Keeps a ring of finished keyframes
// (image, depth pyramid, averaged tracking weights) resident on the device, selects candidates for a test keyframe by
// intensity-histogram KL divergence, frame gap and view angle (findMatch :274-416), and aligns the test keyframe against
// ALL selected candidates with ONE batched constant-weight alignment (the reference runs them one by one, :566).
// The candidate sequence does not depend on the alignment results (the reference restores the test frame's poses after
// every match, :591-606), so collecting first and aligning once is equivalent.
//   As in the reference with FLAG_DO_PARALLEL_SHORT_LOOP_CLOSURE (on in LC mode, ToggleFlags.h:53-59), the matching of a pushed
// keyframe runs on a thread of its own beside tracking (t_group.create_thread, :241) and is joined at the next pushToArray
// (:161) — and when the object goes away. The ring lives in a Runtime of its own (`ring`: a second context with its own
// streams, so its batch overlaps the tracking context's work on the device); pushToArray deep-copies the finished keyframe
// into it (ellc_copy_slot_across = new frame(*currentframe) / new depthMap(*currentDepthMap), :185-186) and the thread only
// ever touches the ring context. The ring context fixes its launch grids per batch (ellc_ctx_set_grid_batch; Runtime::lc_fixed_grids):
// a rank's shard of a batch has the bits of the whole batch.
class globalOptimize {
 public:
  static const int MAX_LOOP_ARRAY_LENGTH = 20;                                   // ExternVariable.h:161
  static const int MAX_LOOP_ARRAY_LENGTH_SCALE_AVG = MAX_LOOP_ARRAY_LENGTH * 2 + 3;   // :162
  struct loopFrame {   // LoopFrame.h:24-39
    float image_histogram[256];
    int frameId = -1;
    float poseWrtWorld[6];
    float poseWrtOrigin[6];
    bool isValid = false;
    float rescaleFactor = 1.0f;   // this_frame->rescaleFactor
    float seeds = 0.0f;           // this_currentDepthMap->calculate_no_of_Seeds()
    int kf_slot = -1;             // ring-context slot holding this_frame / this_currentDepthMap
    bool isStray = false;         // pushed by findConnection: an image without a depth map (LoopFrame.h)
  };
  Runtime* rt;            // the tracking runtime (source of the finished keyframes; its comm / world / rank shard the batch)
  Runtime ring;           // the loop-closure context: keyframe slots [0, 43) = the ring, frame slot 0 = the test keyframe
  bool FLAG_DO_PARALLEL_SHORT_LOOP_CLOSURE = true;
  loopFrame loopFrameArray[MAX_LOOP_ARRAY_LENGTH_SCALE_AVG];
  loopFrame currentLoopFrame;
  std::ofstream match_file;
  bool isloopClosureDetected = false;
  bool connectionLost = false;
  int loopClosureArrayId = -1, lastTestedLoopClosureArrayId = -1, firstTestedLoopClosureArrayId = -1;
  int currentArrayId = 0, nextArrayId = 1;
  int match_window_beg = 0, match_window_end = MAX_LOOP_ARRAY_LENGTH - 1;
  float matchValue = 0, rms_error = 0, relative_view_angle = 0;

  static ellc_config ring_config(const ellc_config& tracking, bool fixed_grids) {
    ellc_config c = tracking;
    c.max_keyframes = MAX_LOOP_ARRAY_LENGTH_SCALE_AVG;
    c.max_frames = 1;
    c.max_batch = MAX_LOOP_ARRAY_LENGTH_SCALE_AVG;
    c.grid_batch = fixed_grids ? c.max_batch : 0;   // world-size invariant bits (ellc_abi.h; set per batch: lc_grid_bucket)
    c.concurrent_batches = 1;
    c.coalesce = 1;
    c.cache_records = 1;          // the ring's keyframes stay from push to push: only the slot a push replaces has its pixel lists rebuilt (same results)
    return c;
  }
  globalOptimize(Runtime& r, const std::string& matchfilepath)
      : rt(&r), ring(ring_config(r.cfg, r.lc_fixed_grids || r.world > 1)), fixed_grids_(r.lc_fixed_grids || r.world > 1) {
    ring.BATCH_START_ID = r.BATCH_START_ID;
    match_file.open(matchfilepath.c_str());
  }
  // the grids of a loop-closure batch of B candidates: the smallest of {4, 8, 16, ring} that holds the whole batch
  static int lc_grid_bucket(int B) {
    const int buckets[4] = {4, 8, 16, MAX_LOOP_ARRAY_LENGTH_SCALE_AVG};
    for (int k = 0; k < 4; k++)
      if (B <= buckets[k]) return buckets[k];
    return MAX_LOOP_ARRAY_LENGTH_SCALE_AVG;
  }

  // ---- tracking-loss recovery (FLAG_RESTORE_CONNECTION, ExternVariable.h:176: off as shipped; main.cpp:252-324) -------------------
  // GlobalOptimize.cpp:934-943: the connection is lost when the depth map has no seeds left (MIN_SEEDS_FOR_CONNECTION_LOST = 0, ExternVariable.h:171)
  static constexpr float MIN_SEEDS_FOR_CONNECTION_LOST = 0.0f;
  depthMap* temp_depthMap = nullptr;   // the depth map a recovered connection would hand to main (main.cpp:270); never set as shipped
  void checkConnection(depthMap* currentDepthMap) { connectionLost = currentDepthMap->calculate_no_of_Seeds() <= MIN_SEEDS_FOR_CONNECTION_LOST; }
  // GlobalOptimize.cpp:717-760 AS SHIPPED: waits for the match thread, pushes the stray test frame (no depth map) into the ring at
  // currentArrayId — histogram, ids, image — and returns: the search that follows in the source sits behind an unconditional return
  // (:759), so connectionLost stays as checkConnection left it and temp_depthMap stays null.
  void findConnection(frame* testFrame) {
    join_all();   // :725 t_group.join_all()
    loopFrame& slot = loopFrameArray[currentArrayId];
    if (slot.isValid) { slot.isValid = false; slot.isStray = false; slot.frameId = 0; }   // resetArrayElement (:124-147)
    slot.kf_slot = currentArrayId;
    ring.check(ellc_copy_slot_across(ring.ctx, 1, slot.kf_slot, rt->ctx, 0, testFrame->slot), "ellc_copy_slot_across");   // this_frame = new frame(*testFrame): the image
    TestFrame test;
    test.frameId = testFrame->frameId;
    test.ring_slot = slot.kf_slot;
    std::memcpy(test.poseWrtWorld, testFrame->poseWrtWorld, 24);
    std::memcpy(test.poseWrtOrigin, testFrame->poseWrtOrigin, 24);
    calculateImageHistogram(test);
    slot.isStray = true;   // :750 no depth map: findMatch never aligns against it (it needs one) until a keyframe replaces it
    slot.frameId = currentLoopFrame.frameId;
    slot.isValid = currentLoopFrame.isValid;
    std::memcpy(slot.image_histogram, currentLoopFrame.image_histogram, sizeof(slot.image_histogram));
  }
  ~globalOptimize() {
    try { join_all(); } catch (...) {}
  }
  // t_group.join_all() (:161); rethrows what the matching thread failed with
  void join_all() {
    if (t_group.joinable()) t_group.join();
    if (!thread_error.empty()) {
      const std::string e = thread_error;
      thread_error.clear();
      throw std::runtime_error(e);
    }
  }

  // GlobalOptimize.cpp:151-272 (FLAG_USE_LOOP_CLOSURE_TRIGGER off: every finished keyframe is tested)
  void pushToArray(frame* currentframe, depthMap* currentDepthMap) {
    join_all();   // :161 wait for the match thread before pushing another frame
    loopFrame& slot = loopFrameArray[currentArrayId];
    slot.kf_slot = currentArrayId;
    slot.isStray = false;
    // :185-186 deep copies of the keyframe and its depth map, into the ring context
    rt->check(ellc_copy_slot_across(ring.ctx, 1, slot.kf_slot, rt->ctx, 1, currentframe->kf_slot), "ellc_copy_slot_across");
    TestFrame test;
    test.frameId = currentframe->frameId;
    test.ring_slot = slot.kf_slot;
    std::memcpy(test.poseWrtWorld, currentframe->poseWrtWorld, 24);
    std::memcpy(test.poseWrtOrigin, currentframe->poseWrtOrigin, 24);
    calculateImageHistogram(test);
    slot.frameId = currentLoopFrame.frameId;
    slot.isValid = currentLoopFrame.isValid;
    std::memcpy(slot.image_histogram, currentLoopFrame.image_histogram, sizeof(slot.image_histogram));
    std::memcpy(slot.poseWrtWorld, currentframe->poseWrtWorld, 24);
    std::memcpy(slot.poseWrtOrigin, currentframe->poseWrtOrigin, 24);
    slot.rescaleFactor = currentframe->rescaleFactor;
    slot.seeds = currentDepthMap->calculate_no_of_Seeds();
    if (FLAG_DO_PARALLEL_SHORT_LOOP_CLOSURE) {   // :239-241
      t_group = std::thread([this, test]() {
        try { findMatchParallel(test); } catch (const std::exception& e) { thread_error = e.what(); }
      });
    } else {
      findMatchParallel(test);
    }
  }

 private:
  // what the matching thread keeps of the pushed keyframe (the reference hands it loopFrameArray[currentArrayId].this_frame, a
  // deep copy: the caller's frame object may be gone before the thread ends)
  struct TestFrame {
    int frameId, ring_slot;
    float poseWrtWorld[6], poseWrtOrigin[6];
  };
  std::thread t_group;
  std::string thread_error;
  bool fixed_grids_ = true;

  // :40-100 (the histogram of the copy in the ring context: the same image)
  void calculateImageHistogram(const TestFrame& f) {
    isloopClosureDetected = false;
    loopClosureArrayId = -1;
    currentLoopFrame.isValid = true;
    currentLoopFrame.frameId = f.frameId;
    std::memcpy(currentLoopFrame.poseWrtWorld, f.poseWrtWorld, 24);
    std::memcpy(currentLoopFrame.poseWrtOrigin, f.poseWrtOrigin, 24);
    ring.check(ellc_histogram(ring.ctx, 1, f.ring_slot, currentLoopFrame.image_histogram), "ellc_histogram");
  }
  // :436-452  third row of the rotation of exp(pose)
  static void calculateViewVec(const float* pose, float* view_vec) {
    float T[16];
    ellc_se3_exp(pose, T);
    view_vec[0] = T[8]; view_vec[1] = T[9]; view_vec[2] = T[10];
  }
  // :419-434
  void calculateRotationStats(const float* p1, const float* p2) {
    rms_error = (float)std::pow(std::pow(p1[0] - p2[0], 2) + std::pow(p1[1] - p2[1], 2) + std::pow(p1[2] - p2[2], 2), 0.5);
    float v1[3], v2[3];
    calculateViewVec(p1, v1);
    calculateViewVec(p2, v2);
    const float mag1 = (float)std::pow(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], 0.5);
    const float mag2 = (float)std::pow(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2], 0.5);
    relative_view_angle = std::acos((v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]) / (mag1 * mag2));
    relative_view_angle = (relative_view_angle * 180) / 3.14f;   // sic: 3.14
  }
  // :274-416 (strayFlag = false)
  bool findMatch(const TestFrame& currentframe) {
    const int RING = MAX_LOOP_ARRAY_LENGTH_SCALE_AVG;
    calculateImageHistogram(currentframe);
    int i;
    if (lastTestedLoopClosureArrayId == -1) i = currentArrayId - 1;
    else if (lastTestedLoopClosureArrayId != 0) i = lastTestedLoopClosureArrayId - 1;
    else i = RING - 1;
    if (i < 0) i = RING - 1;
    int conditionToTerminateLoop = 0;
    while (1) {
      lastTestedLoopClosureArrayId = i;
      if (match_window_end > match_window_beg) {
        if (!((i >= match_window_beg) && (i <= match_window_end))) conditionToTerminateLoop = 1;
      } else if (match_window_end < match_window_beg) {
        if (!((i >= match_window_beg) || (i <= match_window_end))) conditionToTerminateLoop = 1;
      } else if (loopFrameArray[i].isValid == false) conditionToTerminateLoop = 1;
      if (conditionToTerminateLoop == 1) { lastTestedLoopClosureArrayId = -1; break; }
      if (loopFrameArray[i].isValid == 0) { lastTestedLoopClosureArrayId = -1; return false; }
      // (a stray entry — an image findConnection pushed, no depth map — is never a candidate: the reference would hand its null
      // depth map to GetImagePoseEstimate; unreachable as shipped, FLAG_RESTORE_CONNECTION is off)
      if (!loopFrameArray[i].isStray && currentframe.frameId - loopFrameArray[i].frameId > 8) {   // MIN_MATCH_DIFFERENCE = KEYFRAME_PROPAGATE_INTERVAL
        matchValue = (float)ellc_kl_divergence(loopFrameArray[i].image_histogram, currentLoopFrame.image_histogram, 256);
        calculateRotationStats(loopFrameArray[i].poseWrtWorld, currentLoopFrame.poseWrtWorld);
        if (matchValue <= 0.1f) {                       // MATCH_THRESHOLD
          if (relative_view_angle <= 10.0f) {           // MAX_REL_VIEW_ANGLE
            isloopClosureDetected = true;
            loopClosureArrayId = i;
            break;
          }
        }
      }
      i--;
      if (i < 0) i = RING - 1;
    }
    return isloopClosureDetected;
  }
  // :454-646
  void findMatchParallel(const TestFrame& testFrame) {
    const int RING = MAX_LOOP_ARRAY_LENGTH_SCALE_AVG;
    struct Match { int arrayId; float matchValue, rms, angle; };
    std::vector<Match> matches;
    loopFrameArray[nextArrayId].frameId = testFrame.frameId;
    lastTestedLoopClosureArrayId = -1;
    firstTestedLoopClosureArrayId = -1;
    int num_matches = 0;
    do {
      const bool matchFound = findMatch(testFrame);
      if (num_matches > 0 && lastTestedLoopClosureArrayId == firstTestedLoopClosureArrayId) break;
      if (matchFound) {
        if (num_matches == 0) firstTestedLoopClosureArrayId = lastTestedLoopClosureArrayId;
        num_matches++;
        matches.push_back(Match{loopClosureArrayId, matchValue, rms_error, relative_view_angle});
      }
    } while (lastTestedLoopClosureArrayId != -1);
    if (!matches.empty()) {
      // one batched constant-weight alignment for all candidates (GetImagePoseEstimate(..., fromLoopClosure = true), :566)
      const int B = (int)matches.size();
      std::vector<int> kf(B), fr(B, 0);
      std::vector<float> init((size_t)B * 6), out((size_t)B * 6);
      ring.check(ellc_copy_slot(ring.ctx, 0, 0, 1, testFrame.ring_slot), "ellc_copy_slot");
      for (int b = 0; b < B; b++) {
        const loopFrame& m = loopFrameArray[matches[b].arrayId];
        kf[b] = m.kf_slot;
        ellc_concatenate_origin_pose(testFrame.poseWrtWorld, m.poseWrtWorld, &init[(size_t)b * 6]);   // ImageFunc.cpp:106
      }
      if (fixed_grids_) ring.check(ellc_ctx_set_grid_batch(ring.ctx, lc_grid_bucket(B)), "ellc_ctx_set_grid_batch");   // of the WHOLE batch, on every rank
      if (rt->comm && rt->world > 1) {
        // this rank's contiguous block of the candidates on its own GPU, then the one exchange of the batch: 8 floats per
        // alignment [pose6, weightedPose, iterations] from every rank, in global order on every rank
        int lo = 0, hi = B;
        ellc_shard_range(B, rt->world, rt->rank, &lo, &hi);
        const int n = hi - lo;
        std::vector<float> pose((size_t)std::max(n, 1) * 6), wgt((size_t)std::max(n, 1)), local((size_t)std::max(n, 1) * 8), table((size_t)B * 8);
        std::vector<int> iters((size_t)std::max(n, 1) * ring.cfg.levels);
        if (n > 0)
          ring.check(ellc_align(ring.ctx, n, kf.data() + lo, fr.data() + lo, init.data() + (size_t)lo * 6, ELLC_MODE_ICA, 0, pose.data(), iters.data(), wgt.data()),
                     "ellc_align");
        for (int b = 0; b < n; b++) {
          for (int k = 0; k < 6; k++) local[(size_t)b * 8 + k] = pose[(size_t)b * 6 + k];
          local[(size_t)b * 8 + 6] = wgt[b];
          int it = 0;
          for (int l = 0; l < ring.cfg.levels; l++) it += iters[(size_t)b * ring.cfg.levels + l];
          local[(size_t)b * 8 + 7] = (float)it;
        }
        if (ellc_gather_results(rt->comm, B, local.data(), n, table.data()) != ELLC_OK)
          throw std::runtime_error(std::string("ellc_gather_results failed: ") + ellc_comm_last_error(rt->comm));
        for (int b = 0; b < B; b++)
          for (int k = 0; k < 6; k++) out[(size_t)b * 6 + k] = table[(size_t)b * 8 + k];
      } else {
        ring.check(ellc_align(ring.ctx, B, kf.data(), fr.data(), init.data(), ELLC_MODE_ICA, 0, out.data(), nullptr, nullptr), "ellc_align");
      }
      for (int b = 0; b < B; b++) {
        const loopFrame& m = loopFrameArray[matches[b].arrayId];
        float poseWrtOrigin[6];
        ellc_concatenate_relative_pose(&out[(size_t)b * 6], m.poseWrtOrigin, poseWrtOrigin);   // ImageFunc.cpp:305
        const int seeds_num = (int)m.seeds;   // `int seeds_num` in the reference (:466)
        if (match_file.is_open()) {           // :580
          match_file << (testFrame.frameId + rt->BATCH_START_ID - 1) << " " << (m.frameId + rt->BATCH_START_ID - 1) << " " << poseWrtOrigin[0] << " "
                     << poseWrtOrigin[1] << " " << poseWrtOrigin[2] << " " << poseWrtOrigin[3] << " " << poseWrtOrigin[4] << " " << poseWrtOrigin[5]
                     << " " << m.rescaleFactor << " " << seeds_num << " " << matches[b].matchValue << " " << matches[b].rms << " "
                     << matches[b].angle << "\n";
        }
      }
      match_file.flush();
    }
    // :614-641
    currentArrayId++;
    nextArrayId++;
    if (currentArrayId == match_window_end + 2) { match_window_beg++; match_window_end++; }
    if (currentArrayId == 1 && match_window_end == RING - 1) { match_window_beg++; match_window_end = 0; }
    if (currentArrayId == RING) currentArrayId = 0;
    if (nextArrayId == RING) nextArrayId = 0;
    if (match_window_end == RING) match_window_end = 0;
    if (match_window_beg == RING) match_window_beg = 0;
  }
};

}  // namespace ellc
#endif
