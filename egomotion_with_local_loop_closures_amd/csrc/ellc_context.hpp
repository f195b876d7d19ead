// Host-side context behind the C ABI (include/ellc_abi.h).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <array>
#include <map>
#include <tuple>
#include "../../include/ellc_abi.h"
#ifdef ELLC_DIAG_ABI
#include "../../include/ellc_abi_diag.h"
#endif
#include "ellc_device.hpp"

namespace ellc {

struct DepthSoA {   // DepthHypothesis.h:14-40, live fields, structure of arrays
  float* invDepth = nullptr;
  float* invDepthSmoothed = nullptr;
  float* variance = nullptr;
  float* varianceSmoothed = nullptr;
  int* validity = nullptr;
  int* blacklisted = nullptr;
  uint8_t* isValid = nullptr;
};

}  // namespace ellc

struct ellc_ctx {
  ellc_config cfg;
  hipStream_t stream = nullptr;
  std::string err;
  int L = 0;
  ellc::LevelGeom geom_h[ELLC_MAX_LEVELS];
  ellc::LevelGeom* geom_d = nullptr;
  std::vector<void*> allocs;                 // the arena chunks (freed on destroy)
  char* arena_base = nullptr;                // current chunk: device buffers are carved out of a few large allocations
  size_t arena_size = 0, arena_used = 0;
  bool fast = false;                         // cfg.arith == ELLC_ARITH_FAST
  std::vector<void*> host_allocs;            // hipHostMalloc'ed
  std::vector<ellc::KfLevelDev> kf_tab_h;    // [L][max_kf]
  std::vector<ellc::FrLevelDev> fr_tab_h;    // [L][max_fr]
  ellc::KfLevelDev* kf_tab_d = nullptr;
  ellc::FrLevelDev* fr_tab_d = nullptr;
  std::vector<char> kf_has_image, kf_has_depth, fr_has_image;
  // dense hint: the slot's level-0 depth plane was uploaded with at least nine tenths of its pixels valid (ellc_keyframe_set_depth
  // counts them). A tolerance-mode FCA batch whose keyframes all carry the hint is aligned without compact lists (gn_fca_dense:
  // thread <-> pixel, the planes read directly). A hint, not a promise: pixels without depth are skipped where they occur.
  std::vector<char> kf_dense;
  bool cur_dense = false;   // the schedule being enqueued is the list-free one
  bool dense_maps_off = false;   // ellc_ctx_set_dense_maps(1): dense maps take the list path too (bit-comparison runs)
#ifdef ELLC_NO_DENSE_QUADS
  bool dense_quads = false;   // (A/B builds: the r05 kernel, one pixel per thread)
#else
  bool dense_quads = true;    // list-free launches take four adjacent pixels per thread (gn_fca_dense4) at levels whose width allows
#endif
  // cfg.cache_records: which record set (PrepArgs::need) the compact lists of a keyframe slot hold, 0 = none / stale. The lists
  // are a pure function of the slot's image, depth pyramid and weight planes: every entry point that writes one of those
  // clears the tag (ellc::invalidate_records); a batch rebuilds only the slots whose tag differs from what it needs.
  std::vector<int> kf_rec_tag;
  std::vector<char> kf_rec_eager;   // the slot's lists were built behind the depth map's export and are valid whatever cfg.cache_records says (enqueue_eager_lists)
  bool eager_lists = true;
  bool fold_staging = true;   // a tracking call whose lists are there leaves its staging to the resident launch (PersistStage)
  bool stage_folded = false;  // ... decided by enqueue_stage_in for the launch being enqueued
  // ICA, tolerance mode: H^-1 per (slot, level) — the inverse of sum W J^T J over the keyframe's valid pixels (PixelWisePyramid.cpp:938-939)
  // — is a function of the keyframe's planes alone. kf_hinv_ok[slot]: the inverses the last compaction of the slot left are still
  // current (every writer of the planes clears it through invalidate_records): the next compaction builds the records only (r06)
  std::vector<char> kf_hinv_ok;
  bool hinv_cache = true;
  int cur_need = 0;   // record set of the launch being enqueued when it differs from need_of() (16: records without the H sums)
  bool cache_records = false;
  std::vector<std::array<int, ELLC_MAX_LEVELS>> kf_num_weights;
  std::vector<float*> kf_maxgrad, fr_maxgrad;
  std::vector<int*> kf_maxgrad_count, fr_maxgrad_count;
  std::vector<char> kf_maxgrad_valid, fr_maxgrad_valid;
  // alignment work buffers
  int *kf_slot_d = nullptr, *fr_slot_d = nullptr, *uniq_slot_d = nullptr;
  int *kf_slot_h = nullptr, *fr_slot_h = nullptr, *uniq_slot_h = nullptr;   // pinned
  float *init_pose_d = nullptr, *init_pose_h = nullptr;
  ellc::AlignState *state_d = nullptr, *state_h = nullptr;
  ellc::AlignResult* result_h = nullptr;            // pinned; written by the last kernel of a schedule through result_dev_alias
  ellc::AlignResult* result_dev_alias = nullptr;
  const int* stage_dev_alias = nullptr;             // device-side address of the pinned staging record (kf_slot_h ...)
  // Batches in flight (ellc_align_enqueue several times before ellc_align_fetch). The unit that runs on the device is a GROUP:
  // up to cfg.coalesce full batches (B == max_batch) staged side by side in one set of buffers and launched as ONE sequence
  // over all their alignments — a launch over 96 alignments costs little more than one over 32 (the per-launch costs of the
  // dependent chain: launch gap, cold loads, solve, reduction, and the latency-bound coarse levels that do not fill the
  // device, are paid once), r02: 4.7 -> 6.5 M iterations/s at three groups of three in flight. With cfg.coalesce = 1 (default)
  // a group is one batch and everything is as before. Every group has its own pinned staging / result records and device work
  // buffers (staged slots, AlignState, block partials), sized for coalesce x max_batch alignments; launched groups take one of
  // three streams (the main stream first: a caller with one batch at a time never leaves it; a process holds few hardware
  // queues, and streams that end up on the same one do not overlap), so up to three groups run CONCURRENTLY: the latency-bound
  // coarse iterations of one overlap the throughput-bound fine iterations of another. Groups that share a keyframe slot are
  // ordered one after the other (the compaction, H^-1 and the saved weights live in the keyframe slot). There is one set more than
  // batches may be in flight (max_inflight): it takes the batches that arrive while the oldest group is being fetched. The members above point at the set / slice of the batch being
  // staged or the group being launched. All other entry points work on `stream`, launch a group that is still open and make
  // the stream wait for the groups in flight, so a caller sees one in-order queue per context as before.
  static constexpr int MAX_COALESCE = 4;
#ifndef ELLC_STREAMS
#define ELLC_STREAMS 3
#endif
  static constexpr int STREAMS = ELLC_STREAMS;
  static constexpr int SETS = (STREAMS + 1) * MAX_COALESCE + 1;   // of which max_inflight + 1 are used (n_sets): every batch in flight may be a group of its own
  struct BatchSet {
    int* stage_h = nullptr;                         // 9 * cap ints: kf slots, frame slots, unique slots, initial poses (cap = coalesce * max_batch)
    const int* stage_dev_alias = nullptr;
    ellc::AlignResult* result_h = nullptr;
    ellc::AlignResult* result_dev_alias = nullptr;
    hipEvent_t done = nullptr;
    int* stage_d = nullptr;                         // device copy of the staging record
    ellc::AlignState* state_d = nullptr;            // two launch-parity buffers
    float* partials_d = nullptr;
    unsigned* persist_bar_d = nullptr;              // gn_fca_persist's abort words (two alignments; zero between calls)
    // the group staged / in flight in this set
    int fill = 0;                                   // batches staged side by side: batch j = alignments [j * max_batch, ...)
    int fetched = 0;                                // of which fetched (the set is free again when fetched == fill)
    bool launched = false;
    bool coalescable = false;                       // further full batches of the same mode may join until it is launched
    int slice_B[MAX_COALESCE] = {0, 0, 0, 0};       // size of each staged batch
    int stream_idx = 0;                             // the batch stream it was launched on (0: the context's main stream)
    std::vector<int> kf_slots;                      // unique keyframe slots of the group (all of them: the slots it reads)
    std::vector<int> built_slots;                   // of which it rebuilds the compact lists (or accumulates saved weights into)
    int B = 0;                                      // alignments the launch covers
    bool joined = true;                             // the main stream already waits for `done`
    int mode = 0, save_weights = 0;
    bool adaptive = false;                          // the group (one batch) runs the state-driven schedule (gn_fca_adaptive)
    bool resident = false;                          //   as one resident launch (gn_fca_persist)
    int adaptive_first = 0;                         //   whose first graph holds this many launches
    bool pollable = false;                          // its finish kernel ordered its result records for a polling host (FusedArgs::host_polls)
    bool resolved = true;                           // `done` has been waited for and the continuation, if one was needed, has run
  } batch_set[SETS];
  hipStream_t batch_stream[STREAMS] = {};   // [0] = stream; the others are created when first needed
  int stream_waited_mark[STREAMS] = {};      // the main-stream mark each batch stream has been ordered after
  int coalesce = 1;                                 // cfg.coalesce clamped to 1..MAX_COALESCE
  int max_inflight = 3;                             // batches in flight: 3 with coalesce = 1; 4 x coalesce otherwise (three groups
                                                    // running and a fourth queued behind the oldest, so that the device never waits
                                                    // for the host to gather the next group)
  int n_sets = 4;                                   // sets in use: max_inflight + 1
  int group_cap = 0;                                // alignments per set: coalesce * max_batch
  int open_set = -1;                                // the set whose group is staged but not launched yet (-1: none)
  hipEvent_t ev_main = nullptr;                     // marks the main stream behind the last non-batch call
  hipEvent_t ev_xfer = nullptr;                     // ellc_copy_slot_across: orders this context's stream against another context's
  bool main_dirty = false;                          // a non-batch entry point ran since ev_main was recorded
  int main_mark = 0;
  int inflight[SETS * MAX_COALESCE] = {0};         // FIFO of the batches in flight: set * MAX_COALESCE + slice
  int n_inflight = 0;
  int cur_set = 0;                                  // the batch set the per-batch pointers below refer to (select_batch_set)
  float* partials_d = nullptr;
  unsigned* persist_bar_d = nullptr;
  unsigned persist_spin_limit = 1u << 15;           // polls of a missing record before gn_fca_persist gives a launch up (0: at once — test hook)
  unsigned prep_tag = 0;                            // tag of the last compaction without a count launch (PrepArgs::lb_tag)
  unsigned persist_epoch = 0;                       // calls of gn_fca_persist so far (tags of its partial records)
  int persist_delay_from = 0, persist_delay_polls = 0;   // test hook (ellc_debug_persist_delay): blocks from this index on start late
  int persist_backoff = 0;                          // calls that still run as launches after a resident launch had to be abandoned
  long long persist_launches = 0, persist_abandoned = 0;   // resident launches so far / those the host had to finish with launches
  int persist_capacity = 0;                         // blocks of gn_fca_persist the device holds at once (occupancy x CUs)
  bool cur_resident = false;                        // the schedule being enqueued is the resident form
  bool use_persist = true;                          // the state-driven schedule as one resident launch (gn_fca_persist)
  float* planes_d = nullptr;
  float *scratch_a = nullptr, *scratch_b = nullptr;   // W*H f32 each
  int tile_begin[ELLC_MAX_LEVELS + 1];
  int cap[ELLC_MAX_LEVELS];                            // compact capacity per level (= n)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // captured launch sequences of ellc_align, keyed by (B, unique keyframes, mode, flags: save_weights | persist | run | continuation, batch set)
  std::map<std::tuple<int, int, int, int, int>, hipGraphExec_t> graphs;
  long long counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ELLC_CTR_* (ellc_abi.h): ellc_ctx_counters
  int poll_timeout_us = 2000;   // resolve_batch polls this long before it falls back to the event
  bool poll_results = true;     // resolve_batch polls the pinned result records of small single-stream batches; ELLC_NO_POLL=1 (diag)
  bool use_graph = true;
  bool direct_launch = false;   // the launch sequence being enqueued is not captured: its staging record goes through kernel arguments
  int direct_nu = 0;            //   unique keyframe slots whose lists it (re)builds
  int direct_max_batch = 0;     // level-bound launch sequences over at most this many alignments are launched kernel by kernel too (measured for B = 1: 0.225 ms either way)
  bool graph_adaptive = false;  // the state-driven (tracking) schedule as a captured graph too; ELLC_GRAPH_ADAPTIVE=1 (diag)
  bool age_balance = true;      // age-balanced split of full-round grids (FusedArgs::age_rounds); ELLC_NO_AGE_BALANCE=1 disables
  double age_weight[5][4] = {{1, 1, 1, 1}, {1, 1, 1, 1}, {1.15, 0.85, 1, 1}, {1.2, 1.0, 0.8, 1}, {1.35, 1.15, 0.9, 0.6}};   // [rounds][round], ELLC_AGE_W (r01 sweep at 640x480, batch 32)
  double age_min_px_per_thread = 5.0;   // ELLC_AGE_MIN_PX
  bool use_adaptive = true;             // early-exit FCA schedules are state-driven (gn_fca_adaptive); ELLC_NO_ADAPTIVE=1 (diag) turns it off
  int adaptive_max_batch = 2;           //   for batches of at most this many alignments; ELLC_ADAPTIVE_MAX_BATCH (diag)
  int adaptive_hint = 0;                //   launches of the next first graph, from what the previous call needed (0: none yet)
  int cur_adaptive_first = 0;           //   launches of the first graph of the schedule being enqueued / continued
  int adaptive_first_override = 0;      //   launches of the first graph; ELLC_ADAPTIVE_FIRST (diag)
  bool pipe = true;             // software-pipelined record loads in the fused FCA kernel (ELLC_PIPE=0 disables; r01: -10 % per launch at
                                // 1280x960 dense where the records stream from HBM, neutral at 640x480 semi-dense)
  bool use_fused = true;        // the production schedules (ELLC_NO_FUSE=1, diagnostic builds: one accumulate + one solve launch per iteration)
  int nblk_override[ELLC_MAX_LEVELS] = {0};
  int resident_blocks = 1280;   // 256-thread blocks of the accumulate kernel resident on the device at once
  // image uploads: ring of pinned staging buffers, so an upload only enqueues (ellc_hip.hip: upload_pyramid)
  static constexpr int UPLOAD_RING = 4;
  uint8_t* upload_stage[UPLOAD_RING] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t upload_done[UPLOAD_RING] = {nullptr, nullptr, nullptr, nullptr};
  int upload_cursor = 0;
  // frame uploads run on a stream of their own (created by the first ellc_frame_upload): per frame slot the event of its last
  // main-stream reader (mark_frame_use) and of its last upload
  hipStream_t upload_stream = nullptr;
  std::vector<hipEvent_t> fr_use_ev, fr_ready_ev;
  // frame ingest (ellc_ingest_configure): fixed-point undistortion map of the 2x2 source pixels of every output pixel
  void* ingest_map = nullptr;
  uint8_t* ingest_bgr = nullptr;    // staging for one full-size BGR frame
  int ingest_w = 0, ingest_h = 0;
  // depth map (one per context)
  ellc::DepthSoA dm_cur, dm_oth;
  int dm_kf_slot = -1;
  bool dm_ready = false;
  float dm_depth_scale = 1.0f, dm_global_scale = 1.0f;
  float *dm_deptharr0 = nullptr, *dm_vararr0 = nullptr;     // level-0 arrays in the reference's array convention
  // propagate scratch
  int *pr_tgt = nullptr, *pr_cnt = nullptr, *pr_slots = nullptr, *pr_val = nullptr, *pr_remaining = nullptr;   // pr_remaining: a counter word (ellc_depth_seeds)
  float *pr_id = nullptr, *pr_var = nullptr;
  double* red_scratch = nullptr;                            // reductions (rescale factor)
  double* sum_parts = nullptr;                              // per-tile (sum, count) of dm_reg_fill_reg for the rescale in the export's launch
  // ellc_track_frame: the observation's matrices and the gate, built on the device behind the alignment; a host-visible record
  void* track_mats_d = nullptr;
  int* track_gate_d = nullptr;
  const void* track_count_valid = nullptr;   // ellc_track_frame: the validity plane whose count the staging launch takes along
  int track_count_n = 0;                     //   and its size (0: nothing pending)
  bool cur_pollable = false;    // the launch sequence being enqueued ends in a finish kernel the host may poll
  bool done_deferred = false;   // launch_group left the group's `done` event to its caller (ellc_track_frame records it behind the depth stages)
  bool track_call = false;   // the alignment being enqueued belongs to ellc_track_frame (set_track_fields)
  int* seed_acc = nullptr;   // dm_count_valid_block: sum and arrival ticket (zero between calls)
  int obs_parity = 0;              // which of the two counter sets the next observation uses
  float2* obs_list_ep = nullptr;   // the epipolar direction of every list entry
  int *obs_list = nullptr, *obs_ctr = nullptr;   // work list of dm_observe_select / dm_observe_walk and its counters (zero between calls)
  int* track_h = nullptr;    // host-visible: [0] the valid hypotheses before the observation, [1] the number of the count that wrote it
  bool track_ride_weights = false;   // the enqueued tracking call's saved weights wait for its selection launch (launch_observe)
  bool ride_saved_weights = true;    //   (ellc_debug_set_fold_staging(0) keeps the launch of their own as well)
  int track_counts = 0;      // counts launched so far (ellc_track_frame waits for [1] to say this one)
  int* track_dev_alias = nullptr;
  float Kinv[9], Kmat[9];
};

namespace ellc {
ellc_status fail(ellc_ctx* c, ellc_status s, const std::string& msg);
ellc_status enqueue_eager_lists(ellc_ctx* c, int slot);   // the tracking call's lists of a keyframe slot, built behind the export of its planes
void invalidate_records(ellc_ctx* c, int slot);   // cfg.cache_records: the slot's compact lists no longer match its planes
#define ELLC_HIP(ctx, expr)                                                                               \
  do {                                                                                                    \
    hipError_t e__ = (expr);                                                                              \
    if (e__ != hipSuccess) return ellc::fail(ctx, ELLC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

// Every entry point makes the context's device current for the calling thread first: HIP's current device is per
// thread, and the reference calls GetImagePoseEstimate from a second (loop-closure) thread (GlobalOptimize.cpp:241).
ellc_status enter(ellc_ctx* c, bool join);
#define ELLC_ENTER_IMPL(ctx, join)                      \
  do {                                                  \
    if (ctx) {                                          \
      const ellc_status s__ = ellc::enter(ctx, join);   \
      if (s__ != ELLC_OK) return s__;                   \
    }                                                   \
  } while (0)
#define ELLC_ENTER(ctx) ELLC_ENTER_IMPL(ctx, true)          // runs on the main stream, behind the batches in flight
#define ELLC_ENTER_BATCH(ctx) ELLC_ENTER_IMPL(ctx, false)   // ellc_align_enqueue / ellc_align_fetch

int choose_nblk(const ellc_ctx* c, int level, int B);
ellc_status run_prep(ellc_ctx* c, int n_unique, int need);
ellc_status build_depth_pyramid(ellc_ctx* c, int slot);
ellc_status build_depth_pyramid_from(ellc_ctx* c, int slot, int first_level);
ellc_status build_maxgrad(ellc_ctx* c, bool is_kf, int slot);
ellc_status build_image_pyramid(ellc_ctx* c, uint8_t* const* img, hipStream_t st);
ellc_status mark_frame_use(ellc_ctx* c, int slot);
}  // namespace ellc
