/* Diagnostic extension of the ELLC C ABI (include/ellc_abi.h): measurement hooks (bench.py's roofline leg, tools/), device self-tests
 * (tests/) and test hooks of the resident schedule. NOT part of the product interface: libellc_hip.so does not export these symbols;
 * libellc_hip_diag.so — csrc/ built with -DELLC_DIAG_ABI, the same kernels and launch paths — exports them in addition to everything
 * ellc_abi.h declares. Nothing here replaces a member of the reference. */
#ifndef ELLC_ABI_DIAG_H
#define ELLC_ABI_DIAG_H
#include "ellc_abi.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement hooks (bench.py) ------------------------------------------------------------------- */
/* Launch the dominant kernel (FCA residual/Jacobian/accumulate at `level` over a batch) `reps` times on
 * the context stream between two HIP events; returns the average milliseconds per launch and the
 * algorithmic bytes one launch covers (4*N + 14*V summed over the batch). */
ellc_status ellc_profile_gn_kernel(ellc_ctx* ctx, int B, const int* kf_slots, const int* frame_slots, int level, int reps,
                                   float* avg_ms, double* algorithmic_bytes, long long* valid_pixels);
/* Time `reps` full ellc_align_enqueue passes with HIP events on the context stream (ms per pass). (A call that runs the
 * early-exit tracking schedule described above is timed without a remainder it might need.) */
ellc_status ellc_profile_align(ellc_ctx* ctx, int B, const int* kf_slots, const int* frame_slots, const float* init_pose,
                               int mode, int reps, float* avg_ms);

/* `reps` enqueues of one depth-map stage between two HIP events on the context stream (ms per call). stage 0:
 * regularizeDepthMap(false), 1: fillDepthHoles, 2: observeDepthRow against frame_slot / pose, 3: updateDepthImage, 4: createKeyFrame's
 * regularise(remove occlusions) + fill + regularise in one launch, 5: the tracked frame's fill + regularise + updateDepthImage in
 * one launch. */
ellc_status ellc_profile_depth_stage(ellc_ctx* ctx, int stage, int frame_slot, const float* pose_frame_wrt_kf, int reps, float* avg_ms);

/* Counter calibration: stream `bytes` of device memory once per launch with 4-byte-per-lane loads (the access
 * width of the compacted pixel arrays), `reps` launches, so FETCH_SIZE can be scaled against a known byte count
 * (MI355X_MICROARCH.md, HBM section). Returns average milliseconds per launch. */
ellc_status ellc_profile_calibrate_read(ellc_ctx* ctx, size_t bytes, int reps, float* avg_ms);
/* Streaming-read rate with 16-byte lanes over `bytes` of device memory: the practical ceiling behind the nominal HBM peak;
 * bench.py reports it beside the roofline (SURVEY.md section 8d asks for the measured figure). */
ellc_status ellc_profile_stream_read(ellc_ctx* ctx, size_t bytes, int reps, float* avg_ms);

/* Device self-test: q_pair[i] from the kernels' packed two-at-a-time IEEE division, q_ref[i] = a[i] / b[i] as the
 * compiler emits it; n even. The per-pixel code relies on the two being bit-identical (tests/test_gpu_gn.py). */
ellc_status ellc_selftest_div_pair(ellc_ctx* ctx, int n, const float* a, const float* b, float* q_pair, float* q_ref);

/* Device self-test of the solve's 6x6 inverse (cv::Mat::inv(DECOMP_LU) restated, PixelWisePyramid.cpp:451): n symmetric
 * matrices, each given by its 21 upper-triangular entries by rows (f64, rounded to f32 as the solve does); inv36 receives
 * the row-major f32 inverses (all zeros for a singular matrix). */
ellc_status ellc_selftest_lu(ellc_ctx* ctx, int n, const double* tri21, float* inv36);


/* ---- test hooks of the resident schedule (gn_fca_persist, the tracking call's one-launch form) ------------------------------ */
/* Blocks with index >= first_block of every later resident launch start `polls` sleeps (about a microsecond each) late, as if the
 * dispatcher had placed them late: what a block that writes no records at the coarse levels must survive (it is lapped by the
 * writers and re-joins through the state line). polls = 0 switches the delay off. polls = -r (1 .. 255): no delay; block 0 raises
 * the abort word at the top of round r of every later resident launch — a launch abandoned half-way, which the launch path has to
 * finish with the same bits. */
ellc_status ellc_debug_persist_delay(ellc_ctx* ctx, int first_block, int polls);
/* The call counter the records' tags are made of (24 bits are used): tests set it just below the wrap. */
ellc_status ellc_debug_set_persist_epoch(ellc_ctx* ctx, unsigned epoch);
/* Eager lists (default on in a context that tracks): the depth map's export builds the tracking call's compact lists of its keyframe
 * right behind itself, so that the next alignment against that keyframe starts without the compaction. 0 switches that off (every
 * alignment builds its lists itself, as up to r05): the results must not change by a bit (tests). */
ellc_status ellc_debug_set_eager_lists(ellc_ctx* ctx, int on);
/* A tracking call whose compact lists are already there (eager lists) has no staging launch: its resident launch builds the state
 * records from its arguments and takes the batch description and the seeds count along, and ellc_track_frame's saved weights are
 * added by further blocks of its selection launch (default on). 0: the staging kernel and gn_add_saved_weights_all run as up to r05 —
 * the results must not change by a bit (tests). */
ellc_status ellc_debug_set_fold_staging(ellc_ctx* ctx, int on);
/* Constant-weight path, tolerance mode: the per-(slot, level) H^-1 are kept while the keyframe's planes are unchanged and the
 * per-call compaction then builds the records only (default on). 0: every compaction recomputes them, as up to r05 — the results
 * must not change by a bit (tests). */
ellc_status ellc_debug_set_hinv_cache(ellc_ctx* ctx, int on);
/* Resident launches of this context so far, how many of them the host had to finish with launches (abandoned), and — device-wide,
 * since the library was loaded — how many blocks were lapped and re-joined through the state line. Any pointer may be NULL. */
ellc_status ellc_debug_persist_counters(ellc_ctx* ctx, long long* resident_launches, long long* abandoned_launches, long long* rejoined_blocks);

#ifdef __cplusplus
}
#endif
#endif /* ELLC_ABI_DIAG_H */
