// libellc_hip.so — C ABI implementation (context, uploads, alignment scheduling). gfx950 only.
// Reference interfaces replaced by each entry point are cited in include/ellc_abi.h.
#include "ellc_context.hpp"
#include "ellc_kernels_image.hpp"
#include "ellc_kernels_gn.hpp"
#include "ellc_kernels_prep.hpp"
#include <cstring>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <chrono>
#include <thread>
#include <atomic>
#include <mutex>

using namespace ellc;

static ellc_status resolve_batch(ellc_ctx* c, int set);   // waits for a group in flight (and runs its continuation), defined with ellc_align_fetch
static ellc_status launch_group(ellc_ctx* c, int set);    // launches the group staged in a set

namespace ellc {

ellc_status fail(ellc_ctx* c, ellc_status s, const std::string& msg) {
  if (c) c->err = msg;
  return s;
}

// the compact lists of a keyframe slot no longer match its planes (cfg.cache_records)
void invalidate_records(ellc_ctx* c, int slot) {
  if (slot >= 0 && slot < (int)c->kf_rec_tag.size()) { c->kf_rec_tag[slot] = 0; c->kf_rec_eager[slot] = 0; c->kf_hinv_ok[slot] = 0; }
}

// blocking copy on the context's own stream: the legacy default stream would synchronise with every other stream of the
// process (another context's batches in flight) and take a hardware queue of its own
static hipError_t copy_blocking(ellc_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  return e;
}

ellc_status enter(ellc_ctx* c, bool join) {
  int dev = -1;
  if (!(hipGetDevice(&dev) == hipSuccess && dev == c->cfg.device) && hipSetDevice(c->cfg.device) != hipSuccess)
    return fail(c, ELLC_ERR_HIP, "cannot make the context's device current");
  if (join) {   // everything but the batch entry points runs on the main stream, after the batches in flight
    if (c->open_set >= 0) {   // a group still waiting for batches to join: this call comes after them in the caller's order
      const ellc_status s = ::launch_group(c, c->open_set);
      if (s != ELLC_OK) return s;
    }
    for (int p = 0; p < ellc_ctx::SETS; p++) {
      ellc_ctx::BatchSet& bs = c->batch_set[p];
      if (!bs.launched) continue;
      if (bs.adaptive && !bs.resolved) {
        // a state-driven batch may need a continuation that only the host can start, and this call may change what it
        // reads (an upload into one of its slots, a depth stage): the host finishes the batch first
        const ellc_status s = ::resolve_batch(c, p);
        if (s != ELLC_OK) return s;
      }
      if (bs.joined) continue;   // it runs on the main stream itself
      ELLC_HIP(c, hipStreamWaitEvent(c->stream, bs.done, 0));
      bs.joined = true;
    }
    c->main_dirty = true;   // the next group on another stream has to be ordered after what this call enqueues
  }
  return ELLC_OK;
}

// Device memory of a context comes from a few large chunks (first 64 MiB, then doubling up to 4 GiB), each allocated and
// zeroed once; buffers are carved out at 256-byte granularity. A 640x480 context with 96 keyframe slots is thousands of
// buffers: one hipMalloc + fill launch each made context creation take seconds.
static ellc_status dev_alloc_bytes(ellc_ctx* c, void** p, size_t bytes) {
  bytes = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
  if (c->arena_used + bytes > c->arena_size) {
    const size_t next = std::min<size_t>(std::max<size_t>(c->arena_size * 2, (size_t)64 << 20), (size_t)4 << 30);
    const size_t chunk = std::max(bytes, next);
    void* q = nullptr;
    ELLC_HIP(c, hipMalloc(&q, chunk));
    c->allocs.push_back(q);
    ELLC_HIP(c, hipMemsetAsync(q, 0, chunk, c->stream));
    c->arena_base = (char*)q;
    c->arena_size = chunk;
    c->arena_used = 0;
  }
  *p = c->arena_base + c->arena_used;
  c->arena_used += bytes;
  return ELLC_OK;
}
template <class T>
static ellc_status dev_alloc(ellc_ctx* c, T** p, size_t count) {
  void* q = nullptr;
  const ellc_status s = dev_alloc_bytes(c, &q, count * sizeof(T));
  *p = (T*)q;
  return s;
}
template <class T>
static ellc_status host_alloc(ellc_ctx* c, T** p, size_t count) {
  void* q = nullptr;
  ELLC_HIP(c, hipHostMalloc(&q, std::max<size_t>(count * sizeof(T), 16), hipHostMallocDefault));
  c->host_allocs.push_back(q);
  *p = (T*)q;
  return ELLC_OK;
}

// Blocks per alignment for the accumulate kernels. Enough blocks to give every thread about one pixel at the
// coarse levels (those launches are latency-bound); at the fine levels exactly the number of blocks the device
// holds at once, so every CU gets the same share and each block pays the 27-value reduction once.
// does a list-free launch at this level take four adjacent pixels per thread (gn_fca_dense4)? A row must be a whole number of quads
static bool dense_quads_at(const ellc_ctx* c, int level) {
  const LevelGeom& lg = c->geom_h[level];
  return c->fast && c->dense_quads && lg.cols % 4 == 0 && lg.sw % 4 == 0 && lg.cols >= 16 && lg.rows >= 8;   // (tolerance mode only)
}
int choose_nblk(const ellc_ctx* c, int level, int B) {
#ifdef ELLC_DIAG
  if (c->nblk_override[level] > 0) return std::min(ELLC_NBLK_MAX, c->nblk_override[level]);   // tuning knob (ELLC_NBLK=l0,l1,..)
#endif
  const int n = c->geom_h[level].n;
  const int by_px = std::max(1, n / 640);   // semi-dense maps are ~30 % valid: about one valid pixel per thread
  B = std::max(1, B);
  // one round of resident blocks (measured best for a batch that has the device to itself: r01 sweep); half a round when
  // the caller keeps several batches in flight (cfg.concurrent_batches), so that the fine-level launches of two batches can share the device (r01 sweep with three
  // in flight at B=32: 32/32 blocks 0.375 ms per batch, 16/16 0.338, 12/12 0.339, 8/8 0.348, 64/32 0.392)
  // (r03: a launch that covers a whole group of batches keeps the full round where that still leaves a thread plenty of pixels —
  // level 0 of 640x480 over 128 alignments: 8 blocks per alignment instead of 4, the launch 49 instead of 61 us alone, the
  // pipeline's rate unchanged (tools/dbg/nblk_bench.sh); level 1 is slower with the full round)
  // (list-free launches with four pixels per thread hold three blocks per CU, not four: gn_fca_dense4)
  const int blocks_per_cu = c->use_fused ? 4 : 5;
  const bool quads = c->cur_dense && dense_quads_at(c, level);
  const int resident = quads ? c->resident_blocks / blocks_per_cu * ELLC_QUAD_BLOCKS_PER_CU : c->resident_blocks;
  const int per_full = std::max(1, resident / B);
  const bool share = c->cfg.concurrent_batches > 1 && (double)n / (256.0 * per_full) < 100.0;
  const int per = std::max(1, per_full / (share ? 2 : 1));
  int nblk = std::min(ELLC_NBLK_MAX, std::min(by_px, per));
  // small levels: one block per CU runs the (serial) solve prologue and the short pixel pass fastest, as long as a
  // thread does not get more than ~4 pixels (r01 sweep: level 2 at B=32, 8 blocks beat 30; level 1 keeps 32)
  const int cus = std::max(1, c->resident_blocks / blocks_per_cu);
  if (nblk * B > cus) {
    const int one_per_cu = std::max(1, cus / B);
    if (0.3 * n / (256.0 * one_per_cu) <= 4.0) nblk = std::min(nblk, one_per_cu);
  }
  // One or two alignments on their own (the tracking call): an iteration is a latency chain, and every block reads every block's
  // record — with 256 blocks a level-0 round waits 4.3 us for 8 MB of records, with 128 it waits 2.3 and its pixel pass is one
  // step longer (tools/dbg/persist_trace.py, tools/dbg/sweep_persist_nblk.sh: one early-exit alignment 0.1132 -> 0.1100 ms, tracked
  // frame 0.1850 -> 0.1826; 64 blocks: 0.1108 / 0.187). Resident launch and launch-per-iteration path alike (the same bits). Only
  // where the state-driven schedule runs (a context with the early exit): a fixed schedule of one alignment is throughput-bound at
  // level 0 and lost 4 % with the cap (C1 with 32 iterations: 0.211 -> 0.219 ms).
  if (B <= 2 && c->use_fused && c->use_adaptive && c->cfg.early_exit && c->cfg.grid_batch == 0) nblk = std::min(nblk, 128);
  return nblk;
}

static dim3 grid2d(int w, int h, dim3 blk) { return dim3((w + blk.x - 1) / blk.x, (h + blk.y - 1) / blk.y); }

// the u8 pyramid below level 0 of `img[]`: one launch per three levels (pyr_down_chain_u8)
ellc_status build_image_pyramid(ellc_ctx* c, uint8_t* const* img, hipStream_t st) {
  const LevelGeom* g = c->geom_h;
  for (int l = 0; l + 1 < c->L; l += 3) {
    PyrChainArgs a;
    a.steps = std::min(3, c->L - 1 - l);
    a.src = img[l];
    for (int k = 0; k < 4; k++) {
      const int q = std::min(l + k, c->L - 1);
      a.w[k] = g[q].sw;
      a.h[k] = g[q].sh;
    }
    for (int k = 0; k < 3; k++) a.dst[k] = img[std::min(l + 1 + k, c->L - 1)];
    const int dw = g[l + a.steps].sw, dh = g[l + a.steps].sh;
    hipLaunchKernelGGL(pyr_down_chain_u8, dim3((dw + ELLC_PT - 1) / ELLC_PT, (dh + ELLC_PT - 1) / ELLC_PT), dim3(256), 0, st, a);
  }
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

// Upload + pyramid without stalling the caller: the image is copied into one of a ring of pinned staging buffers (so the
// caller's buffer is free when this returns, as with the blocking copy it replaces), and the host-to-device copy and the
// pyramid launch are only enqueued. A staging buffer is reused once the copy that read it has completed (event).
// frame_slot >= 0: the upload of a current-frame slot runs on a stream of its own, so that the copy and the pyramid overlap what
// is already enqueued on the main stream (in a tracking loop: the previous frame's depth stages, which read the OTHER frame
// slot). Ordered on the device: behind the readers of this slot that are already enqueued (mark_frame_use, batches in flight),
// and everything enqueued later on the main stream waits for it.
static ellc_status upload_pyramid(ellc_ctx* c, uint8_t* const* img, const uint8_t* host, int frame_slot = -1) {
  const LevelGeom* g = c->geom_h;
  hipStream_t st = c->stream;
  if (frame_slot >= 0) {
    if (!c->upload_stream) {
      ELLC_HIP(c, hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking));
      c->fr_use_ev.assign(c->cfg.max_frames, nullptr);
      c->fr_ready_ev.assign(c->cfg.max_frames, nullptr);
      // readers enqueued before the stream existed were not marked: this once, behind everything on the main stream
      hipEvent_t all = nullptr;
      ELLC_HIP(c, hipEventCreateWithFlags(&all, hipEventDisableTiming));
      hipError_t e = hipEventRecord(all, c->stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(c->upload_stream, all, 0);
      (void)hipEventDestroy(all);   // (released once the wait has completed)
      ELLC_HIP(c, e);
    }
    st = c->upload_stream;
    if (c->fr_use_ev[frame_slot]) ELLC_HIP(c, hipStreamWaitEvent(st, c->fr_use_ev[frame_slot], 0));   // its last main-stream reader
    for (int p = 0; p < ellc_ctx::SETS; p++)   // batches in flight may read it (ELLC_ENTER has launched an open group)
      if (c->batch_set[p].launched) ELLC_HIP(c, hipStreamWaitEvent(st, c->batch_set[p].done, 0));
  }
  const size_t bytes = (size_t)g[0].sw * g[0].sh;
  const int k = c->upload_cursor;
  c->upload_cursor = (c->upload_cursor + 1) % ellc_ctx::UPLOAD_RING;
  if (!c->upload_stage[k]) {
    ELLC_HIP(c, hipHostMalloc((void**)&c->upload_stage[k], bytes, hipHostMallocDefault));
    ELLC_HIP(c, hipEventCreateWithFlags(&c->upload_done[k], hipEventDisableTiming));
  } else {
    ELLC_HIP(c, hipEventSynchronize(c->upload_done[k]));
  }
  std::memcpy(c->upload_stage[k], host, bytes);
  {   // the staging buffer is read by a kernel (ingest_copy_u8): no copy-engine hand-over in front of the pyramid launch
    void* stage_dev = nullptr;
    ELLC_HIP(c, hipHostGetDevicePointer(&stage_dev, c->upload_stage[k], 0));
    const int blocks = (int)std::min<size_t>(1024, ((bytes >> 4) + 255) / 256 + 1);
    hipLaunchKernelGGL(ingest_copy_u8, dim3(blocks), dim3(256), 0, st, img[0], (const uint8_t*)stage_dev, bytes);
  }
  const ellc_status s = build_image_pyramid(c, img, st);
  // (behind the pyramid launch, not between the two kernels: an event record there held the pyramid back ~7 us; the staging buffer is
  // one of a ring of four and is free by the time its turn comes again either way)
  ELLC_HIP(c, hipEventRecord(c->upload_done[k], st));
  if (s != ELLC_OK || frame_slot < 0) return s;
  if (!c->fr_ready_ev[frame_slot]) ELLC_HIP(c, hipEventCreateWithFlags(&c->fr_ready_ev[frame_slot], hipEventDisableTiming));
  ELLC_HIP(c, hipEventRecord(c->fr_ready_ev[frame_slot], st));
  ELLC_HIP(c, hipStreamWaitEvent(c->stream, c->fr_ready_ev[frame_slot], 0));   // whatever is enqueued from here on sees the new pyramid
  return ELLC_OK;
}

// a main-stream reader of frame slot `slot` has just been enqueued: a later upload into the slot waits for it (upload_pyramid)
ellc_status mark_frame_use(ellc_ctx* c, int slot) {
  if (!c->upload_stream) return ELLC_OK;   // no upload has left the main stream yet: plain stream order holds
  if (!c->fr_use_ev[slot]) ELLC_HIP(c, hipEventCreateWithFlags(&c->fr_use_ev[slot], hipEventDisableTiming));
  ELLC_HIP(c, hipEventRecord(c->fr_use_ev[slot], c->stream));
  return ELLC_OK;
}

ellc_status build_maxgrad(ellc_ctx* c, bool is_kf, int slot) {
  const LevelGeom& g = c->geom_h[0];
  const uint8_t* img = is_kf ? c->kf_tab_h[slot].img : c->fr_tab_h[slot].img;
  float* out = is_kf ? c->kf_maxgrad[slot] : c->fr_maxgrad[slot];
  int* cnt = is_kf ? c->kf_maxgrad_count[slot] : c->fr_maxgrad_count[slot];
  ELLC_HIP(c, hipMemsetAsync(cnt, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(maxgrad_fused, dim3((g.cols + 31) / 32, (g.rows + 7) / 8), dim3(256), 0, c->stream, img, g.sw, g.cols, g.rows, out, cnt);
  ELLC_HIP(c, hipGetLastError());
  (is_kf ? c->kf_maxgrad_valid : c->fr_maxgrad_valid)[slot] = 1;
  return ELLC_OK;
}

ellc_status build_depth_pyramid(ellc_ctx* c, int slot) { return build_depth_pyramid_from(c, slot, 1); }

ellc_status build_depth_pyramid_from(ellc_ctx* c, int slot, int first_level) {
  for (int l = std::max(1, first_level); l < c->L; l++) {
    const KfLevelDev& s = c->kf_tab_h[(l - 1) * c->cfg.max_keyframes + slot];
    const KfLevelDev& d = c->kf_tab_h[l * c->cfg.max_keyframes + slot];
    const int w = c->cfg.width >> l, h = c->cfg.height >> l;
    dim3 blk(32, 8);
    hipLaunchKernelGGL(depth_pyr_level, grid2d(w, h, blk), blk, 0, c->stream, s.depth, s.var, d.depth, d.var, w, h);
  }
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

// compaction for pyramid levels lvl_lo .. lvl_hi of the listed keyframes, on stream `st`
ellc_status run_prep_levels(ellc_ctx* c, int n_unique, int need, int lvl_lo, int lvl_hi, hipStream_t st) {
  if (lvl_lo > lvl_hi) return ELLC_OK;
  PrepArgs a;
  a.need = need;
  a.geom = c->geom_d;
  a.kf_tab = c->kf_tab_d;
  a.slots = c->uniq_slot_d;
  a.levels = c->L;
  a.max_kf = c->cfg.max_keyframes;
  for (int l = 0; l <= ELLC_MAX_LEVELS; l++) a.tile_begin[l] = c->tile_begin[std::min(l, c->L)];
  a.tile0 = c->tile_begin[lvl_lo];
  a.level0 = lvl_lo;
  const int tiles = c->tile_begin[lvl_hi + 1] - c->tile_begin[lvl_lo];
  a.slot_inline = 0;
  a.lb_tag = 0;
  if (c->direct_launch && n_unique <= 2 && lvl_lo == 0 && lvl_hi == c->L - 1 && tiles * n_unique <= c->resident_blocks) {
    c->prep_tag = c->prep_tag % 0xfffffu + 1u;   // never 0, never what a count launch leaves in a word (its upper bits are 0)
    a.lb_tag = c->prep_tag;
  } else {
    hipLaunchKernelGGL(prep_count, dim3(tiles, n_unique), dim3(256), 0, st, a);
  }
  switch (need) {
    case 1: hipLaunchKernelGGL(prep_scatter<1>, dim3(tiles, n_unique), dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL(prep_scatter<2>, dim3(tiles, n_unique), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL(prep_scatter<4>, dim3(tiles, n_unique), dim3(256), 0, st, a); break;
    case 20: hipLaunchKernelGGL(prep_scatter<20>, dim3(tiles, n_unique), dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL(prep_scatter<16>, dim3(tiles, n_unique), dim3(256), 0, st, a); break;   // fast ICA records only: the slots' H^-1 are current
    case 8: hipLaunchKernelGGL(prep_scatter<8>, dim3(tiles, n_unique), dim3(256), 0, st, a); break;
    default: return fail(c, ELLC_ERR_BAD_ARG, "run_prep: unknown record set");
  }
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

ellc_status run_prep(ellc_ctx* c, int n_unique, int need) { return run_prep_levels(c, n_unique, need, 0, c->L - 1, c->stream); }

// ICA: H^-1 of every (unique keyframe, level) from the per-tile sums the compaction (need bit 2) left behind
static void enqueue_ica_hinv(ellc_ctx* c, int n_unique) {
  PrepArgs a;
  a.need = 4;
  a.geom = c->geom_d;
  a.kf_tab = c->kf_tab_d;
  a.slots = c->uniq_slot_d;
  a.slot_inline = 0;
  a.levels = c->L;
  a.max_kf = c->cfg.max_keyframes;
  for (int l = 0; l <= ELLC_MAX_LEVELS; l++) a.tile_begin[l] = c->tile_begin[std::min(l, c->L)];
  a.tile0 = 0;
  a.level0 = 0;
  a.lb_tag = 0;
  hipLaunchKernelGGL(ica_hinv, dim3(c->L, n_unique), dim3(ELLC_SOLVE_THREADS), 0, c->stream, a);
}

static bool slot_ok(int s, int n) { return s >= 0 && s < n; }

// The batch size the grids are chosen for. A full batch (B = max_batch) of a coalescing context may run alone or side by
// side with others in one launch: its block counts — they fix the order of the sums, i.e. the bits of the result — are those
// of a full group either way. B here is what a launch covers: one batch, or k full batches.
static int grid_batch(const ellc_ctx* c, int B) {
  // cfg.grid_batch: every call's grids are those of a batch of that size, whatever B is — a shard of a batch (one rank's
  // block of a loop-closure batch, ellc_shard_range) then produces the bits the whole batch produces on one GPU
  if (c->cfg.grid_batch > 0) return c->cfg.grid_batch;
  return (c->coalesce > 1 && B % c->cfg.max_batch == 0) ? c->cfg.max_batch * c->coalesce : B;
}

static GnArgs make_gn_args(ellc_ctx* c, int level, int B, int save_w, float* planes) {
  GnArgs a;
  a.geom = c->geom_d;
  a.kf_tab = c->kf_tab_d;
  a.fr_tab = c->fr_tab_d;
  a.kf_slot = c->kf_slot_d;
  a.fr_slot = c->fr_slot_d;
  a.state = c->state_d;
  a.partials = c->partials_d;
  a.planes = planes;
  a.level = level;
  a.max_kf = c->cfg.max_keyframes;
  a.max_fr = c->cfg.max_frames;
  a.nblk = choose_nblk(c, level, grid_batch(c, B));
  a.save_w = save_w;
  return a;
}

static void launch_fca(ellc_ctx* c, dim3 grd, dim3 blk, const GnArgs& a) {
  if (c->fast) hipLaunchKernelGGL((gn_fca_accumulate<false, false, true>), grd, blk, 0, c->stream, a);
  else if (c->geom_h[0].divc_ok) hipLaunchKernelGGL((gn_fca_accumulate<false, true>), grd, blk, 0, c->stream, a);
  else hipLaunchKernelGGL((gn_fca_accumulate<false, false>), grd, blk, 0, c->stream, a);
}

// Exhaustive check of div_const(a, b, RN(1/b)) == a / b for every f32 mantissa of a (division commutes with the
// power-of-two scaling of a and of b, so one binade of a covers all normal inputs and every pyramid level).
#if defined(__x86_64__)
__attribute__((target("fma")))
#endif
static bool verify_div_const_fma(float b) {
  const float rb = (float)(1.0 / (double)b);
  for (uint32_t m = 0; m < (1u << 23); m++) {
    uint32_t bits = 0x3f800000u | m;
    float a;
    std::memcpy(&a, &bits, 4);
    const float q = a * rb;
    const float e = __builtin_fmaf(-b, q, a);
    const float r = __builtin_fmaf(e, rb, q);
    if (r != a / b) return false;
  }
  return true;
}
static bool verify_div_const(float b) {
  static std::map<uint32_t, bool> cache;
  static std::mutex cache_mutex;   // contexts may be created from several threads
  std::lock_guard<std::mutex> lock(cache_mutex);
  if (!(b > 0.0f) || !std::isfinite(b)) return false;
#if defined(__x86_64__)
  if (!__builtin_cpu_supports("fma")) return false;   // no hardware fma on this host: keep the plain divisions
#elif !defined(__FP_FAST_FMAF)
  return false;                                       // host without a known-fused fmaf: keep the plain divisions
#endif
  uint32_t bits;
  std::memcpy(&bits, &b, 4);
  bits &= 0x007fffffu;   // mantissa only
  auto it = cache.find(bits);
  if (it != cache.end()) return it->second;
  const bool ok = verify_div_const_fma(b);
  cache[bits] = ok;
  return ok;
}

static void launch_solve(ellc_ctx* c, int level, int B, int nblk, int mode, int early_exit) {
  SolveArgs s;
  s.state = c->state_d;
  s.partials = c->partials_d;
  s.level = level;
  s.nblk = nblk;
  s.mode = mode;
  s.early_exit = early_exit;
  if (c->fast) hipLaunchKernelGGL(gn_solve<true>, dim3(B), dim3(ELLC_SOLVE_THREADS), 0, c->stream, s);
  else hipLaunchKernelGGL(gn_solve<false>, dim3(B), dim3(ELLC_SOLVE_THREADS), 0, c->stream, s);
}

// points the staging / result / work-buffer members at batch set p: the host-side ones at batch `slice` of the group (where a
// batch is staged and its results are read), the device-side ones at the whole group (what a launch covers)
static void select_batch_set(ellc_ctx* c, int p, int slice = 0) {
  const int MB = c->cfg.max_batch, cap = c->group_cap;
  c->cur_set = p;
  ellc_ctx::BatchSet& bs = c->batch_set[p];
  c->kf_slot_h = bs.stage_h + slice * MB;
  c->fr_slot_h = bs.stage_h + cap + slice * MB;
  c->uniq_slot_h = bs.stage_h + 2 * cap;
  c->init_pose_h = (float*)(bs.stage_h + 3 * cap) + 6 * slice * MB;
  c->stage_dev_alias = bs.stage_dev_alias;
  c->result_h = bs.result_h + slice * MB;
  c->result_dev_alias = bs.result_dev_alias;
  c->kf_slot_d = bs.stage_d;
  c->fr_slot_d = bs.stage_d + cap;
  c->uniq_slot_d = bs.stage_d + 2 * cap;
  c->init_pose_d = (float*)(bs.stage_d + 3 * cap);
  c->state_d = bs.state_d;
  c->partials_d = bs.partials_d;
  c->persist_bar_d = bs.persist_bar_d;
}

// stage slots / initial poses on the device and list the unique keyframe slots
// (the selected set / slice; the unique slots of the batch go to `uniq` when given, else to the set's list: a batch that is
// launched by itself)
static ellc_status stage_batch(ellc_ctx* c, int B, const int* kf_slots, const int* frame_slots, const float* init_pose, int* n_unique,
                               std::vector<int>* uniq = nullptr, bool whole_group = false) {
  if (B < 1 || B > (whole_group ? c->group_cap : c->cfg.max_batch)) return fail(c, ELLC_ERR_BAD_ARG, "B out of range");
  if (!kf_slots || !frame_slots) return fail(c, ELLC_ERR_BAD_ARG, "null slot arrays");
  for (int b = 0; b < B; b++) {   // validated first: a refused batch leaves the set as it was
    if (!slot_ok(kf_slots[b], c->cfg.max_keyframes) || !slot_ok(frame_slots[b], c->cfg.max_frames))
      return fail(c, ELLC_ERR_BAD_ARG, "slot index out of range");
    if (!c->kf_has_image[kf_slots[b]] || !c->kf_has_depth[kf_slots[b]]) return fail(c, ELLC_ERR_NOT_READY, "keyframe slot lacks image or depth");
    if (!c->fr_has_image[frame_slots[b]]) return fail(c, ELLC_ERR_NOT_READY, "frame slot lacks image");
  }
  std::vector<int> local_uniq;
  std::vector<int>& un = uniq ? *uniq : local_uniq;
  un.clear();
  int nu = 0;
  for (int b = 0; b < B; b++) {
    if (!slot_ok(kf_slots[b], c->cfg.max_keyframes) || !slot_ok(frame_slots[b], c->cfg.max_frames))
      return fail(c, ELLC_ERR_BAD_ARG, "slot index out of range");
    if (!c->kf_has_image[kf_slots[b]] || !c->kf_has_depth[kf_slots[b]]) return fail(c, ELLC_ERR_NOT_READY, "keyframe slot lacks image or depth");
    if (!c->fr_has_image[frame_slots[b]]) return fail(c, ELLC_ERR_NOT_READY, "frame slot lacks image");
    c->kf_slot_h[b] = kf_slots[b];
    c->fr_slot_h[b] = frame_slots[b];
    bool seen = false;
    for (int u = 0; u < nu; u++) seen = seen || (un[u] == kf_slots[b]);
    if (!seen) { un.push_back(kf_slots[b]); nu++; }
    for (int i = 0; i < 6; i++) c->init_pose_h[b * 6 + i] = init_pose ? init_pose[b * 6 + i] : 0.0f;
    c->result_h[b].pad = -1;   // the kernel that exports the result clears it
  }
  // The pinned staging record is read by the first kernel of the schedule (enqueue_stage_in); it may still be in flight
  // from the previous call only if the caller skipped the fetch: ellc_align always fetches (synchronises),
  // ellc_align_enqueue callers must fetch before enqueueing again.
  if (!uniq)
    for (int u = 0; u < nu; u++) c->uniq_slot_h[u] = un[u];
  *n_unique = nu;
  return ELLC_OK;
}

// device copy of the staged batch description, as a kernel reading the pinned record (part of the captured graph)
static void enqueue_stage_in(ellc_ctx* c, int B) {
  c->stage_folded = false;
  if (c->direct_launch && B <= 2 && c->cur_resident && c->direct_nu == 0 && c->fold_staging) {
    // the tracking call whose lists are already there (built behind the export): no staging launch — the resident launch builds the
    // state records itself and takes the batch description and the seeds count along (PersistStage, enqueue_schedule_persist)
    c->stage_folded = true;
    return;
  }
  if (c->direct_launch && B <= 2) {   // not being captured: the record travels in the kernel arguments (stage_in_args)
    StageSmall ss;
    for (int b = 0; b < 2; b++) {
      ss.kf[b] = b < B ? c->kf_slot_h[b] : 0;
      ss.fr[b] = b < B ? c->fr_slot_h[b] : 0;
      ss.uniq[b] = b < c->direct_nu ? c->uniq_slot_h[b] : 0;   // the slots whose lists this launch (re)builds (launch_group)
      for (int i = 0; i < 6; i++) ss.pose[b * 6 + i] = b < B ? c->init_pose_h[b * 6 + i] : 0.0f;
    }
    // (ellc_track_frame's count of the valid hypotheses rides along in further blocks of the same launch)
    const int count_blocks = c->track_count_n > 0 ? std::max(1, ((c->track_count_n >> 4) + 1023) / 1024) : 0;
    hipLaunchKernelGGL(stage_in_args, dim3(1 + count_blocks), dim3(1024), 0, c->stream, c->kf_slot_d, ss, B, c->direct_nu, c->group_cap, c->state_d, c->L - 1,
                       (const uint8_t*)c->track_count_valid, c->track_count_n, c->seed_acc, c->track_dev_alias);
    c->track_count_n = 0;   // done (the caller launches the count by itself if this launch did not take it)
    return;
  }
  const int n = 9 * c->group_cap;
  const int copy_blocks = (n + 255) / 256;
  hipLaunchKernelGGL(stage_in, dim3(copy_blocks + (B + 255) / 256), dim3(256), 0, c->stream, c->kf_slot_d, c->stage_dev_alias, n, copy_blocks,
                     c->state_d, B, c->group_cap, c->L - 1);
}

// fills the age-balanced split of a launch (FusedArgs::age_rounds): on when the grid is 2..4 full rounds of one block per CU-slot
static void set_age_split(ellc_ctx* c, FusedArgs& fa, int B) {
  fa.age_rounds = 0;
  for (int i = 0; i < 5; i++) fa.age_cum[i] = 0;
  const int per_round = std::max(1, c->resident_blocks / 4);   // one block per CU
  // decided for the batch size the grids are chosen for (grid_batch): the split is part of what fixes a batch's bits, so a full
  // batch gets the same one whether it runs alone or side by side with others (the balance is tuned for the full group)
  const int total = fa.g.nblk * grid_batch(c, B);
  if (!c->age_balance || total % per_round != 0) return;
  const int R = total / per_round;
  if (R < 2 || R > 4 || fa.g.nblk % R != 0) return;
  // only throughput-bound launches gain (several pixels per thread); a light level finishes before the arbitration matters
  if (0.3 * c->geom_h[fa.g.level].n / (256.0 * fa.g.nblk) < c->age_min_px_per_thread) return;
  double sum = 0;
  for (int q = 0; q < R; q++) sum += c->age_weight[R][q];
  double cum = 0;
  for (int q = 0; q < R; q++) {
    fa.age_cum[q] = (int)(65536.0 * cum / sum + 0.5);
    cum += c->age_weight[R][q];
  }
  fa.age_cum[R] = 65536;
  fa.age_rounds = R;
}

static void launch_fused(ellc_ctx* c, dim3 grd, dim3 blk, const FusedArgs& fa, hipStream_t st) {
  const AlignState* src_state = fa.g.state + (size_t)(fa.seq & 1) * fa.stride_state;
  const float* prev_part = fa.g.partials + (size_t)((fa.seq + 1) & 1) * fa.stride_part;
  if (c->cur_dense) {   // dense maps: no compact lists (gn_fca_dense; launch_group decided)
    if (!c->fast) {   // the exact mode: a thread per pixel, the planes and the slot's 1 / Z plane in double (no 20-byte records)
      if (c->geom_h[0].divc_ok) hipLaunchKernelGGL(gn_fca_dense_x<true>, grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
      else hipLaunchKernelGGL(gn_fca_dense_x<false>, grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
      return;
    }
    // four adjacent pixels per thread where a row is a whole number of them (r06: one tap window per row for the four)
    if (dense_quads_at(c, fa.g.level))
      hipLaunchKernelGGL(gn_fca_dense4, grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
    else
      hipLaunchKernelGGL(gn_fca_dense, grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
    return;
  }
  if (c->fast) {
    if (fa.g.save_w) hipLaunchKernelGGL((gn_fca_fused<false, true, true, 1>), grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
    else hipLaunchKernelGGL((gn_fca_fused<false, true, true, 0>), grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
  } else if (c->pipe) {
    if (c->geom_h[0].divc_ok) hipLaunchKernelGGL((gn_fca_fused<true, true>), grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
    else hipLaunchKernelGGL((gn_fca_fused<false, true>), grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
  } else {
    if (c->geom_h[0].divc_ok) hipLaunchKernelGGL((gn_fca_fused<true, false>), grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
    else hipLaunchKernelGGL((gn_fca_fused<false, false>), grd, blk, 0, st, src_state, prev_part, fa.prev_nblk, fa.g.nblk, fa.age_rounds, fa);
  }
}

static void launch_finish(ellc_ctx* c, int B, const FusedArgs& fa, bool adaptive = false) {
  if (adaptive) {
    if (c->fast) hipLaunchKernelGGL((gn_fused_finish<true, true>), dim3(B), dim3(ELLC_SOLVE_THREADS), 0, c->stream, fa);
    else hipLaunchKernelGGL((gn_fused_finish<false, true>), dim3(B), dim3(ELLC_SOLVE_THREADS), 0, c->stream, fa);
  } else {
    if (c->fast) hipLaunchKernelGGL(gn_fused_finish<true>, dim3(B), dim3(ELLC_SOLVE_THREADS), 0, c->stream, fa);
    else hipLaunchKernelGGL(gn_fused_finish<false>, dim3(B), dim3(ELLC_SOLVE_THREADS), 0, c->stream, fa);
  }
}

// saved weights of every level, after the finish kernel (gn_add_saved_weights_all reads the record it wrote)
static void launch_add_saved_weights(ellc_ctx* c, int B) {
  // (blocks per alignment and level: enough for ~2 records per thread at level 0 of a semi-dense map when the batch is small —
  // the loop is a chain of three dependent memory operations per record)
  hipLaunchKernelGGL(gn_add_saved_weights_all, dim3(B <= 4 ? 128 : 32, B, c->L), dim3(256), 0, c->stream, c->kf_tab_d, c->kf_slot_d, c->geom_d, c->state_d,
                     c->cfg.max_keyframes, c->fast ? 1 : 0);
}

// The FCA schedule is state-driven (gn_fca_adaptive) for one or two alignments in contexts with early exit on — the tracking
// call; level-bound launches otherwise. Measured (r02, 640x480, fast, ms per batch, state-driven / level-bound,
// tools/dbg/early_exit_modes.py): B = 1 0.146 / 0.181, B = 2 0.158 / 0.195, B = 4 0.215 / 0.215, B = 8 0.237 / 0.241,
// B = 32 0.371 / 0.334 — a batch ends with its slowest alignment, every launch carries the finest level's grid, and the
// age-balanced split of the level-bound launches is lost.
static bool schedule_is_adaptive(const ellc_ctx* c, int mode, int B) {
  if (!(c->use_fused && c->use_adaptive && c->cfg.early_exit && mode == ELLC_MODE_FCA && B <= c->adaptive_max_batch)) return false;
  if (c->cfg.grid_batch > 0) return false;   // fixed grids: every call runs the level-bound schedule of the whole batch (same split, same bits)
  for (int l = 0; l < c->L; l++)
    if (c->cfg.max_iter[l] < 1) return false;   // a level without iterations: the level-bound schedule simply has no launch for it
  return true;
}
// Eager lists (r06): a tracked frame's alignment was preceded by the staging kernel AND the compaction of the keyframe's planes
// (prep_scatter: 6.7 us, and a kernel boundary of ~5 us) although those planes were written by the PREVIOUS frame's export launch,
// which runs beside the next frame's upload (tools/dbg/track_kernels.sh). In a context that tracks (state-driven schedule) the
// export therefore builds the tracking call's lists right behind itself — the same kernel, the count-free form, the same order and
// chunks, hence the same bits — and marks them as the cached lists of cfg.cache_records are marked: the next alignment against
// that keyframe finds them (launch_group) and starts with its first Gauss-Newton launch; any other writer of the slot's planes
// clears the mark (invalidate_records). Matches Frame.cpp:295-301, 316-327 fed by DepthPropagation.cpp:1254-1315, 1637-1746.
ellc_status enqueue_eager_lists(ellc_ctx* c, int slot) {
  if (!c->eager_lists || !schedule_is_adaptive(c, ELLC_MODE_FCA, 1)) return ELLC_OK;
  const int tiles = c->tile_begin[c->L] - c->tile_begin[0];
  if (tiles > c->resident_blocks) return ELLC_OK;   // (the count-free form needs the tiles a block waits for resident or done)
  const int need = c->fast ? 8 : 2;   // the FCA record set of the context's arithmetic mode (need_of)
  PrepArgs a;
  a.need = need;
  a.geom = c->geom_d;
  a.kf_tab = c->kf_tab_d;
  a.slots = nullptr;
  a.slot_inline = slot;
  a.levels = c->L;
  a.max_kf = c->cfg.max_keyframes;
  for (int l = 0; l <= ELLC_MAX_LEVELS; l++) a.tile_begin[l] = c->tile_begin[std::min(l, c->L)];
  a.tile0 = c->tile_begin[0];
  a.level0 = 0;
  c->prep_tag = c->prep_tag % 0xfffffu + 1u;
  a.lb_tag = c->prep_tag;
  if (need == 8) hipLaunchKernelGGL(prep_scatter<8>, dim3(tiles, 1), dim3(256), 0, c->stream, a);
  else if (need == 2) hipLaunchKernelGGL(prep_scatter<2>, dim3(tiles, 1), dim3(256), 0, c->stream, a);
  else return ELLC_OK;
  ELLC_HIP(c, hipGetLastError());
  c->kf_rec_tag[slot] = need;
  c->kf_rec_eager[slot] = 1;
  return ELLC_OK;
}

static int schedule_total_iters(const ellc_ctx* c) {
  int total = 0;
  for (int l = 0; l < c->L; l++) total += c->cfg.max_iter[l];
  return total;
}
// launches of the first graph: five eighths of the iteration caps (20 of {4,7,9,12}; tracked frames run 13-17 iterations); the
// continuation holds the rest
// — or, once the context has run such a call, what the previous one needed plus two (adaptive_hint: consecutive frames of a
// tracked sequence need about the same; r03: 20 launches of which a tracked frame used 15, the other five still cost 4.8 us each)
static int adaptive_first_launches(const ellc_ctx* c, int B) {
  if (c->cur_resident) return 0;   // one resident launch runs the whole schedule; a continuation (only after an abandoned launch) holds all of it
  const int total = schedule_total_iters(c);
  int first = c->adaptive_hint > 0 ? c->adaptive_hint : (total * 5 + 7) / 8;
#ifdef ELLC_DIAG
  if (c->adaptive_first_override > 0) first = c->adaptive_first_override;   // ELLC_ADAPTIVE_FIRST
#endif
  return std::min(total, std::max(c->L, first));
}

// ellc_track_frame marks the alignment it enqueues (c->track_call): that schedule's finish kernel also builds the observation's
// matrices and sets the depth stages' gate. (A continuation does not: the host then runs the depth stages the usual way.)
static void set_track_fields(const ellc_ctx* c, FusedArgs& fa, bool continuation) {
  fa.host_polls = c->cur_pollable ? 1 : 0;
  const bool on = c->track_call && !continuation;
  fa.track_mats = on ? (ObsMats*)c->track_mats_d : nullptr;
  fa.track_gate = on ? c->track_gate_d : nullptr;
  for (int i = 0; i < 9; i++) fa.track_K[i] = c->Kmat[i];
}

// State-driven FCA schedule: `launches` launches of gn_fca_adaptive and the finish kernel. continuation: the records were
// left by an earlier graph of the same batch (buffer 0, nothing pending), otherwise by stage_in.
// The state-driven schedule as one resident launch (gn_fca_persist). Blocks per alignment and level are the launch-per-iteration
// schedule's own (the grid is the largest of them, ELLC_NBLK_MAX at most), so the sums are grouped and combined exactly as there:
// the two forms, and a schedule that starts in one and is finished in the other, give the same bits. (Measured with the counts
// capped at 32 / 64 / 128 / 256 blocks, one early-exit alignment 640x480, fast: 0.120 / 0.116 / 0.115 / 0.117 ms against 0.128 with
// launches; exact 0.171 / 0.161 / 0.156 / 0.158 against 0.164; tracked frame 0.208 / 0.192 / 0.188 / 0.190 against 0.205.)
// true: the batch's launch sequence is launched kernel by kernel; false: replayed from a captured graph
static bool launches_directly(const ellc_ctx* c, int mode, int B) {
  return !c->use_graph || (!c->graph_adaptive && (schedule_is_adaptive(c, mode, B) || B <= c->direct_max_batch));
}
// rounds of a resident launch: every iteration, a round per level change, the first and the last
static int persist_rounds(const ellc_ctx* c) { return schedule_total_iters(c) + c->L + 2; }
// May this call's state-driven schedule run as ONE resident launch? The round travels in the low byte of the records' tags
// (call epoch << 8 | round): a schedule of more than 255 rounds would carry into the epoch and make round 256 + k of one call
// the twin of round k of the next (cfg.max_iter has no upper bound) — such a schedule runs as launches. Not while a hipGraph is
// being captured either (diagnostic builds, ELLC_GRAPH_ADAPTIVE): a replay would repeat the epoch baked into its arguments, and
// with it every tag. And not for a while after a launch had to be abandoned (persist_backoff: another context or process holds
// part of the device; each attempt costs a poll limit).
static bool may_run_resident(ellc_ctx* c, int mode, int B) {
  if (!(c->use_persist && B <= 2 && schedule_is_adaptive(c, mode, B))) return false;
  if (persist_rounds(c) > 255) return false;
  if (!launches_directly(c, mode, B)) return false;
  if (c->persist_backoff > 0) { c->persist_backoff--; return false; }
  return true;
}
// blocks a resident launch of B alignments needs on the device at once
static int persist_blocks(ellc_ctx* c, int B) {
  int G = 1;
  for (int l = 0; l < c->L; l++) G = std::max(G, choose_nblk(c, l, grid_batch(c, B)));
  return G * B;
}
static ellc_status enqueue_schedule_persist(ellc_ctx* c, int B, int save_weights) {
  FusedArgs fa;
  fa.continuation = 0;
  set_track_fields(c, fa, false);
  fa.seq = 0;
  fa.prev_level = -1;
  fa.prev_nblk = 0;
  fa.early_exit = c->cfg.early_exit;
  fa.stride_state = c->group_cap;
  fa.stride_part = (size_t)c->group_cap * ELLC_NBLK_MAX * ELLC_PART_STRIDE;
  fa.g = make_gn_args(c, 0, B, save_weights ? 1 : 0, nullptr);
  fa.res = c->result_dev_alias;
  fa.ica = 0;
  fa.xcd_map = 0;
  fa.age_rounds = 0;
  for (int i = 0; i < 5; i++) fa.age_cum[i] = 0;
  int G = 1;
  for (int l = 0; l < ELLC_MAX_LEVELS; l++) {
    fa.nblk_lv[l] = l < c->L ? choose_nblk(c, l, grid_batch(c, B)) : 1;
    fa.max_it[l] = l < c->L ? c->cfg.max_iter[l] : 0;
    G = std::max(G, fa.nblk_lv[l]);
  }
  fa.nblk_grid = G;
  fa.persist_bar = c->persist_bar_d;
  const int max_rounds = persist_rounds(c);   // <= 255 (may_run_resident)
  const unsigned epoch = (++c->persist_epoch) & 0xffffffu;   // the records of earlier calls never match (the round sits in the low byte)
  c->persist_launches++;
  PersistStage ps;
  std::memset(&ps, 0, sizeof(ps));
  ps.persist_blocks = G;
  if (c->stage_folded) {   // (enqueue_stage_in left the staging to this launch)
    ps.on = 1;
    for (int b = 0; b < 2; b++) {
      ps.s.kf[b] = b < B ? c->kf_slot_h[b] : 0;
      ps.s.fr[b] = b < B ? c->fr_slot_h[b] : 0;
      ps.s.uniq[b] = 0;
      for (int i = 0; i < 6; i++) ps.s.pose[b * 6 + i] = b < B ? c->init_pose_h[b * 6 + i] : 0.0f;
    }
    ps.dst = c->kf_slot_d;
    ps.cap = c->group_cap;
    ps.top_level = c->L - 1;
    ps.count_valid = (const uint8_t*)c->track_count_valid;
    ps.count_n = c->track_count_n;
    ps.count_blocks = c->track_count_n > 0 ? std::max(1, ((c->track_count_n >> 4) + ELLC_GN_THREADS - 1) / ELLC_GN_THREADS) : 0;
    ps.count_acc = c->seed_acc;
    ps.count_host = c->track_dev_alias;
    c->track_count_n = 0;   // (taken along)
    c->stage_folded = false;
  }
  const dim3 grd(G + ps.count_blocks, B), blk(ELLC_GN_THREADS);
  if (c->fast) {
    if (save_weights) hipLaunchKernelGGL((gn_fca_persist<false, true, 1>), grd, blk, 0, c->stream, fa, max_rounds, epoch, c->persist_spin_limit, c->persist_delay_from, c->persist_delay_polls, ps);
    else hipLaunchKernelGGL((gn_fca_persist<false, true, 0>), grd, blk, 0, c->stream, fa, max_rounds, epoch, c->persist_spin_limit, c->persist_delay_from, c->persist_delay_polls, ps);
  } else if (c->geom_h[0].divc_ok) {
    hipLaunchKernelGGL((gn_fca_persist<true, false, -1>), grd, blk, 0, c->stream, fa, max_rounds, epoch, c->persist_spin_limit, c->persist_delay_from, c->persist_delay_polls, ps);
  } else {
    hipLaunchKernelGGL((gn_fca_persist<false, false, -1>), grd, blk, 0, c->stream, fa, max_rounds, epoch, c->persist_spin_limit, c->persist_delay_from, c->persist_delay_polls, ps);
  }
  // (no finish kernel: the launch's first block per alignment has written the final record, the result and the tracking fields)
  if (save_weights) {
    // (the tracking call: its observation's selection launch, the next one in this stream, takes them along — launch_observe)
    if (c->track_call && B == 1 && c->ride_saved_weights) c->track_ride_weights = true;
    else launch_add_saved_weights(c, B);
  }
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

static ellc_status enqueue_schedule_adaptive(ellc_ctx* c, int B, int save_weights, int launches, bool continuation = false) {
  if (!continuation && c->cur_resident) return enqueue_schedule_persist(c, B, save_weights);
  FusedArgs fa;
  fa.continuation = continuation ? 1 : 0;
  set_track_fields(c, fa, continuation);
  fa.seq = 0;
  fa.prev_level = -1;
  fa.prev_nblk = 0;
  fa.early_exit = c->cfg.early_exit;
  fa.stride_state = c->group_cap;
  fa.stride_part = (size_t)c->group_cap * ELLC_NBLK_MAX * ELLC_PART_STRIDE;
  fa.g = make_gn_args(c, 0, B, save_weights ? 1 : 0, nullptr);
  fa.res = c->result_dev_alias;
  fa.ica = 0;
  fa.xcd_map = (B % 8 == 0) ? 1 : 0;
  fa.age_rounds = 0;
  for (int i = 0; i < 5; i++) fa.age_cum[i] = 0;
  int grid_x = 1;
  for (int l = 0; l < ELLC_MAX_LEVELS; l++) {
    fa.nblk_lv[l] = l < c->L ? choose_nblk(c, l, grid_batch(c, B)) : 1;
    fa.max_it[l] = l < c->L ? c->cfg.max_iter[l] : 0;
    grid_x = std::max(grid_x, fa.nblk_lv[l]);
  }
  fa.nblk_grid = grid_x;
  const dim3 grd(grid_x, B), blk(ELLC_GN_THREADS);
  for (int i = 0; i < launches; i++) {
    const AlignState* src_state = fa.g.state + (size_t)(fa.seq & 1) * fa.stride_state;
    const float* prev_part = fa.g.partials + (size_t)((fa.seq + 1) & 1) * fa.stride_part;
    if (c->fast) {
      if (save_weights) hipLaunchKernelGGL((gn_fca_adaptive<false, true, 1>), grd, blk, 0, c->stream, src_state, prev_part, grid_x, fa);
      else hipLaunchKernelGGL((gn_fca_adaptive<false, true, 0>), grd, blk, 0, c->stream, src_state, prev_part, grid_x, fa);
    } else if (c->geom_h[0].divc_ok) {
      hipLaunchKernelGGL((gn_fca_adaptive<true, false, -1>), grd, blk, 0, c->stream, src_state, prev_part, grid_x, fa);
    } else {
      hipLaunchKernelGGL((gn_fca_adaptive<false, false, -1>), grd, blk, 0, c->stream, src_state, prev_part, grid_x, fa);
    }
    fa.seq++;
  }
  launch_finish(c, B, fa, true);
  if (save_weights) launch_add_saved_weights(c, B);
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

// FCA schedule: the solve of iteration n is folded into the prologue of launch n+1 (gn_fca_fused): one launch per
// Gauss-Newton iteration plus one final solve. (r01 experiments that did not pay and were removed: cutting the batch
// into independent chains on parallel graph branches — the queues interleave poorly and the per-node submission cost
// dominates; compacting the fine levels on a second stream beside the coarse iterations — fork/join cost more than the
// overlap gave; a one-block-per-alignment kernel running all coarse iterations — one CU is VALU-bound on a level.)
static ellc_status enqueue_schedule_fused(ellc_ctx* c, int B, int save_weights) {
  FusedArgs fa;
  fa.continuation = 0;
  set_track_fields(c, fa, false);
  fa.seq = 0;
  fa.prev_level = -1;
  fa.prev_nblk = 0;
  fa.early_exit = c->cfg.early_exit;
  fa.stride_state = c->group_cap;
  fa.stride_part = (size_t)c->group_cap * ELLC_NBLK_MAX * ELLC_PART_STRIDE;
  fa.g = make_gn_args(c, 0, B, save_weights ? 1 : 0, nullptr);
  fa.res = c->result_dev_alias;
  fa.ica = 0;
  fa.xcd_map = (B % 8 == 0) ? 1 : 0;
  for (int l = 0; l < ELLC_MAX_LEVELS; l++) { fa.nblk_lv[l] = 1; fa.max_it[l] = 0; }
  fa.nblk_grid = 1;
  fa.age_rounds = 0;
  for (int i = 0; i < 5; i++) fa.age_cum[i] = 0;
  for (int level = c->L - 1; level >= 0; level--) {
    fa.g = make_gn_args(c, level, B, save_weights ? 1 : 0, nullptr);
    set_age_split(c, fa, B);
    const dim3 grd(fa.g.nblk, B), blk(ELLC_GN_THREADS);
    for (int it = 0; it < c->cfg.max_iter[level]; it++) {
      launch_fused(c, grd, blk, fa, c->stream);
      fa.prev_level = level;
      fa.prev_nblk = fa.g.nblk;
      fa.seq++;
    }
  }
  launch_finish(c, B, fa);
  if (save_weights) launch_add_saved_weights(c, B);
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

// Constant-weight (ICA) schedule in the fused form: H^-1 per (keyframe, level) comes from the compaction (ica_hinv), every
// launch solves the previous launch's b sums in its prologue: one launch per iteration plus the final solve.
static ellc_status enqueue_schedule_ica_fused(ellc_ctx* c, int B) {
  FusedArgs fa;
  fa.continuation = 0;
  set_track_fields(c, fa, false);
  fa.seq = 0;
  fa.prev_level = -1;
  fa.prev_nblk = 0;
  fa.early_exit = c->cfg.early_exit;
  fa.stride_state = c->group_cap;
  fa.stride_part = (size_t)c->group_cap * ELLC_NBLK_MAX * ELLC_PART_STRIDE;
  fa.res = c->result_dev_alias;
  fa.ica = 1;
  fa.xcd_map = 0;
  fa.age_rounds = 0;
  for (int i = 0; i < 5; i++) fa.age_cum[i] = 0;
  fa.g = make_gn_args(c, 0, B, 0, nullptr);
  for (int level = c->L - 1; level >= 0; level--) {
    fa.g = make_gn_args(c, level, B, 0, nullptr);
    const dim3 grd(fa.g.nblk, B), blk(ELLC_GN_THREADS);
    for (int it = 0; it < c->cfg.max_iter[level]; it++) {
      const AlignState* src_state = fa.g.state + (size_t)(fa.seq & 1) * fa.stride_state;
      const float* prev_part = fa.g.partials + (size_t)((fa.seq + 1) & 1) * fa.stride_part;
      if (c->fast) hipLaunchKernelGGL(gn_ica_fused<true>, grd, blk, 0, c->stream, src_state, prev_part, fa.prev_nblk, fa);
      else hipLaunchKernelGGL(gn_ica_fused<false>, grd, blk, 0, c->stream, src_state, prev_part, fa.prev_nblk, fa);
      fa.prev_level = level;
      fa.prev_nblk = fa.g.nblk;
      fa.seq++;
    }
  }
  launch_finish(c, B, fa);
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

// the level / iteration schedule of GetImagePoseEstimate (ImageFunc.cpp:150-292) as a launch sequence
static ellc_status enqueue_schedule(ellc_ctx* c, int B, int mode, int save_weights) {
  if (schedule_is_adaptive(c, mode, B)) return enqueue_schedule_adaptive(c, B, save_weights, c->cur_adaptive_first);
  if (mode == ELLC_MODE_FCA && c->use_fused) return enqueue_schedule_fused(c, B, save_weights);
  if (mode == ELLC_MODE_ICA && c->use_fused) return enqueue_schedule_ica_fused(c, B);
  for (int level = c->L - 1; level >= 0; level--) {
    GnArgs a = make_gn_args(c, level, B, (save_weights && mode == ELLC_MODE_FCA) ? 1 : 0, nullptr);
    const dim3 grd(a.nblk, B), blk(ELLC_GN_THREADS);
    for (int it = 0; it < c->cfg.max_iter[level]; it++) {
      if (mode == ELLC_MODE_FCA) {
        launch_fca(c, grd, blk, a);
        launch_solve(c, level, B, a.nblk, 0, c->cfg.early_exit);
      } else {
        if (it == 0) {
          hipLaunchKernelGGL(gn_ica_precompute, grd, blk, 0, c->stream, a, c->cap[level]);
          launch_solve(c, level, B, a.nblk, 1, 0);
        }
        hipLaunchKernelGGL(gn_ica_iterate<false>, grd, blk, 0, c->stream, a, c->cap[level]);
        launch_solve(c, level, B, a.nblk, 2, c->cfg.early_exit);
      }
    }
    if (save_weights && mode == ELLC_MODE_FCA) {
      hipLaunchKernelGGL(gn_add_saved_weights, dim3(64, B), dim3(256), 0, c->stream, c->kf_tab_d, c->kf_slot_d, c->geom_d, level,
                         c->cfg.max_keyframes, c->fast ? 1 : 0);
    }
  }
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

}  // namespace ellc

// =====================================================================================================
extern "C" {

int ellc_abi_version(void) { return ELLC_ABI_VERSION; }

int ellc_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

void ellc_default_config(ellc_config* cfg, int width, int height, int levels) {
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->width = width;
  cfg->height = height;
  cfg->levels = levels;
  cfg->fx = cfg->fy = 0.855f * (float)width;   // same fx/W ratio as ExternVariable.h:53 (410.6/480)
  cfg->cx = (float)width / 2.0f;
  cfg->cy = (float)height / 2.0f;
  const int mi[ELLC_MAX_LEVELS] = {4, 7, 9, 12, 12, 12, 12, 12};   // main.cpp:34
  for (int i = 0; i < ELLC_MAX_LEVELS; i++) cfg->max_iter[i] = mi[i];
  cfg->early_exit = 1;
  cfg->max_keyframes = 4;
  cfg->max_frames = 4;
  cfg->max_batch = 4;
  cfg->device = 0;
  cfg->concurrent_batches = 1;
  cfg->coalesce = 1;
  cfg->cache_records = 0;
  cfg->grid_batch = 0;
}

const char* ellc_last_error(const ellc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
void* ellc_stream(ellc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

ellc_status ellc_sync(ellc_ctx* c) {
  ELLC_ENTER(c);
  if (!c) return ELLC_ERR_BAD_ARG;
  ELLC_HIP(c, hipStreamSynchronize(c->stream));   // ELLC_ENTER made it wait for every batch in flight
  return ELLC_OK;
}

ellc_status ellc_ctx_create(const ellc_config* cfg, ellc_ctx** out) {
  if (!cfg || !out) return ELLC_ERR_BAD_ARG;
  *out = nullptr;
  if (cfg->width < 16 || cfg->height < 16 || cfg->width > 65535 || cfg->height > 65535) return ELLC_ERR_BAD_ARG;
  // the kernels index a plane with 32-bit offsets (24 * i and 48 * i byte offsets into the record lists) and the compaction
  // converts a pixel index to f32 exactly: planes of at most 2^24 pixels
  if ((long long)cfg->width * cfg->height > (1ll << 24)) return ELLC_ERR_BAD_ARG;
  if (cfg->arith != ELLC_ARITH_EXACT && cfg->arith != ELLC_ARITH_FAST) return ELLC_ERR_BAD_ARG;
  if (cfg->width > 4096 || cfg->height > 4096) return ELLC_ERR_BAD_ARG;   // 12-bit x / y in the compact records (FcaRec, FcaRecF)
  if (cfg->levels < 1 || cfg->levels > ELLC_MAX_LEVELS) return ELLC_ERR_BAD_ARG;
  if ((cfg->width >> (cfg->levels - 1)) < 4 || (cfg->height >> (cfg->levels - 1)) < 4) return ELLC_ERR_BAD_ARG;
  if (cfg->max_keyframes < 1 || cfg->max_frames < 1 || cfg->max_batch < 1) return ELLC_ERR_BAD_ARG;
  if (cfg->coalesce < 0 || cfg->coalesce > ellc_ctx::MAX_COALESCE) return ELLC_ERR_BAD_ARG;   // 0: as 1
  if (cfg->concurrent_batches < 0 || cfg->concurrent_batches > (ellc_ctx::STREAMS + 1) * ellc_ctx::MAX_COALESCE) return ELLC_ERR_BAD_ARG;   // 0: as 1
  if (cfg->grid_batch < 0 || cfg->grid_batch > 65536) return ELLC_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ELLC_ERR_NO_DEVICE;
  if (cfg->device < 0 || cfg->device >= ndev) return ELLC_ERR_BAD_ARG;
  ellc_ctx* c = new ellc_ctx();
  c->cfg = *cfg;
  c->L = cfg->levels;
  c->fast = (cfg->arith == ELLC_ARITH_FAST);
  if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
    delete c;
    return ELLC_ERR_HIP;
  }
#define TRY(expr) do { ellc_status s__ = (expr); if (s__ != ELLC_OK) { *out = c; return s__; } } while (0)
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    *out = c;
    return fail(c, ELLC_ERR_HIP, "cannot create the timing events");
  }
  // ---- level geometry + Jacobian tables (UserDefinedFunc.cpp:34-50; PixelWisePyramid.cpp:296-303)
  int sw = cfg->width, sh = cfg->height;
  c->tile_begin[0] = 0;
  for (int l = 0; l < c->L; l++) {
    LevelGeom& g = c->geom_h[l];
    g.cols = cfg->width >> l;
    g.rows = cfg->height >> l;
    g.sw = sw;
    g.sh = sh;
    g.n = g.cols * g.rows;
    const double s = std::pow(2.0, l);
    g.fx = (float)((double)cfg->fx / s);
    g.fy = (float)((double)cfg->fy / s);
    g.cx = (float)((double)cfg->cx / s);
    g.cy = (float)((double)cfg->cy / s);
    g.rfx = (float)(1.0 / (double)g.fx);
    g.rfy = (float)(1.0 / (double)g.fy);
    g.divc_ok = (!c->fast && verify_div_const(cfg->fx) && verify_div_const(cfg->fy)) ? 1 : 0;   // the tolerance mode multiplies by rfx, rfy
#ifdef ELLC_DIAG
    if (const char* nd = getenv("ELLC_NO_DIVC")) if (nd[0] == '1') g.divc_ok = 0;
#endif
    std::vector<double> colA(g.cols), rowA(g.rows);
    std::vector<float> colB(g.cols), rowB(g.rows);
    for (int x = 0; x < g.cols; x++) {
      const float u = -g.cx + (float)x;
      colA[x] = (double)g.fx + (std::pow((double)u, 2) / (double)g.fx);
      colB[x] = (g.fy * u) / g.fx;
    }
    for (int y = 0; y < g.rows; y++) {
      const float v = -g.cy + (float)y;
      rowA[y] = -((double)g.fy + (std::pow((double)v, 2) / (double)g.fy));
      rowB[y] = -(g.fx * v / g.fy);
    }
    double *dA, *dR;
    float *dB, *dRB;
    TRY(dev_alloc(c, &dA, g.cols)); TRY(dev_alloc(c, &dR, g.rows)); TRY(dev_alloc(c, &dB, g.cols)); TRY(dev_alloc(c, &dRB, g.rows));
    if (copy_blocking(c, dA, colA.data(), g.cols * 8, hipMemcpyHostToDevice) != hipSuccess ||
        copy_blocking(c, dR, rowA.data(), g.rows * 8, hipMemcpyHostToDevice) != hipSuccess ||
        copy_blocking(c, dB, colB.data(), g.cols * 4, hipMemcpyHostToDevice) != hipSuccess ||
        copy_blocking(c, dRB, rowB.data(), g.rows * 4, hipMemcpyHostToDevice) != hipSuccess) {
      *out = c;
      return fail(c, ELLC_ERR_HIP, "cannot upload the Jacobian tables");
    }
    g.colA = dA; g.rowA = dR; g.colB = dB; g.rowB = dRB;
    c->cap[l] = g.n;
    c->tile_begin[l + 1] = c->tile_begin[l] + (g.n + ELLC_TILE - 1) / ELLC_TILE;
    sw = (sw + 1) / 2;
    sh = (sh + 1) / 2;
  }
  TRY(dev_alloc(c, &c->geom_d, ELLC_MAX_LEVELS));
  if (copy_blocking(c, c->geom_d, c->geom_h, sizeof(LevelGeom) * c->L, hipMemcpyHostToDevice) != hipSuccess) {
    *out = c;
    return fail(c, ELLC_ERR_HIP, "cannot upload the level geometry");
  }
  // ---- slots
  const int MK = cfg->max_keyframes, MF = cfg->max_frames;
  c->kf_tab_h.assign((size_t)c->L * MK, KfLevelDev());
  c->fr_tab_h.assign((size_t)c->L * MF, FrLevelDev());
  for (int l = 0; l < c->L; l++) {
    const LevelGeom& g = c->geom_h[l];
    const size_t n = g.n, ni = (size_t)g.sw * g.sh;
    const int tiles = c->tile_begin[l + 1] - c->tile_begin[l];
    for (int s = 0; s < MK; s++) {
      KfLevelDev& k = c->kf_tab_h[(size_t)l * MK + s];
      TRY(dev_alloc(c, &k.img, ni + 16));   // + 16: the window staging reads whole 16-byte words (stage_window), the last may reach past the image
      TRY(dev_alloc(c, &k.depth, n)); TRY(dev_alloc(c, &k.var, n)); TRY(dev_alloc(c, &k.weight, n));
      TRY(dev_alloc(c, &k.cxy, n)); TRY(dev_alloc(c, &k.cZ, n)); TRY(dev_alloc(c, &k.cI, n));
      TRY(dev_alloc(c, &k.crec, n)); TRY(dev_alloc(c, &k.cW, n)); TRY(dev_alloc(c, &k.wlast, n)); TRY(dev_alloc(c, &k.sd, 6 * n));
      TRY(dev_alloc(c, &k.count, 4)); TRY(dev_alloc(c, &k.tile_count, tiles + 1)); TRY(dev_alloc(c, &k.idepth, n));
      k.invz = nullptr;
      if (!c->fast) TRY(dev_alloc(c, &k.invz, n));   // (the exact mode's list-free kernel: KfLevelDev::invz)
      TRY(dev_alloc(c, &k.irec, n)); TRY(dev_alloc(c, &k.hpart, (size_t)(tiles + 1) * ELLC_PART_STRIDE)); TRY(dev_alloc(c, &k.hinv, 36));
    }
    // (+ one row: the fifth row of gn_fca_dense4's tap windows may be the one below the image)
    for (int s = 0; s < MF; s++) TRY(dev_alloc(c, &c->fr_tab_h[(size_t)l * MF + s].img, ni + (size_t)g.sw + 32));
  }
  TRY(dev_alloc(c, &c->kf_tab_d, c->kf_tab_h.size()));
  TRY(dev_alloc(c, &c->fr_tab_d, c->fr_tab_h.size()));
  if (copy_blocking(c, c->kf_tab_d, c->kf_tab_h.data(), c->kf_tab_h.size() * sizeof(KfLevelDev), hipMemcpyHostToDevice) != hipSuccess ||
      copy_blocking(c, c->fr_tab_d, c->fr_tab_h.data(), c->fr_tab_h.size() * sizeof(FrLevelDev), hipMemcpyHostToDevice) != hipSuccess) {
    *out = c;
    return fail(c, ELLC_ERR_HIP, "cannot upload the slot tables");
  }
  c->kf_has_image.assign(MK, 0); c->kf_has_depth.assign(MK, 0); c->fr_has_image.assign(MF, 0);
  c->kf_dense.assign(MK, 0);
  c->kf_num_weights.assign(MK, std::array<int, ELLC_MAX_LEVELS>{});
  c->kf_rec_tag.assign(MK, 0);
  c->kf_rec_eager.assign(MK, 0);
  c->kf_hinv_ok.assign(MK, 0);
  c->cache_records = cfg->cache_records != 0;
  c->kf_maxgrad.assign(MK, nullptr); c->fr_maxgrad.assign(MF, nullptr);
  c->kf_maxgrad_count.assign(MK, nullptr); c->fr_maxgrad_count.assign(MF, nullptr);
  c->kf_maxgrad_valid.assign(MK, 0); c->fr_maxgrad_valid.assign(MF, 0);
  const size_t n0 = (size_t)cfg->width * cfg->height;
  for (int s = 0; s < MK; s++) { TRY(dev_alloc(c, &c->kf_maxgrad[s], n0)); TRY(dev_alloc(c, &c->kf_maxgrad_count[s], 4)); }
  for (int s = 0; s < MF; s++) { TRY(dev_alloc(c, &c->fr_maxgrad[s], n0)); TRY(dev_alloc(c, &c->fr_maxgrad_count[s], 4)); }
  // ---- alignment work buffers
  const int MB = cfg->max_batch;
  c->coalesce = std::max(1, cfg->coalesce);
  c->group_cap = c->coalesce * MB;
  c->max_inflight = c->coalesce > 1 ? (ellc_ctx::STREAMS + 1) * c->coalesce : ellc_ctx::STREAMS;
  c->n_sets = c->max_inflight + 1;
  c->batch_stream[0] = c->stream;
  const size_t CAP = (size_t)c->group_cap;
  // per batch set (group): one staging record [kf_slot cap][fr_slot cap][unique cap][init_pose 6*cap] (pinned, and its device
  // copy), the result records (pinned), the alignment states (two launch-parity buffers), the block partials
  for (int p = 0; p < c->n_sets; p++) {
    ellc_ctx::BatchSet& bs = c->batch_set[p];
    TRY(dev_alloc(c, &bs.stage_d, 9 * CAP));
    TRY(host_alloc(c, &bs.stage_h, 9 * CAP));
    TRY(host_alloc(c, &bs.result_h, CAP));
    TRY(dev_alloc(c, &bs.state_d, 2 * CAP));
    TRY(dev_alloc(c, &bs.partials_d, 2 * CAP * ELLC_NBLK_MAX * ELLC_PART_STRIDE));
    TRY(dev_alloc(c, &bs.persist_bar_d, 2 * ELLC_PERSIST_BAR_WORDS));   // gn_fca_persist's abort words (device memory comes zeroed)
    void *da = nullptr, *db = nullptr;
    if (hipHostGetDevicePointer(&da, bs.stage_h, 0) != hipSuccess || hipHostGetDevicePointer(&db, bs.result_h, 0) != hipSuccess ||
        hipEventCreateWithFlags(&bs.done, hipEventDisableTiming) != hipSuccess) {
      *out = c;
      return fail(c, ELLC_ERR_HIP, "pinned host memory is not device-visible, or no event for a batch set");
    }
    bs.stage_dev_alias = (const int*)da;
    bs.result_dev_alias = (AlignResult*)db;
  }
  if (hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming) != hipSuccess) {
    *out = c;
    return fail(c, ELLC_ERR_HIP, "cannot create the ordering event");
  }
  select_batch_set(c, 0);
  TRY(host_alloc(c, &c->state_h, MB));
  TRY(dev_alloc(c, &c->planes_d, 10 * n0));
  TRY(dev_alloc(c, &c->scratch_a, n0)); TRY(dev_alloc(c, &c->scratch_b, n0));
  // ---- depth map state
  DepthSoA* ds[2] = {&c->dm_cur, &c->dm_oth};
  for (int i = 0; i < 2; i++) {
    TRY(dev_alloc(c, &ds[i]->invDepth, n0)); TRY(dev_alloc(c, &ds[i]->invDepthSmoothed, n0));
    TRY(dev_alloc(c, &ds[i]->variance, n0)); TRY(dev_alloc(c, &ds[i]->varianceSmoothed, n0));
    TRY(dev_alloc(c, &ds[i]->validity, n0)); TRY(dev_alloc(c, &ds[i]->blacklisted, n0)); TRY(dev_alloc(c, &ds[i]->isValid, n0));
  }
  TRY(dev_alloc(c, &c->dm_deptharr0, n0)); TRY(dev_alloc(c, &c->dm_vararr0, n0));
  TRY(dev_alloc(c, &c->pr_tgt, n0)); TRY(dev_alloc(c, &c->pr_cnt, n0)); TRY(dev_alloc(c, &c->pr_slots, 4 * n0)); TRY(dev_alloc(c, &c->pr_val, n0));
  TRY(dev_alloc(c, &c->pr_id, n0)); TRY(dev_alloc(c, &c->pr_var, n0)); TRY(dev_alloc(c, &c->pr_remaining, 4));
  TRY(dev_alloc(c, &c->red_scratch, 4096));
  TRY(dev_alloc(c, &c->sum_parts, 2 * (size_t)((cfg->width + 31) / 32) * ((cfg->height + 7) / 8)));   // the 32 x 8 tiles of the depth kernels
  TRY(dev_alloc(c, (char**)&c->track_mats_d, 256)); TRY(dev_alloc(c, &c->track_gate_d, 4)); TRY(dev_alloc(c, &c->seed_acc, 4));
  {   // (the arena is zeroed: the counters start at 0)
    const size_t blocks = (size_t)((cfg->width + 31) / 32) * ((cfg->height + 7) / 8);
    TRY(dev_alloc(c, &c->obs_list, ((blocks + DM_OBS_REGIONS - 1) / DM_OBS_REGIONS) * 256 * DM_OBS_REGIONS));
    TRY(dev_alloc(c, &c->obs_list_ep, ((blocks + DM_OBS_REGIONS - 1) / DM_OBS_REGIONS) * 256 * DM_OBS_REGIONS));
    TRY(dev_alloc(c, &c->obs_ctr, 4 * DM_OBS_REGIONS));   // two sets of counters, alternating from call to call (zeroed by the arena)
  }
  // K and Kinv (EigenInitialization.cpp:20-34): cv 3x3 f32 inverse = f32 cofactors scaled by 1/det in double
  {
    const float K[9] = {cfg->fx, 0, cfg->cx, 0, cfg->fy, cfg->cy, 0, 0, 1};
    std::memcpy(c->Kmat, K, sizeof(K));
#define S(i, j) K[(i) * 3 + (j)]
    double d = S(0, 0) * ((double)S(1, 1) * S(2, 2) - (double)S(1, 2) * S(2, 1)) - S(0, 1) * ((double)S(1, 0) * S(2, 2) - (double)S(1, 2) * S(2, 0)) +
               S(0, 2) * ((double)S(1, 0) * S(2, 1) - (double)S(1, 1) * S(2, 0));
    d = (d != 0.) ? 1. / d : 0.;
    c->Kinv[0] = (float)((S(1, 1) * S(2, 2) - S(1, 2) * S(2, 1)) * d);
    c->Kinv[1] = (float)((S(0, 2) * S(2, 1) - S(0, 1) * S(2, 2)) * d);
    c->Kinv[2] = (float)((S(0, 1) * S(1, 2) - S(0, 2) * S(1, 1)) * d);
    c->Kinv[3] = (float)((S(1, 2) * S(2, 0) - S(1, 0) * S(2, 2)) * d);
    c->Kinv[4] = (float)((S(0, 0) * S(2, 2) - S(0, 2) * S(2, 0)) * d);
    c->Kinv[5] = (float)((S(0, 2) * S(1, 0) - S(0, 0) * S(1, 2)) * d);
    c->Kinv[6] = (float)((S(1, 0) * S(2, 1) - S(1, 1) * S(2, 0)) * d);
    c->Kinv[7] = (float)((S(0, 1) * S(2, 0) - S(0, 0) * S(2, 1)) * d);
    c->Kinv[8] = (float)((S(0, 0) * S(1, 1) - S(0, 1) * S(1, 0)) * d);
#undef S
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) { *out = c; return fail(c, ELLC_ERR_HIP, "initial sync failed"); }
  {
    for (int l = 0; l < ELLC_MAX_LEVELS; l++) c->nblk_override[l] = 0;
#ifdef ELLC_DIAG
    // Diagnostic builds only (make diag -> build/libellc_hip_diag.so, loaded through ELLC_LIB_PATH by the tools): the
    // shipping library reads nothing from the environment.
    if (const char* nf = getenv("ELLC_NO_FUSE")) c->use_fused = !(nf[0] == '1');
    if (const char* ab = getenv("ELLC_NO_AGE_BALANCE")) c->age_balance = !(ab[0] == '1');
    if (const char* pp = getenv("ELLC_PIPE")) c->pipe = (pp[0] == '1');
    if (const char* am = getenv("ELLC_AGE_MIN_PX")) c->age_min_px_per_thread = atof(am);
    if (getenv("ELLC_NO_ADAPTIVE")) c->use_adaptive = false;
    if (getenv("ELLC_NO_PERSIST")) c->use_persist = false;
    if (const char* ab = getenv("ELLC_ADAPTIVE_MAX_BATCH")) c->adaptive_max_batch = atoi(ab);
    if (const char* af = getenv("ELLC_ADAPTIVE_FIRST")) c->adaptive_first_override = atoi(af);
    if (const char* aw = getenv("ELLC_AGE_W")) {   // "R:w0,w1,..": weights for grids of R rounds
      int R = 0, pos = 0;
      if (sscanf(aw, "%d:%n", &R, &pos) == 1 && R >= 2 && R <= 4) {
        const char* p = aw + pos;
        for (int q = 0; q < R && *p; q++) {
          c->age_weight[R][q] = atof(p);
          while (*p && *p != ',') p++;
          if (*p == ',') p++;
        }
      }
    }
    if (const char* ng = getenv("ELLC_NO_GRAPH")) c->use_graph = !(ng[0] == '1');
    if (const char* np = getenv("ELLC_NO_POLL")) c->poll_results = !(np[0] == '1');
    if (const char* ga = getenv("ELLC_GRAPH_ADAPTIVE")) c->graph_adaptive = (ga[0] == '1');
    if (const char* nb = getenv("ELLC_NBLK")) {
      int l = 0;
      for (const char* q = nb; *q && l < ELLC_MAX_LEVELS; l++) {
        c->nblk_override[l] = atoi(q);
        while (*q && *q != ',') q++;
        if (*q == ',') q++;
      }
    }
#endif
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0)
      c->resident_blocks = cus * (c->use_fused ? 4 : 5);   // 256-thread blocks per CU: 126 VGPRs (fused) -> 4 waves/SIMD, 92 -> 5
    // blocks of the resident schedule (gn_fca_persist) this device holds at once: a launch of more could never become resident
    // (a partition of the device, e.g. one XCD's 32 CUs) and is not attempted
    if (cus > 0) {
      // (the smaller occupancy of the two variants this context may launch: with / without saved weights, with / without the
      // verified division by a constant)
      int per_cu = 0, per_cu2 = 0;
      hipError_t oe = c->fast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gn_fca_persist<false, true, 1>, ELLC_GN_THREADS, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gn_fca_persist<false, false, -1>, ELLC_GN_THREADS, 0);
      if (oe == hipSuccess)
        oe = c->fast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, gn_fca_persist<false, true, 0>, ELLC_GN_THREADS, 0)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, gn_fca_persist<true, false, -1>, ELLC_GN_THREADS, 0);
      per_cu = std::min(per_cu, per_cu2);
      c->persist_capacity = (oe == hipSuccess && per_cu > 0) ? per_cu * cus : 0;
      if (oe != hipSuccess) (void)hipGetLastError();
    }
  }
#undef TRY
  *out = c;
  return ELLC_OK;
}

ellc_status ellc_ctx_destroy(ellc_ctx* c) {
  if (!c) return ELLC_ERR_BAD_ARG;
  (void)ellc::enter(c, true);   // launches an open group / resolves state-driven batches; whatever it reports, everything is torn down
  for (int i = 1; i < ellc_ctx::STREAMS; i++) if (c->batch_stream[i]) (void)hipStreamSynchronize(c->batch_stream[i]);
  (void)hipStreamSynchronize(c->stream);
#ifdef ELLC_QUAD_STATS   // diagnostic variant builds only (tools/quad_stats.sh): how often gn_fca_dense4's quads did not fit their window
  {
    unsigned long long qs[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(qs, HIP_SYMBOL(ellc::g_quad_stats), sizeof(qs)) == hipSuccess)
      std::fprintf(stderr, "ELLC_QUAD_STATS quads %llu nofit %llu (%.3f %%) queued_pixels %llu drain_rounds %llu\n", qs[0], qs[1],
                   qs[0] ? 100.0 * (double)qs[1] / (double)qs[0] : 0.0, qs[2], qs[3]);
  }
#endif
  for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.second);
  for (void* p : c->allocs) (void)hipFree(p);
  for (void* p : c->host_allocs) (void)hipHostFree(p);
  for (int k = 0; k < ellc_ctx::UPLOAD_RING; k++) {
    if (c->upload_stage[k]) (void)hipHostFree(c->upload_stage[k]);
    if (c->upload_done[k]) (void)hipEventDestroy(c->upload_done[k]);
  }
  if (c->ingest_map) (void)hipFree(c->ingest_map);
  if (c->ingest_bgr) (void)hipFree(c->ingest_bgr);
  for (int p = 0; p < ellc_ctx::SETS; p++)
    if (c->batch_set[p].done) (void)hipEventDestroy(c->batch_set[p].done);
  for (int i = 1; i < ellc_ctx::STREAMS; i++)
    if (c->batch_stream[i]) (void)hipStreamDestroy(c->batch_stream[i]);
  if (c->ev_main) (void)hipEventDestroy(c->ev_main);
  if (c->ev_xfer) (void)hipEventDestroy(c->ev_xfer);
  if (c->upload_stream) {
    (void)hipStreamSynchronize(c->upload_stream);
    for (hipEvent_t e : c->fr_use_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->fr_ready_ev) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->upload_stream);
  }
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return ELLC_OK;
}

// diagnostic counters of the context (how results were waited for, continuations run): see ellc_abi.h
ellc_status ellc_ctx_counters(ellc_ctx* c, long long* out, int n) {
  if (!c || !out || n < 0) return ELLC_ERR_BAD_ARG;
  for (int i = 0; i < n; i++) out[i] = i < ELLC_CTR_COUNT ? c->counters[i] : 0;
  return ELLC_OK;
}
// the time resolve_batch polls the result records before it falls back to the event (default 2000 us; tests shorten it)
ellc_status ellc_ctx_set_poll_timeout_us(ellc_ctx* c, int us) {
  if (!c || us < 0) return ELLC_ERR_BAD_ARG;
  c->poll_timeout_us = us;
  return ELLC_OK;
}

// ---- frame side ----------------------------------------------------------------------------------
ellc_status ellc_ctx_set_persistent_schedule(ellc_ctx* c, int mode) {
  if (!c) return ELLC_ERR_BAD_ARG;
  ELLC_ENTER(c);
  if (mode < 0 || mode > 2) return fail(c, ELLC_ERR_BAD_ARG, "ellc_ctx_set_persistent_schedule: mode 0, 1 or 2");
  c->use_persist = mode != 0;
  c->persist_spin_limit = mode == 2 ? 0u : ELLC_PERSIST_SPIN_LIMIT;
  c->adaptive_hint = 0;
  return ELLC_OK;
}

#ifdef ELLC_DIAG_ABI
// test hooks of the resident schedule (include/ellc_abi_diag.h; libellc_hip_diag.so only)
ellc_status ellc_debug_persist_delay(ellc_ctx* c, int first_block, int polls) {
  if (!c) return ELLC_ERR_BAD_ARG;
  ELLC_ENTER(c);
  if (first_block < 0 || polls < -255 || polls > (1 << 16)) return fail(c, ELLC_ERR_BAD_ARG, "ellc_debug_persist_delay: first_block >= 0, -255 <= polls <= 65536");
  c->persist_delay_from = first_block;
  c->persist_delay_polls = polls;
  return ELLC_OK;
}
ellc_status ellc_debug_set_persist_epoch(ellc_ctx* c, unsigned epoch) {
  if (!c) return ELLC_ERR_BAD_ARG;
  ELLC_ENTER(c);
  c->persist_epoch = epoch;
  return ELLC_OK;
}
ellc_status ellc_debug_set_eager_lists(ellc_ctx* c, int on) {
  if (!c) return ELLC_ERR_BAD_ARG;
  ELLC_ENTER(c);
  c->eager_lists = on != 0;
  if (!on) std::fill(c->kf_rec_eager.begin(), c->kf_rec_eager.end(), 0);
  return ELLC_OK;
}
ellc_status ellc_debug_set_fold_staging(ellc_ctx* c, int on) {
  if (!c) return ELLC_ERR_BAD_ARG;
  ELLC_ENTER(c);
  c->fold_staging = on != 0;
  c->ride_saved_weights = on != 0;   // (the other launch the tracking call has folded away: the saved weights in the selection launch)
  return ELLC_OK;
}
ellc_status ellc_debug_set_hinv_cache(ellc_ctx* c, int on) {
  if (!c) return ELLC_ERR_BAD_ARG;
  ELLC_ENTER(c);
  c->hinv_cache = on != 0;
  return ELLC_OK;
}
ellc_status ellc_debug_persist_counters(ellc_ctx* c, long long* resident_launches, long long* abandoned_launches, long long* rejoined_blocks) {
  if (!c) return ELLC_ERR_BAD_ARG;
  ELLC_ENTER(c);
  if (resident_launches) *resident_launches = c->persist_launches;
  if (abandoned_launches) *abandoned_launches = c->persist_abandoned;
  if (rejoined_blocks) {   // (device-wide, since the library was loaded)
    unsigned long long v = 0;
    ELLC_HIP(c, hipDeviceSynchronize());
    ELLC_HIP(c, hipMemcpyFromSymbol(&v, HIP_SYMBOL(ellc::g_persist_adoptions), sizeof(v)));
    *rejoined_blocks = (long long)v;
  }
  return ELLC_OK;
}
#endif   // ELLC_DIAG_ABI

ellc_status ellc_ctx_set_dense_maps(ellc_ctx* c, int mode) {
  ELLC_ENTER_BATCH(c);
  if (!c || mode < 0 || mode > 1) return fail(c, ELLC_ERR_BAD_ARG, "ellc_ctx_set_dense_maps: mode 0 (automatic) or 1 (always the lists)");
  if (c->open_set >= 0) {   // a group still waiting for batches to join was staged under the old value: it runs as it is
    const ellc_status s = launch_group(c, c->open_set);
    if (s != ELLC_OK) return s;
  }
  c->dense_maps_off = mode == 1;
  return ELLC_OK;
}

ellc_status ellc_ctx_set_grid_batch(ellc_ctx* c, int n) {
  ELLC_ENTER_BATCH(c);
  if (!c || n < 0 || n > 65536) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (c->open_set >= 0) {   // a group still waiting for batches to join was staged under the old value: it runs as it is
    const ellc_status s = launch_group(c, c->open_set);
    if (s != ELLC_OK) return s;
  }
  c->cfg.grid_batch = n;
  return ELLC_OK;
}

ellc_status ellc_frame_upload(ellc_ctx* c, int slot, const uint8_t* image) {
  ELLC_ENTER(c);
  if (!c || !image || !slot_ok(slot, c->cfg.max_frames)) return fail(c, ELLC_ERR_BAD_ARG, "ellc_frame_upload: bad argument");
  uint8_t* img[ELLC_MAX_LEVELS];
  for (int l = 0; l < c->L; l++) img[l] = c->fr_tab_h[(size_t)l * c->cfg.max_frames + slot].img;
  // (a stream of its own only in contexts that keep one batch in flight — the tracking context: a process gets three hardware
  // queues that run concurrently, and a context with batches in flight needs them for its batch streams; r03: with the fourth
  // stream the pipeline of sixteen batches fell from 7.7 to 6.3 M iterations/s)
  const bool own_stream = c->cfg.concurrent_batches <= 1 && c->coalesce <= 1;
  ellc_status s = upload_pyramid(c, img, image, own_stream ? slot : -1);
  if (s != ELLC_OK) return s;
  c->fr_has_image[slot] = 1;
  c->fr_maxgrad_valid[slot] = 0;
  return ELLC_OK;
}

ellc_status ellc_keyframe_upload(ellc_ctx* c, int slot, const uint8_t* image) {
  ELLC_ENTER(c);
  if (!c || !image || !slot_ok(slot, c->cfg.max_keyframes)) return fail(c, ELLC_ERR_BAD_ARG, "ellc_keyframe_upload: bad argument");
  uint8_t* img[ELLC_MAX_LEVELS];
  for (int l = 0; l < c->L; l++) img[l] = c->kf_tab_h[(size_t)l * c->cfg.max_keyframes + slot].img;
  invalidate_records(c, slot);   // before the first write: a failure half-way must not leave a valid tag on overwritten planes
  ellc_status s = upload_pyramid(c, img, image);
  if (s != ELLC_OK) return s;
  c->kf_has_image[slot] = 1;
  c->kf_dense[slot] = 0;
  c->kf_has_depth[slot] = 0;         // a fresh frame has no depth yet (as ellc_keyframe_from_frame): set_depth / update_depth_image follow
  for (int l = 0; l < c->L; l++) {   // frame::frame zeroes weight_pyramid / numWeightsAdded (Frame.cpp:114-122)
    ELLC_HIP(c, hipMemsetAsync(c->kf_tab_h[(size_t)l * c->cfg.max_keyframes + slot].weight, 0, (size_t)c->geom_h[l].n * 4, c->stream));
    c->kf_num_weights[slot][l] = 0;
  }
  return build_maxgrad(c, true, slot);
}

ellc_status ellc_keyframe_from_frame(ellc_ctx* c, int kf_slot, int frame_slot) {
  ELLC_ENTER(c);
  if (!c || !slot_ok(kf_slot, c->cfg.max_keyframes) || !slot_ok(frame_slot, c->cfg.max_frames)) return fail(c, ELLC_ERR_BAD_ARG, "bad slot");
  if (!c->fr_has_image[frame_slot]) return fail(c, ELLC_ERR_NOT_READY, "frame slot empty");
  invalidate_records(c, kf_slot);
  for (int l = 0; l < c->L; l++) {
    const LevelGeom& g = c->geom_h[l];
    ELLC_HIP(c, hipMemcpyAsync(c->kf_tab_h[(size_t)l * c->cfg.max_keyframes + kf_slot].img, c->fr_tab_h[(size_t)l * c->cfg.max_frames + frame_slot].img,
                               (size_t)g.sw * g.sh, hipMemcpyDeviceToDevice, c->stream));
    ELLC_HIP(c, hipMemsetAsync(c->kf_tab_h[(size_t)l * c->cfg.max_keyframes + kf_slot].weight, 0, (size_t)g.n * 4, c->stream));
    c->kf_num_weights[kf_slot][l] = 0;
  }
  c->kf_has_image[kf_slot] = 1;
  c->kf_has_depth[kf_slot] = 0;
  c->kf_dense[kf_slot] = 0;
  const ellc_status ms = mark_frame_use(c, frame_slot);
  if (ms != ELLC_OK) return ms;
  return build_maxgrad(c, true, kf_slot);
}

ellc_status ellc_get_image_level(ellc_ctx* c, int is_kf, int slot, int level, uint8_t* out, int* stored_w, int* stored_h, int* cols, int* rows) {
  ELLC_ENTER(c);
  if (!c || level < 0 || level >= c->L || !slot_ok(slot, is_kf ? c->cfg.max_keyframes : c->cfg.max_frames)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (!(is_kf ? c->kf_has_image[slot] : c->fr_has_image[slot])) return fail(c, ELLC_ERR_NOT_READY, "slot empty");
  const LevelGeom& g = c->geom_h[level];
  if (stored_w) *stored_w = g.sw;
  if (stored_h) *stored_h = g.sh;
  if (cols) *cols = g.cols;
  if (rows) *rows = g.rows;
  if (out) {
    const uint8_t* src = is_kf ? c->kf_tab_h[(size_t)level * c->cfg.max_keyframes + slot].img : c->fr_tab_h[(size_t)level * c->cfg.max_frames + slot].img;
    ELLC_HIP(c, hipMemcpyAsync(out, src, (size_t)g.sw * g.sh, hipMemcpyDeviceToHost, c->stream));
    ELLC_HIP(c, hipStreamSynchronize(c->stream));
  }
  return ELLC_OK;
}

ellc_status ellc_get_gradient(ellc_ctx* c, int is_kf, int slot, int level, float* gx, float* gy) {
  ELLC_ENTER(c);
  if (!c || !gx || !gy || level < 0 || level >= c->L || !slot_ok(slot, is_kf ? c->cfg.max_keyframes : c->cfg.max_frames)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (!(is_kf ? c->kf_has_image[slot] : c->fr_has_image[slot])) return fail(c, ELLC_ERR_NOT_READY, "slot empty");
  const LevelGeom& g = c->geom_h[level];
  const uint8_t* src = is_kf ? c->kf_tab_h[(size_t)level * c->cfg.max_keyframes + slot].img : c->fr_tab_h[(size_t)level * c->cfg.max_frames + slot].img;
  dim3 blk(32, 8);
  hipLaunchKernelGGL(gradient_planes, grid2d(g.cols, g.rows, blk), blk, 0, c->stream, src, g.sw, g.cols, g.rows, c->scratch_a, c->scratch_b);
  ELLC_HIP(c, hipGetLastError());
  ELLC_HIP(c, hipMemcpyAsync(gx, c->scratch_a, (size_t)g.n * 4, hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(gy, c->scratch_b, (size_t)g.n * 4, hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  return ELLC_OK;
}

ellc_status ellc_get_max_gradient(ellc_ctx* c, int is_kf, int slot, float* out, int* n_substantial) {
  ELLC_ENTER(c);
  if (!c || !slot_ok(slot, is_kf ? c->cfg.max_keyframes : c->cfg.max_frames)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (!(is_kf ? c->kf_has_image[slot] : c->fr_has_image[slot])) return fail(c, ELLC_ERR_NOT_READY, "slot empty");
  if (!(is_kf ? c->kf_maxgrad_valid[slot] : c->fr_maxgrad_valid[slot])) {
    ellc_status s = build_maxgrad(c, is_kf != 0, slot);
    if (s != ELLC_OK) return s;
  }
  const size_t n0 = (size_t)c->cfg.width * c->cfg.height;
  if (out) ELLC_HIP(c, hipMemcpyAsync(out, is_kf ? c->kf_maxgrad[slot] : c->fr_maxgrad[slot], n0 * 4, hipMemcpyDeviceToHost, c->stream));
  if (n_substantial) ELLC_HIP(c, hipMemcpyAsync(n_substantial, is_kf ? c->kf_maxgrad_count[slot] : c->fr_maxgrad_count[slot], 4, hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  return ELLC_OK;
}

// ---- keyframe depth / variance / weights -----------------------------------------------------------
ellc_status ellc_keyframe_set_depth(ellc_ctx* c, int slot, const float* depth0, const float* var0) {
  ELLC_ENTER(c);
  if (!c || !depth0 || !var0 || !slot_ok(slot, c->cfg.max_keyframes)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  const KfLevelDev& k = c->kf_tab_h[slot];
  const size_t n0 = (size_t)c->geom_h[0].n;
  invalidate_records(c, slot);
  ELLC_HIP(c, hipMemcpyAsync(k.depth, depth0, n0 * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(k.var, var0, n0 * 4, hipMemcpyHostToDevice, c->stream));
  ellc_status s = build_depth_pyramid(c, slot);
  if (s != ELLC_OK) return s;
  // dense hint (see gn_fca_dense): counted here, on the host's copy, while the upload is in flight — a map uploaded (nearly) full is
  // aligned without compact lists; maps the depth stages export are semi-dense by construction and never are
  size_t nvalid = 0;
  for (size_t i = 0; i < n0; i++) nvalid += depth0[i] > 0.0f ? 1 : 0;
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  c->kf_has_depth[slot] = 1;
  c->kf_dense[slot] = (nvalid * 10 >= n0 * 9) ? 1 : 0;
  if (c->kf_dense[slot])   // the reciprocal planes gn_fca_dense4 reads (KfLevelDev::idepth)
    for (int l = 0; l < c->L; l++) {
      const KfLevelDev& kl = c->kf_tab_h[(size_t)l * c->cfg.max_keyframes + slot];
      const int n = c->geom_h[l].n;
      hipLaunchKernelGGL(idepth_plane, dim3((n + 255) / 256), dim3(256), 0, c->stream, kl.depth, kl.idepth, n);
      if (kl.invz) hipLaunchKernelGGL(invz_plane, dim3((n + 255) / 256), dim3(256), 0, c->stream, kl.depth, kl.invz, n);
    }
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

ellc_status ellc_keyframe_set_depth_level(ellc_ctx* c, int slot, int level, const float* depth, const float* var) {
  ELLC_ENTER(c);
  if (!c || !depth || !var || level < 0 || level >= c->L || !slot_ok(slot, c->cfg.max_keyframes)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  const KfLevelDev& k = c->kf_tab_h[(size_t)level * c->cfg.max_keyframes + slot];
  invalidate_records(c, slot);
  ELLC_HIP(c, hipMemcpyAsync(k.depth, depth, (size_t)c->geom_h[level].n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipMemcpyAsync(k.var, var, (size_t)c->geom_h[level].n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  c->kf_has_depth[slot] = 1;
  c->kf_dense[slot] = 0;   // (a level written by itself: no hint)
  return ELLC_OK;
}

ellc_status ellc_keyframe_get_depth_level(ellc_ctx* c, int slot, int level, float* depth, float* var) {
  ELLC_ENTER(c);
  if (!c || level < 0 || level >= c->L || !slot_ok(slot, c->cfg.max_keyframes)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  const KfLevelDev& k = c->kf_tab_h[(size_t)level * c->cfg.max_keyframes + slot];
  if (depth) ELLC_HIP(c, hipMemcpyAsync(depth, k.depth, (size_t)c->geom_h[level].n * 4, hipMemcpyDeviceToHost, c->stream));
  if (var) ELLC_HIP(c, hipMemcpyAsync(var, k.var, (size_t)c->geom_h[level].n * 4, hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  return ELLC_OK;
}

ellc_status ellc_keyframe_set_weights(ellc_ctx* c, int slot, int level, const float* w, int num_added) {
  ELLC_ENTER(c);
  if (!c || !w || level < 0 || level >= c->L || !slot_ok(slot, c->cfg.max_keyframes)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  const KfLevelDev& k = c->kf_tab_h[(size_t)level * c->cfg.max_keyframes + slot];
  invalidate_records(c, slot);
  ELLC_HIP(c, hipMemcpyAsync(k.weight, w, (size_t)c->geom_h[level].n * 4, hipMemcpyHostToDevice, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  c->kf_num_weights[slot][level] = num_added;
  return ELLC_OK;
}

ellc_status ellc_keyframe_get_weights(ellc_ctx* c, int slot, int level, float* w, int* num_added) {
  ELLC_ENTER(c);
  if (!c || level < 0 || level >= c->L || !slot_ok(slot, c->cfg.max_keyframes)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  const KfLevelDev& k = c->kf_tab_h[(size_t)level * c->cfg.max_keyframes + slot];
  if (w) {
    ELLC_HIP(c, hipMemcpyAsync(w, k.weight, (size_t)c->geom_h[level].n * 4, hipMemcpyDeviceToHost, c->stream));
    ELLC_HIP(c, hipStreamSynchronize(c->stream));
  }
  if (num_added) *num_added = c->kf_num_weights[slot][level];
  return ELLC_OK;
}

ellc_status ellc_keyframe_finalise_weights(ellc_ctx* c, int slot) {
  ELLC_ENTER(c);
  if (!c || !slot_ok(slot, c->cfg.max_keyframes)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  invalidate_records(c, slot);
  for (int l = c->L - 1; l >= 0; l--) {
    const int na = c->kf_num_weights[slot][l];
    if (na > 0) {
      const int n = c->geom_h[l].n;
      const float s = (float)(1.0 / (double)na);
      hipLaunchKernelGGL(scale_plane, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->kf_tab_h[(size_t)l * c->cfg.max_keyframes + slot].weight, n, s);
    }
  }
  ELLC_HIP(c, hipGetLastError());
  return ELLC_OK;
}

// ---- loop-closure support ----------------------------------------------------------------------------
ellc_status ellc_histogram(ellc_ctx* c, int is_kf, int slot, float* hist256) {
  ELLC_ENTER(c);
  if (!c || !hist256 || !slot_ok(slot, is_kf ? c->cfg.max_keyframes : c->cfg.max_frames)) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (!(is_kf ? c->kf_has_image[slot] : c->fr_has_image[slot])) return fail(c, ELLC_ERR_NOT_READY, "slot empty");
  const LevelGeom& g = c->geom_h[0];
  const uint8_t* img = is_kf ? c->kf_tab_h[slot].img : c->fr_tab_h[slot].img;
  unsigned* bins = (unsigned*)c->scratch_a;
  ELLC_HIP(c, hipMemsetAsync(bins, 0, 256 * sizeof(unsigned), c->stream));
  hipLaunchKernelGGL(hist256_u8, dim3(128), dim3(256), 0, c->stream, img, g.sw, g.cols, g.rows, bins);
  ELLC_HIP(c, hipGetLastError());
  unsigned counts[256];
  ELLC_HIP(c, hipMemcpyAsync(counts, bins, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  float sum = 0;   // GlobalOptimize.cpp:79-89: f32 sum of the (integer-valued) bin counts, then bin /= sum
  for (int i = 0; i < 256; i++) sum += (float)counts[i];
  for (int i = 0; i < 256; i++) hist256[i] = (float)counts[i] / sum;
  return ELLC_OK;
}

double ellc_kl_divergence(const float* p, const float* q, int n) {
  double result = 0;
  for (int j = 0; j < n; j++) {
    const double a = p[j];
    double b = q[j];
    if (std::fabs(a) <= 2.220446049250313e-16) continue;
    if (std::fabs(b) <= 2.220446049250313e-16) b = 1e-10;
    result += a * std::log(a / b);
  }
  return result;
}

// device-to-device copy of a slot's planes, enqueued on dc's stream; sc == dc, or another context on the same device with the
// same geometry (the caller has ordered dc's stream behind sc's pending work)
static ellc_status copy_slot_planes(ellc_ctx* dc, int dst_is_kf, int dst, ellc_ctx* sc, int src_is_kf, int src) {
  ellc_ctx* c = dc;
  const int MK = dc->cfg.max_keyframes, MF = dc->cfg.max_frames, SK = sc->cfg.max_keyframes, SF = sc->cfg.max_frames;
  if (dst_is_kf) invalidate_records(dc, dst);
  for (int l = 0; l < dc->L; l++) {
    const LevelGeom& g = dc->geom_h[l];
    const uint8_t* s_img = src_is_kf ? sc->kf_tab_h[(size_t)l * SK + src].img : sc->fr_tab_h[(size_t)l * SF + src].img;
    uint8_t* d_img = dst_is_kf ? dc->kf_tab_h[(size_t)l * MK + dst].img : dc->fr_tab_h[(size_t)l * MF + dst].img;
    ELLC_HIP(c, hipMemcpyAsync(d_img, s_img, (size_t)g.sw * g.sh, hipMemcpyDeviceToDevice, dc->stream));
    if (dst_is_kf) {
      KfLevelDev& d = dc->kf_tab_h[(size_t)l * MK + dst];
      if (src_is_kf) {
        const KfLevelDev& k = sc->kf_tab_h[(size_t)l * SK + src];
        ELLC_HIP(c, hipMemcpyAsync(d.depth, k.depth, (size_t)g.n * 4, hipMemcpyDeviceToDevice, dc->stream));
        ELLC_HIP(c, hipMemcpyAsync(d.var, k.var, (size_t)g.n * 4, hipMemcpyDeviceToDevice, dc->stream));
        if (sc->kf_dense[src]) ELLC_HIP(c, hipMemcpyAsync(d.idepth, k.idepth, (size_t)g.n * 4, hipMemcpyDeviceToDevice, dc->stream));   // (the dense hint travels with the slot)
        if (sc->kf_dense[src] && d.invz) {
          if (k.invz) ELLC_HIP(c, hipMemcpyAsync(d.invz, k.invz, (size_t)g.n * 8, hipMemcpyDeviceToDevice, dc->stream));
          else hipLaunchKernelGGL(invz_plane, dim3((g.n + 255) / 256), dim3(256), 0, dc->stream, d.depth, d.invz, g.n);   // (a tolerance-mode source context keeps none)
        }
        ELLC_HIP(c, hipMemcpyAsync(d.weight, k.weight, (size_t)g.n * 4, hipMemcpyDeviceToDevice, dc->stream));
        dc->kf_num_weights[dst][l] = sc->kf_num_weights[src][l];
      } else {
        ELLC_HIP(c, hipMemsetAsync(d.weight, 0, (size_t)g.n * 4, dc->stream));
        dc->kf_num_weights[dst][l] = 0;
      }
    }
  }
  if (dst_is_kf) {
    dc->kf_has_image[dst] = 1;
    dc->kf_has_depth[dst] = src_is_kf ? sc->kf_has_depth[src] : 0;
    dc->kf_dense[dst] = src_is_kf ? sc->kf_dense[src] : 0;
    if (src_is_kf && sc->kf_maxgrad_valid[src]) {
      const size_t n0 = (size_t)dc->cfg.width * dc->cfg.height;
      ELLC_HIP(c, hipMemcpyAsync(dc->kf_maxgrad[dst], sc->kf_maxgrad[src], n0 * 4, hipMemcpyDeviceToDevice, dc->stream));
      ELLC_HIP(c, hipMemcpyAsync(dc->kf_maxgrad_count[dst], sc->kf_maxgrad_count[src], 4, hipMemcpyDeviceToDevice, dc->stream));
      dc->kf_maxgrad_valid[dst] = 1;
    } else {
      return build_maxgrad(dc, true, dst);
    }
  } else {
    dc->fr_has_image[dst] = 1;
    dc->fr_maxgrad_valid[dst] = 0;
  }
  return ELLC_OK;
}
static ellc_status mark_copy_uses(ellc_ctx* dc, int dst_is_kf, int dst, ellc_ctx* sc, int src_is_kf, int src) {
  ellc_status s = ELLC_OK;
  if (!src_is_kf) s = mark_frame_use(sc, src);
  if (s == ELLC_OK && !dst_is_kf && dc == sc) s = mark_frame_use(dc, dst);
  return s;
}

ellc_status ellc_copy_slot(ellc_ctx* c, int dst_is_kf, int dst, int src_is_kf, int src) {
  ELLC_ENTER(c);
  if (!c || !slot_ok(dst, dst_is_kf ? c->cfg.max_keyframes : c->cfg.max_frames) || !slot_ok(src, src_is_kf ? c->cfg.max_keyframes : c->cfg.max_frames))
    return fail(c, ELLC_ERR_BAD_ARG, "bad slot");
  if (!(src_is_kf ? c->kf_has_image[src] : c->fr_has_image[src])) return fail(c, ELLC_ERR_NOT_READY, "source slot empty");
  if (dst_is_kf == src_is_kf && dst == src) return ELLC_OK;
  // a frame slot written here: behind an upload of its own that may still be running on the upload stream? (uploads make the
  // main stream wait at once, so plain stream order covers it); read or written here: later uploads wait for this copy
  const ellc_status st = copy_slot_planes(c, dst_is_kf, dst, c, src_is_kf, src);
  return st != ELLC_OK ? st : mark_copy_uses(c, dst_is_kf, dst, c, src_is_kf, src);
}

// The same between two contexts of one device (same width / height / levels): the reference's loop-closure thread works on deep
// copies of the finished keyframe and its depth map (GlobalOptimize.cpp:185-186) while tracking goes on; here the copy goes
// into the loop-closure context's ring slot. Device-side ordering only (events): dst's stream waits for what src has enqueued,
// and src's later work waits for the copy. The caller serialises this call with every other call on EITHER context.
ellc_status ellc_copy_slot_across(ellc_ctx* dc, int dst_is_kf, int dst, ellc_ctx* sc, int src_is_kf, int src) {
  if (!dc || !sc) return ELLC_ERR_BAD_ARG;
  if (dc == sc) return ellc_copy_slot(dc, dst_is_kf, dst, src_is_kf, src);
  if (dc->cfg.device != sc->cfg.device || dc->cfg.width != sc->cfg.width || dc->cfg.height != sc->cfg.height || dc->L != sc->L)
    return fail(dc, ELLC_ERR_BAD_ARG, "ellc_copy_slot_across: the contexts differ in device or geometry");
  if (!slot_ok(dst, dst_is_kf ? dc->cfg.max_keyframes : dc->cfg.max_frames) || !slot_ok(src, src_is_kf ? sc->cfg.max_keyframes : sc->cfg.max_frames))
    return fail(dc, ELLC_ERR_BAD_ARG, "bad slot");
  if (!(src_is_kf ? sc->kf_has_image[src] : sc->fr_has_image[src])) return fail(dc, ELLC_ERR_NOT_READY, "source slot empty");
  {
    ellc_ctx* c = sc;   // (ELLC_ENTER / ELLC_HIP report into the context named c)
    ELLC_ENTER(c);
    if (!sc->ev_xfer) ELLC_HIP(c, hipEventCreateWithFlags(&sc->ev_xfer, hipEventDisableTiming));
    ELLC_HIP(c, hipEventRecord(sc->ev_xfer, sc->stream));
  }
  ellc_ctx* c = dc;
  ELLC_ENTER(c);
  if (!dc->ev_xfer) ELLC_HIP(c, hipEventCreateWithFlags(&dc->ev_xfer, hipEventDisableTiming));
  ELLC_HIP(c, hipStreamWaitEvent(dc->stream, sc->ev_xfer, 0));
  const ellc_status st = copy_slot_planes(dc, dst_is_kf, dst, sc, src_is_kf, src);
  // whatever was enqueued: the source must not be overwritten before it has been read (the source context's main stream waits
  // for the copy below, and with it every later upload there: mark_frame_use on sc is not needed)
  ELLC_HIP(c, hipEventRecord(dc->ev_xfer, dc->stream));
  if (hipStreamWaitEvent(sc->stream, dc->ev_xfer, 0) != hipSuccess) return fail(sc, ELLC_ERR_HIP, "ellc_copy_slot_across: cannot order the source stream");
  sc->main_dirty = true;
  return st;
}

// ---- alignment -------------------------------------------------------------------------------------
// prep + init + the whole level/iteration schedule; captured once per (B, unique keyframes, mode, save_weights)
// into a hipGraph and replayed afterwards (the launches are too short to be issued one by one from the host)
// the record set (PrepArgs::need) a schedule reads
static int need_of(const ellc_ctx* c, int mode) {
  return mode == ELLC_MODE_ICA ? (c->use_fused ? (c->fast ? 20 : 4) : 1) : (c->fast ? 8 : 2);
}

// nu: keyframe slots whose compact lists are (re)built — all the unique slots of the batch, or with cfg.cache_records only
// those whose lists are stale (possibly none)
static ellc_status enqueue_align_body(ellc_ctx* c, int B, int nu, int mode, int save_weights) {
  enqueue_stage_in(c, B);   // also initialises the B alignment states
  // mask / count per level (updationOnPyrChange, ImageFunc.cpp:158) and the pose-independent per-pixel records. (Folding the
  // staging into the count launch — its tile blocks then read their keyframe slot from the pinned record, one PCIe round trip
  // per block — was measured in r02: the count launch went from 10 to 29 us at 32 keyframes; the separate 8 us launch stays.)
  const int need = c->cur_need ? c->cur_need : need_of(c, mode);   // (cur_need: launch_group's choice — 16 instead of 20 when every rebuilt slot's H^-1 is current)
  ellc_status s = ELLC_OK;
  if (nu > 0) {
    s = run_prep(c, nu, need);
    if (s != ELLC_OK) return s;
    if (need & 4) enqueue_ica_hinv(c, nu);
  }
  s = enqueue_schedule(c, B, mode, save_weights);
  if (s != ELLC_OK) return s;
  if (!c->use_fused)   // the fused schedules export from their finish kernel
    hipLaunchKernelGGL(gn_export_results, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->state_d, c->result_dev_alias, B);
  return ELLC_OK;
}

// may a batch of B alignments run the list-free schedule (gn_fca_dense)? The tolerance-mode FCA schedule in its level-bound form,
// without saved weights (they are kept per list entry); launch_group adds: every keyframe slot carries the dense hint
static bool runs_dense(const ellc_ctx* c, int mode, int B, int save_weights) {
  // (r06: the exact mode too — gn_fca_dense_x over the planes and the slot's 1 / Z plane in double)
  return !c->dense_maps_off && c->use_fused && c->pipe && mode == ELLC_MODE_FCA && !save_weights && !schedule_is_adaptive(c, mode, B);
}


// Enqueues the launch sequence of one batch on c->stream — replayed from a hipGraph captured on first use, keyed by
// (B, unique keyframes, mode, save_weights, batch set, part). continuation: the rest of a state-driven schedule whose first
// graph ended before every alignment had (enqueue_schedule_adaptive), for the batch set selected in the context.
static ellc_status launch_align_graph(ellc_ctx* c, int B, int nu, int mode, int save_weights, int set, bool continuation) {
  auto body = [&]() -> ellc_status {
    if (continuation) return enqueue_schedule_adaptive(c, B, save_weights, schedule_total_iters(c) - c->cur_adaptive_first, true);
    return enqueue_align_body(c, B, nu, mode, save_weights);
  };
  // The state-driven schedule (the tracking call: one or two alignments) is launched kernel by kernel: its launches are ~6 us
  // each and dependent, so the host stays ahead of the device without a graph, the depth stages that follow start without the
  // ~14 us a graph's end costs the next launch on the stream (r03 timeline: tracked frame 0.252 -> 0.245 ms), and a first graph
  // whose length follows the previous frame's iteration count (adaptive_hint) needs no re-capture when that count changes.
  if (launches_directly(c, mode, B)) {
    c->direct_launch = !continuation;   // (a continuation has no staging)
    c->direct_nu = nu;
    const ellc_status s = body();
    c->direct_launch = false;
    return s;
  }
  // (cur_adaptive_first: launches of the first graph of a state-driven schedule; it varies with the context's hint)
  const int first = schedule_is_adaptive(c, mode, B) ? c->cur_adaptive_first : 0;
  const auto key = std::make_tuple(B, continuation ? 0 : nu, mode,
                                   (save_weights ? 1 : 0) | (continuation ? 2 : 0) | (c->track_call ? 4 : 0) | (c->cur_pollable ? 8 : 0) | (first << 4) | (c->cfg.grid_batch << 12) | (c->cur_dense ? (1 << 29) : 0) | (c->cur_need == 16 ? (1 << 28) : 0), set);
  auto it = c->graphs.find(key);
  if (it == c->graphs.end()) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ELLC_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    const ellc_status s = body();
    hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (s != ELLC_OK || e != hipSuccess) {
      if (graph) (void)hipGraphDestroy(graph);
      if (s != ELLC_OK) return s;
      return fail(c, ELLC_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(c, ELLC_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    it = c->graphs.emplace(key, exec).first;
  }
  ELLC_HIP(c, hipGraphLaunch(it->second, c->stream));
  return ELLC_OK;
}

// swaps the stream every launch helper uses (c->stream) for the duration of one enqueue
struct StreamScope {
  ellc_ctx* c;
  hipStream_t saved;
  StreamScope(ellc_ctx* c_, hipStream_t s) : c(c_), saved(c_->stream) { c->stream = s; }
  ~StreamScope() { c->stream = saved; }
};

// Waits until the launched group's result records are there. A batch (of at most 64 alignments) with nothing else in flight (the tracking
// call, a single alignment): the host polls the pad words of the records the finish kernel writes into pinned
// memory — a microsecond after the kernel's store instead of the event's completion path — and falls back to the event after
// 2 ms (a failed launch never writes them). Everything the context does next is on the same stream, behind whatever of this
// batch is still running (saved weights, the depth stages of a tracked frame).
// (decided when the group is launched — its finish kernel then orders its stores for the host, FusedArgs::host_polls — and again
// when it is waited for: nothing else may have been enqueued in between)
static bool polls_results(const ellc_ctx* c, int B, int stream_idx) {
  // the only batch in flight, on the main stream: whatever the context launches next — the next group takes the lowest free set
  // and stream, i.e. these — is ordered behind this batch's trailing kernels by the stream itself
  return c->poll_results && c->use_fused && B <= 64 && stream_idx == 0 && c->n_inflight <= 1;
}
static hipError_t wait_batch_results(ellc_ctx* c, ellc_ctx::BatchSet& bs) {
  if (bs.pollable && polls_results(c, bs.B, bs.stream_idx)) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; spin++) {
      bool all = true;
      for (int b = 0; b < bs.B; b++) all = all && (*(volatile const int*)&bs.result_h[b].pad != -1);
      if (all) {
        std::atomic_thread_fence(std::memory_order_acquire);
        // The batch's trailing kernels (saved weights, the depth stages of a tracked frame) may still be running on the main stream,
        // and once the set is freed nothing else remembers them: a group placed on ANOTHER stream next (it may rebuild lists from
        // the weight planes the saved-weights kernel is adding to) has to be ordered behind the main stream as it is now. The
        // event wait this poll replaces gave that order by itself (r03 advisor finding).
        c->main_dirty = true;
        c->counters[ELLC_CTR_POLLED]++;
        return hipSuccess;
      }
      if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(c->poll_timeout_us)) break;
    }
    c->counters[ELLC_CTR_POLL_TIMEOUT]++;
  }
  c->counters[ELLC_CTR_EVENT_WAIT]++;
  return hipEventSynchronize(bs.done);
}

// Waits for a launched group and, when it ran the state-driven schedule and one of its alignments had not ended when the
// first graph did (result pad = 1, gn_fused_finish), replays the continuation graph on the group's stream and waits again.
static ellc_status resolve_batch(ellc_ctx* c, int set) {
  ellc_ctx::BatchSet& bs = c->batch_set[set];
  if (bs.resolved) return ELLC_OK;
  bs.resolved = true;
  hipError_t ev = wait_batch_results(c, bs);   // the last kernel wrote bs.result_h (pinned, zero-copy)
  if (ev != hipSuccess) {
    std::fill(c->kf_rec_tag.begin(), c->kf_rec_tag.end(), 0);   // whatever was being built cannot be trusted
    return fail(c, ELLC_ERR_HIP, std::string("the batch failed on the device: ") + hipGetErrorString(ev));
  }
  if (!bs.adaptive) return ELLC_OK;
  bool unfinished = false;
  for (int b = 0; b < bs.B; b++) unfinished = unfinished || (bs.result_h[b].pad == 1);
  if (!unfinished) return ELLC_OK;
  if (bs.resident) {   // a resident launch that was abandoned: the next calls do not try again at once (may_run_resident)
    c->persist_abandoned++;
    if (c->persist_spin_limit != 0u && c->persist_delay_polls >= 0) c->persist_backoff = 16;   // (not for the test hooks that abandon launches on purpose)
  }
  const int selected = c->cur_set;
  select_batch_set(c, set);
  c->cur_adaptive_first = bs.adaptive_first;
  c->cur_resident = false;   // (a continuation is launches)
  ellc_status s = ELLC_OK;
  {
    StreamScope scope(c, c->batch_stream[bs.stream_idx]);
    s = launch_align_graph(c, bs.B, 0, bs.mode, bs.save_weights, set, true);
    if (s == ELLC_OK && hipEventRecord(bs.done, c->stream) != hipSuccess) s = fail(c, ELLC_ERR_HIP, "hipEventRecord failed");
  }
  select_batch_set(c, selected);
  if (s != ELLC_OK) return s;
  c->counters[ELLC_CTR_CONTINUATION]++;
  ev = hipEventSynchronize(bs.done);
  if (ev != hipSuccess) return fail(c, ELLC_ERR_HIP, std::string("the batch failed on the device: ") + hipGetErrorString(ev));
  return ELLC_OK;
}

// Launches the group staged in `set`: its batches lie side by side (batch j = alignments [j * max_batch, ...)), the launch
// sequence covers all of them. The group takes the lowest batch stream no unfetched group occupies (a caller with one batch at
// a time stays on the main stream), else the stream of the group launched longest ago.
static ellc_status launch_group(ellc_ctx* c, int set) {
  ellc_ctx::BatchSet& bs = c->batch_set[set];
  if (c->open_set == set) c->open_set = -1;
  if (bs.launched || bs.fill < 1) return ELLC_OK;
  const int MB = c->cfg.max_batch;
  const int k = bs.fill;
  const int B = (k == 1) ? bs.slice_B[0] : k * MB;
  select_batch_set(c, set);
  // unique keyframe slots of the group (the compaction runs once per slot)
  bs.kf_slots.clear();
  for (int b = 0; b < B; b++) {
    bool seen = false;
    for (int v : bs.kf_slots) seen = seen || (v == c->kf_slot_h[b]);
    if (!seen) bs.kf_slots.push_back(c->kf_slot_h[b]);
  }
  // the slots whose lists this launch (re)builds: all of them, or with cfg.cache_records those whose lists are stale
  const int need = need_of(c, bs.mode);
  const bool saves = bs.save_weights && bs.mode == ELLC_MODE_FCA;
  bs.built_slots.clear();
  for (int v : bs.kf_slots)
    if (!(c->cache_records || c->kf_rec_eager[v]) || c->kf_rec_tag[v] != need) bs.built_slots.push_back(v);   // (kf_rec_eager: built behind the map's export)
  // dense maps (every keyframe of the launch carries the hint): the list-free schedule — no slot's lists are built or read
  bool dense = runs_dense(c, bs.mode, B, bs.save_weights);
  for (int v : bs.kf_slots) dense = dense && c->kf_dense[v];
  if (dense) bs.built_slots.clear();
  const int nu = (int)bs.built_slots.size();
  for (int u = 0; u < nu; u++) c->uniq_slot_h[u] = bs.built_slots[u];
  // constant-weight path, tolerance mode: when the H^-1 of every slot that is rebuilt are current (kf_hinv_ok), the compaction
  // builds the records only (prep_scatter<16>: no exact template row, no sums, no ica_hinv)
  int need_run = need;
  if (need == 20 && c->hinv_cache && nu > 0) {
    bool all_ok = true;
    for (int v : bs.built_slots) all_ok = all_ok && c->kf_hinv_ok[v];
    if (all_ok) need_run = 16;
  }
  // stream
  bool busy[ellc_ctx::STREAMS] = {};
  for (int p = 0; p < ellc_ctx::SETS; p++)
    if (p != set && c->batch_set[p].launched) busy[c->batch_set[p].stream_idx] = true;
  int si = 0;
  while (si < ellc_ctx::STREAMS && busy[si]) si++;
  if (si == ellc_ctx::STREAMS) {   // all three carry a group: behind the one launched longest ago (the head of the queue)
    si = c->n_inflight > 0 ? c->batch_set[c->inflight[0] / ellc_ctx::MAX_COALESCE].stream_idx : 0;
  }
  if (si > 0 && !c->batch_stream[si]) ELLC_HIP(c, hipStreamCreate(&c->batch_stream[si]));   // created on first use
  hipStream_t run_stream = c->batch_stream[si];
  // A group on another stream runs after everything the caller has put on the main stream through the other entry
  // points (uploads, depth stages), but not after the groups that run there: the mark is recorded before them.
  // (a context that has only ever had one batch in flight has no other stream and never records the mark)
  if (c->main_dirty && (si > 0 || c->batch_stream[1])) {
    ELLC_HIP(c, hipEventRecord(c->ev_main, c->stream));
    c->main_dirty = false;
    c->main_mark++;
  }
  if (si > 0 && c->stream_waited_mark[si] != c->main_mark) {
    ELLC_HIP(c, hipStreamWaitEvent(run_stream, c->ev_main, 0));
    c->stream_waited_mark[si] = c->main_mark;
  }
  // after the groups in flight with which it conflicts on a keyframe slot: one of the two writes what the other reads — the
  // compact lists and H^-1 (a rebuild) or the saved weights, all of which live in the slot. (Without cfg.cache_records every
  // group rebuilds every slot it uses: any shared slot is a conflict.)
  auto hits = [](const std::vector<int>& a, const std::vector<int>& b) {
    for (int u : a)
      for (int v : b)
        if (u == v) return true;
    return false;
  };
  for (int p = 0; p < ellc_ctx::SETS; p++) {
    ellc_ctx::BatchSet& other = c->batch_set[p];
    if (p == set || !other.launched) continue;
    const bool other_saves = other.save_weights && other.mode == ELLC_MODE_FCA;
    const bool shared = hits(bs.built_slots, other.kf_slots) || hits(bs.kf_slots, other.built_slots) ||
                        ((saves || other_saves) && hits(bs.kf_slots, other.kf_slots));
    if (!shared) continue;
    if (other.adaptive && !other.resolved) {
      // a state-driven batch may still need its continuation, which only the host can start: finish it first (the host
      // waits here; batches on disjoint keyframes never do)
      const ellc_status s = resolve_batch(c, p);
      if (s != ELLC_OK) return s;
      select_batch_set(c, set);
    }
    ELLC_HIP(c, hipStreamWaitEvent(run_stream, other.done, 0));
  }
  // The state-driven schedule runs as ONE resident launch (gn_fca_persist) when the call finds the context's pipeline empty — the
  // tracking call, whose latency it shortens — and as one launch per iteration when other groups are in flight: a resident launch
  // wants all its blocks on the device at once, the dispatcher interleaves the blocks of launches on different streams, and two
  // resident launches that each hold part of the device wait for each other until one gives up (r05 soak, three batches of two in
  // flight, exact mode: one abandoned launch in 2 000; ordered one behind the other instead they lose the overlap of the three
  // streams: 0.189 against 0.102 ms per batch). Same bits either way. Launches of other contexts or processes are not known here:
  // against those the abandoned launch and its continuation are the safety net.
  c->cur_resident = false;
  if (may_run_resident(c, bs.mode, B)) {
    bool alone = true;
    for (int p = 0; p < ellc_ctx::SETS; p++)
      if (p != set && c->batch_set[p].launched && !c->batch_set[p].resolved) alone = false;
    c->cur_resident = alone && persist_blocks(c, B) <= c->persist_capacity;
  }
  bs.resident = c->cur_resident;
  c->cur_adaptive_first = adaptive_first_launches(c, B);
  bs.adaptive_first = c->cur_adaptive_first;
  {
    StreamScope scope(c, run_stream);
    bs.pollable = polls_results(c, B, si);
    c->cur_pollable = bs.pollable;
    c->cur_dense = dense;
    c->cur_need = (need_run != need) ? need_run : 0;
    const ellc_status s = launch_align_graph(c, B, nu, bs.mode, bs.save_weights, set, false);
    c->cur_need = 0;
    c->cur_dense = false;
    c->cur_pollable = false;
    if (s != ELLC_OK) {
      for (int v : bs.built_slots) invalidate_records(c, v);
      return s;
    }
    // (ellc_track_frame, whose host side polls the result record: the event goes behind the depth stages it enqueues next — in
    // front of them the record would hold their first launch back ~6 us)
    c->done_deferred = c->track_call && bs.pollable;
    if (!c->done_deferred) ELLC_HIP(c, hipEventRecord(bs.done, c->stream));
  }
  for (int v : bs.built_slots) c->kf_rec_tag[v] = need;
  if (need == 20)
    for (int v : bs.built_slots) c->kf_hinv_ok[v] = 1;   // (computed by this launch, or already current)
  if (saves)   // the weight planes change: lists that carry the saved weight (the constant-weight record sets) are stale, and so are the H^-1
    for (int v : bs.kf_slots) {
      c->kf_hinv_ok[v] = 0;
      if (c->kf_rec_tag[v] != 8 && c->kf_rec_tag[v] != 2) invalidate_records(c, v);
    }
  bs.launched = true;
  bs.stream_idx = si;
  bs.B = B;
  bs.adaptive = schedule_is_adaptive(c, bs.mode, B);
  bs.resolved = false;
  bs.joined = (si == 0);
  return ELLC_OK;
}

// forgets the group of a set (after its last batch has been fetched, or after a failed launch)
static void free_set(ellc_ctx* c, int set) {
  ellc_ctx::BatchSet& bs = c->batch_set[set];
  bs.fill = bs.fetched = 0;
  bs.launched = false;
  bs.resolved = true;
  bs.adaptive = false;
  bs.kf_slots.clear();
  bs.built_slots.clear();
  if (c->open_set == set) c->open_set = -1;
}

// track: the batch is staged in a set (joining the open group when it can, see ellc_ctx::BatchSet) and enters the in-flight
// queue ellc_align_fetch drains; untracked use (the timing hooks, nothing in flight) runs set 0 on the main stream at once.
static ellc_status align_enqueue_impl(ellc_ctx* c, int B, const int* kf_slots, const int* frame_slots, const float* init_pose, int mode,
                                      int save_weights, bool track) {
  if (!c) return ELLC_ERR_BAD_ARG;
  if (mode != ELLC_MODE_FCA && mode != ELLC_MODE_ICA) return fail(c, ELLC_ERR_BAD_ARG, "unknown mode");
  const int MB = c->cfg.max_batch;
  if (!track) {
    if (c->n_inflight > 0 || c->open_set >= 0) return fail(c, ELLC_ERR_NOT_READY, "fetch the enqueued batches first");
    select_batch_set(c, 0);
    int nu = 0;
    ellc_status s = stage_batch(c, B, kf_slots, frame_slots, init_pose, &nu);
    if (s != ELLC_OK) return s;
    for (int b = 0; b < B; b++) invalidate_records(c, kf_slots[b]);   // rebuilt here, outside the cache's bookkeeping
    c->cur_resident = may_run_resident(c, mode, B) && persist_blocks(c, B) <= c->persist_capacity;   // (nothing in flight)
    c->cur_adaptive_first = adaptive_first_launches(c, B);
    bool dense = runs_dense(c, mode, B, save_weights);
    for (int b = 0; b < B; b++) dense = dense && c->kf_dense[kf_slots[b]];
    c->cur_dense = dense;
    s = launch_align_graph(c, B, dense ? 0 : nu, mode, save_weights, 0, false);
    c->cur_dense = false;
    return s;
  }
  // may this batch share a launch with others? full batches of one mode, nothing per-slot written (saved weights), not the
  // state-driven tracking schedule
  const bool co = c->coalesce > 1 && B == MB && !(save_weights && mode == ELLC_MODE_FCA) && !schedule_is_adaptive(c, mode, B);
  if (c->open_set >= 0) {
    ellc_ctx::BatchSet& og = c->batch_set[c->open_set];
    if (!(co && og.coalescable && og.mode == mode && og.fill < c->coalesce)) {
      const ellc_status s = launch_group(c, c->open_set);
      if (s != ELLC_OK) return s;
    }
  }
  int set = c->open_set;
  if (set < 0) {
    // the lowest free set: a caller with one batch at a time stays on set 0
    if (c->n_inflight >= c->max_inflight)
      return fail(c, ELLC_ERR_NOT_READY, "the limit of batches in flight is reached: call ellc_align_fetch first");
    for (int p = 0; p < c->n_sets && set < 0; p++)
      if (c->batch_set[p].fill == 0) set = p;
    if (set < 0) return fail(c, ELLC_ERR_NOT_READY, "no batch set is free: call ellc_align_fetch first");
  }
  ellc_ctx::BatchSet& bs = c->batch_set[set];
  const int slice = bs.fill;
  select_batch_set(c, set, slice);
  int nu = 0;
  std::vector<int> uniq;
  ellc_status s = stage_batch(c, B, kf_slots, frame_slots, init_pose, &nu, &uniq);
  if (s != ELLC_OK) return s;
  // saved weights are accumulated per keyframe slot (wlast, weight plane, numWeightsAdded): two alignments of one batch on the
  // same slot would race on them
  if (save_weights && mode == ELLC_MODE_FCA && nu < B)
    return fail(c, ELLC_ERR_BAD_ARG, "save_weights needs a different keyframe slot for every alignment of the batch");
  if (slice == 0) {
    bs.coalescable = co;
    bs.mode = mode;
    bs.save_weights = save_weights ? 1 : 0;
    bs.fetched = 0;
    bs.launched = false;
  }
  bs.slice_B[slice] = B;
  bs.fill = slice + 1;
  c->inflight[c->n_inflight++] = set * ellc_ctx::MAX_COALESCE + slice;
  if (save_weights && mode == ELLC_MODE_FCA)
    for (int b = 0; b < B; b++)
      for (int l = 0; l < c->L; l++) c->kf_num_weights[kf_slots[b]][l]++;
  if (co && bs.fill < c->coalesce) {
    c->open_set = set;   // waits for more batches (or for a fetch / another entry point, which launch it as it is)
    return ELLC_OK;
  }
  s = launch_group(c, set);
  if (s != ELLC_OK) {   // the batches of the group leave the queue: nothing of it runs
    int w = 0;
    for (int i = 0; i < c->n_inflight; i++)
      if (c->inflight[i] / ellc_ctx::MAX_COALESCE != set) c->inflight[w++] = c->inflight[i];
    c->n_inflight = w;
    free_set(c, set);
  }
  return s;
}

ellc_status ellc_align_enqueue(ellc_ctx* c, int B, const int* kf_slots, const int* frame_slots, const float* init_pose, int mode, int save_weights) {
  ELLC_ENTER_BATCH(c);
  return align_enqueue_impl(c, B, kf_slots, frame_slots, init_pose, mode, save_weights, true);
}

ellc_status ellc_align_fetch(ellc_ctx* c, int B, float* out_pose, int* out_iters, float* out_weighted) {
  ELLC_ENTER_BATCH(c);
  if (!c || B < 1 || B > c->cfg.max_batch) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (c->n_inflight < 1) return fail(c, ELLC_ERR_NOT_READY, "ellc_align_fetch: no batch in flight");
  const int set = c->inflight[0] / ellc_ctx::MAX_COALESCE, slice = c->inflight[0] % ellc_ctx::MAX_COALESCE;   // the oldest batch
  ellc_ctx::BatchSet& bs = c->batch_set[set];
  if (B != bs.slice_B[slice]) return fail(c, ELLC_ERR_BAD_ARG, "ellc_align_fetch: the oldest batch in flight has a different size");
  ellc_status rs = ELLC_OK;
  if (!bs.launched) rs = launch_group(c, set);   // its group was still open: it runs as it is
  if (rs == ELLC_OK) rs = resolve_batch(c, set);   // waits; runs the continuation of a state-driven schedule if one is needed
  for (int i = 1; i < c->n_inflight; i++) c->inflight[i - 1] = c->inflight[i];   // the batch leaves the queue either way
  c->n_inflight--;
  const ellc::AlignResult* res = bs.result_h + slice * c->cfg.max_batch;
  bs.fetched++;
  const bool last = (bs.fetched >= bs.fill);
  ellc_status out = rs;
  if (out == ELLC_OK) {
    for (int b = 0; b < B && out == ELLC_OK; b++)
      if (res[b].pad != 0) {   // the last kernel of the schedule never wrote the record
        out = fail(c, ELLC_ERR_HIP, "ellc_align_fetch: the schedule did not complete on the device (no result was exported)");
      }
  }
  if (out == ELLC_OK && bs.adaptive) {   // the next state-driven call starts with a graph as long as this one needed, plus two
    int most = 0;
    for (int b = 0; b < B; b++) {
      int it = 0;
      for (int l = 0; l < c->L; l++) it += res[b].iters[l];
      most = std::max(most, it);
    }
    c->adaptive_hint = std::min(schedule_total_iters(c), std::max(c->L, (most + 3) & ~1));
  }
  if (out == ELLC_OK)
    for (int b = 0; b < B; b++) {
      if (out_pose) std::memcpy(out_pose + b * 6, res[b].pose, 24);
      if (out_iters) for (int l = 0; l < c->L; l++) out_iters[b * c->L + l] = res[b].iters[l];
      if (out_weighted) out_weighted[b] = res[b].weighted;
    }
  if (last) free_set(c, set);
  return out;
}

ellc_status ellc_align(ellc_ctx* c, int B, const int* kf_slots, const int* frame_slots, const float* init_pose, int mode, int save_weights,
                       float* out_pose, int* out_iters, float* out_weighted) {
  ELLC_ENTER(c);
  if (c && c->n_inflight > 0) return fail(c, ELLC_ERR_NOT_READY, "ellc_align: fetch the enqueued batches first");
  ellc_status s = ellc_align_enqueue(c, B, kf_slots, frame_slots, init_pose, mode, save_weights);
  if (s != ELLC_OK) return s;
  return ellc_align_fetch(c, B, out_pose, out_iters, out_weighted);
}

// set pose / S of state[0] without touching Hinv (ICA iterations reuse the level's precomputed inverse)
__global__ void gn_set_pose0(AlignState* state, const float* pose) {
  AlignState& st = state[0];
  float p[6], S[12];
  for (int i = 0; i < 6; i++) { p[i] = pose[i]; st.pose[i] = p[i]; }
  exp_se3_f32(p, S);
  for (int i = 0; i < 12; i++) st.S[i] = S[i];
  st.level_done = -1;
}

ellc_status ellc_gn_iterate(ellc_ctx* c, int kf_slot, int frame_slot, int level, int mode, int iter, const float* pose, float* H36, float* b6,
                            float* delta6, float* new_pose6, float* weighted, float* planes) {
  ELLC_ENTER(c);
  if (!c || !pose || level < 0 || level >= c->L) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (c->n_inflight > 0) return fail(c, ELLC_ERR_NOT_READY, "ellc_gn_iterate: fetch the enqueued batches first");
  int nu = 0;
  select_batch_set(c, 0);
  invalidate_records(c, kf_slot);   // the single-step API builds its own record set
  ellc_status s = stage_batch(c, 1, &kf_slot, &frame_slot, pose, &nu);
  if (s != ELLC_OK) return s;
  enqueue_stage_in(c, 0);   // staging only: the state keeps the level's H^-1 (gn_set_pose0)
  s = run_prep(c, nu, mode == ELLC_MODE_ICA ? 1 : (c->fast ? 8 : 2));
  if (s != ELLC_OK) return s;
  hipLaunchKernelGGL(gn_set_pose0, dim3(1), dim3(1), 0, c->stream, c->state_d, c->init_pose_d);
  const size_t n = (size_t)c->geom_h[level].n;
  if (planes) ELLC_HIP(c, hipMemsetAsync(c->planes_d, 0, 10 * n * 4, c->stream));
  GnArgs a = make_gn_args(c, level, 1, 0, planes ? c->planes_d : nullptr);
  const dim3 grd(a.nblk, 1), blk(ELLC_GN_THREADS);
  if (mode == ELLC_MODE_FCA) {
    if (planes && c->fast) hipLaunchKernelGGL((gn_fca_accumulate<true, false, true>), grd, blk, 0, c->stream, a);
    else if (planes && c->geom_h[0].divc_ok) hipLaunchKernelGGL((gn_fca_accumulate<true, true>), grd, blk, 0, c->stream, a);
    else if (planes) hipLaunchKernelGGL((gn_fca_accumulate<true, false>), grd, blk, 0, c->stream, a);
    else launch_fca(c, grd, blk, a);
    launch_solve(c, level, 1, a.nblk, 0, 0);
  } else {
    if (iter == 0) {
      hipLaunchKernelGGL(gn_ica_precompute, grd, blk, 0, c->stream, a, c->cap[level]);
      launch_solve(c, level, 1, a.nblk, 1, 0);
    }
    if (planes) hipLaunchKernelGGL(gn_ica_iterate<true>, grd, blk, 0, c->stream, a, c->cap[level]);
    else hipLaunchKernelGGL(gn_ica_iterate<false>, grd, blk, 0, c->stream, a, c->cap[level]);
    launch_solve(c, level, 1, a.nblk, 2, 0);
  }
  ELLC_HIP(c, hipGetLastError());
  ELLC_HIP(c, hipMemcpyAsync(c->state_h, c->state_d, sizeof(AlignState), hipMemcpyDeviceToHost, c->stream));
  if (planes) {
    // masked pixels: savedWarpedPoints = -2 (PixelWisePyramid.cpp:218-219); the kernel only touches valid pixels
    ELLC_HIP(c, hipMemcpyAsync(planes, c->planes_d, 10 * n * 4, hipMemcpyDeviceToHost, c->stream));
  }
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  const AlignState& st = c->state_h[0];
  if (H36) std::memcpy(H36, st.H, 144);
  if (b6) std::memcpy(b6, st.b, 24);
  if (delta6) std::memcpy(delta6, st.delta, 24);
  if (new_pose6) std::memcpy(new_pose6, st.pose, 24);
  if (weighted) *weighted = st.weighted;
  return ELLC_OK;
}

ellc_status ellc_gn_display_planes(ellc_ctx* c, int kf_slot, int frame_slot, int level, const float* pose, uint8_t* templateimg,
                                   uint8_t* tobewarpedimg, float* warpedimg, float* origres) {
  ELLC_ENTER(c);
  if (!c || !pose || level < 0 || level >= c->L) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (c->n_inflight > 0) return fail(c, ELLC_ERR_NOT_READY, "ellc_gn_display_planes: fetch the enqueued batches first");
  int nu = 0;
  select_batch_set(c, 0);
  ellc_status s = stage_batch(c, 1, &kf_slot, &frame_slot, pose, &nu);
  if (s != ELLC_OK) return s;
  enqueue_stage_in(c, 0);   // staging only
  hipLaunchKernelGGL(gn_set_pose0, dim3(1), dim3(1), 0, c->stream, c->state_d, c->init_pose_d);
  const LevelGeom& g = c->geom_h[level];
  const size_t n = (size_t)g.n;
  float* warped_d = c->planes_d;             // scratch: two f32 planes and two u8 planes of the level
  float* orig_d = c->planes_d + n;
  uint8_t* tmpl_d = (uint8_t*)(c->planes_d + 2 * n);
  uint8_t* tbw_d = tmpl_d + n;
  GnArgs a = make_gn_args(c, level, 1, 0, nullptr);
  dim3 blk(32, 8);
  hipLaunchKernelGGL(gn_display_planes, grid2d(g.cols, g.rows, blk), blk, 0, c->stream, a, tmpl_d, tbw_d, warped_d, orig_d);
  ELLC_HIP(c, hipGetLastError());
  if (templateimg) ELLC_HIP(c, hipMemcpyAsync(templateimg, tmpl_d, n, hipMemcpyDeviceToHost, c->stream));
  if (tobewarpedimg) ELLC_HIP(c, hipMemcpyAsync(tobewarpedimg, tbw_d, n, hipMemcpyDeviceToHost, c->stream));
  if (warpedimg) ELLC_HIP(c, hipMemcpyAsync(warpedimg, warped_d, n * 4, hipMemcpyDeviceToHost, c->stream));
  if (origres) ELLC_HIP(c, hipMemcpyAsync(origres, orig_d, n * 4, hipMemcpyDeviceToHost, c->stream));
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  return ELLC_OK;
}

void ellc_concatenate_relative_pose(const float* a, const float* b, float* dest) {
  float o[6];
  concat_relative_f32(a, b, o);
  std::memcpy(dest, o, sizeof(o));
}
void ellc_concatenate_origin_pose(const float* a, const float* b, float* dest) {
  float o[6];
  concat_origin_f32(a, b, o);
  std::memcpy(dest, o, sizeof(o));
}
void ellc_se3_exp(const float* pose6, float* T16) {
  float S[12];
  exp_se3_f32(pose6, S);
  std::memcpy(T16, S, sizeof(S));
  T16[12] = T16[13] = T16[14] = 0.0f;
  T16[15] = 1.0f;
}
void ellc_se3_log(const float* T16, float* pose6) {
  float S[12];
  std::memcpy(S, T16, sizeof(S));
  log_se3_f32(S, pose6);
}

// ---- measurement hooks and device self-tests: include/ellc_abi_diag.h, compiled into libellc_hip_diag.so only (-DELLC_DIAG_ABI) ----
#ifdef ELLC_DIAG_ABI
ellc_status ellc_profile_gn_kernel(ellc_ctx* c, int B, const int* kf_slots, const int* frame_slots, int level, int reps, float* avg_ms,
                                   double* algorithmic_bytes, long long* valid_pixels) {
  ELLC_ENTER(c);
  if (!c || level < 0 || level >= c->L || reps < 1) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (c->n_inflight > 0) return fail(c, ELLC_ERR_NOT_READY, "ellc_profile_gn_kernel: fetch the enqueued batches first");
  int nu = 0;
  select_batch_set(c, 0);
  if (kf_slots)
    for (int b = 0; b < B; b++) invalidate_records(c, kf_slots[b]);
  ellc_status s = stage_batch(c, B, kf_slots, frame_slots, nullptr, &nu, nullptr, true);   // up to a whole launch group (cfg.coalesce batches)
  if (s != ELLC_OK) return s;
  bool dense = runs_dense(c, ELLC_MODE_FCA, B, 0);   // the kernel the production schedule would launch for these keyframes
  for (int b = 0; b < B; b++) dense = dense && c->kf_dense[kf_slots[b]];
  enqueue_stage_in(c, 0);
  if (!dense) {
    s = run_prep(c, nu, c->fast ? 8 : 2);
    if (s != ELLC_OK) return s;
  }
  hipLaunchKernelGGL(gn_init_state, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->state_d, c->init_pose_d, B, c->L - 1);
  c->cur_dense = dense;   // (the grids of list-free launches are sized for their kernel: choose_nblk)
  GnArgs a = make_gn_args(c, level, B, 0, nullptr);
  c->cur_dense = false;
  const dim3 grd(a.nblk, B), blk(ELLC_GN_THREADS);
  if (c->use_fused) {
    // the production kernel of the FCA path: every launch first solves the previous launch's partial sums
    FusedArgs fa;
    fa.continuation = 0;
    set_track_fields(c, fa, false);
  set_track_fields(c, fa, false);
    fa.g = a;
    fa.res = nullptr;
    fa.ica = 0;
    fa.xcd_map = (B % 8 == 0) ? 1 : 0;
    set_age_split(c, fa, B);
    fa.seq = 0;
    fa.prev_level = level;
    fa.prev_nblk = a.nblk;
    fa.early_exit = 0;
    fa.stride_state = c->group_cap;
    fa.stride_part = (size_t)c->group_cap * ELLC_NBLK_MAX * ELLC_PART_STRIDE;
    c->cur_dense = dense;
    auto launch = [&]() {
      launch_fused(c, grd, blk, fa, c->stream);
      fa.seq++;
    };
    for (int i = 0; i < 3; i++) launch();
    if (c->use_graph) {
      // as in production: the launches are replayed from a captured graph (kernel arguments resident on the device)
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      ELLC_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < reps; i++) launch();
      hipError_t e = hipStreamEndCapture(c->stream, &graph);
      if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (e != hipSuccess) { c->cur_dense = false; return fail(c, ELLC_ERR_HIP, std::string("profile graph capture: ") + hipGetErrorString(e)); }
      e = hipGraphLaunch(exec, c->stream);   // warm
      if (e == hipSuccess) e = hipEventRecord(c->ev0, c->stream);
      if (e == hipSuccess) e = hipGraphLaunch(exec, c->stream);
      if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
      if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
      (void)hipGraphExecDestroy(exec);
      if (e != hipSuccess) { c->cur_dense = false; return fail(c, ELLC_ERR_HIP, std::string("profile graph: ") + hipGetErrorString(e)); }
    } else {
      ELLC_HIP(c, hipEventRecord(c->ev0, c->stream));
      for (int i = 0; i < reps; i++) launch();
      ELLC_HIP(c, hipEventRecord(c->ev1, c->stream));
    }
  } else {
    for (int i = 0; i < 3; i++) launch_fca(c, grd, blk, a);
    ELLC_HIP(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; i++) launch_fca(c, grd, blk, a);
    ELLC_HIP(c, hipEventRecord(c->ev1, c->stream));
  }
  c->cur_dense = false;
  ELLC_HIP(c, hipEventSynchronize(c->ev1));
  float ms = 0;
  ELLC_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if (avg_ms) *avg_ms = ms / reps;
  long long V = 0;
  for (int b = 0; b < B; b++) {
    int v = dense ? c->geom_h[level].n : 0;   // (no list, no count: the hint says at least nine tenths; the figure is the plane's size)
    if (!dense)
    ELLC_HIP(c, copy_blocking(c, &v, c->kf_tab_h[(size_t)level * c->cfg.max_keyframes + kf_slots[b]].count, 4, hipMemcpyDeviceToHost));
    V += v;
  }
  if (valid_pixels) *valid_pixels = V;
  if (algorithmic_bytes) *algorithmic_bytes = 4.0 * (double)c->geom_h[level].n * B + 14.0 * (double)V;
  return ELLC_OK;
}

ellc_status ellc_profile_align(ellc_ctx* c, int B, const int* kf_slots, const int* frame_slots, const float* init_pose, int mode, int reps, float* avg_ms) {
  ELLC_ENTER(c);
  if (!c || reps < 1) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  if (c->n_inflight > 0) return fail(c, ELLC_ERR_NOT_READY, "ellc_profile_align: fetch the enqueued batches first");
  ellc_status s = align_enqueue_impl(c, B, kf_slots, frame_slots, init_pose, mode, 0, false);
  if (s != ELLC_OK) return s;
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  ELLC_HIP(c, hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < reps; i++) {
    s = align_enqueue_impl(c, B, kf_slots, frame_slots, init_pose, mode, 0, false);
    if (s != ELLC_OK) return s;
  }
  ELLC_HIP(c, hipEventRecord(c->ev1, c->stream));
  const hipError_t pe = hipEventSynchronize(c->ev1);
  ELLC_HIP(c, pe);
  float ms = 0;
  ELLC_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if (avg_ms) *avg_ms = ms / reps;
  return ELLC_OK;
}

__global__ __launch_bounds__(256) void calib_read_f32(const float* __restrict__ p, size_t n, float* __restrict__ sink) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 1.2345e-30f) sink[0] = acc;   // keeps the loads alive
}

#ifdef ELLC_SEQ_STAMPS
// experiment build only: entry / exit stamps of block (0, 0) of the 32 launches of the last FCA sequence (100 MHz)
extern "C" ellc_status ellc_debug_seq_stamps(ellc_ctx* c, unsigned long long* out64) {
  if (!c || !out64) return ELLC_ERR_BAD_ARG;
  ELLC_ENTER(c);
  ELLC_HIP(c, hipDeviceSynchronize());
  ELLC_HIP(c, hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_seq_stamps), 64 * sizeof(unsigned long long)));
  return ELLC_OK;
}
#endif
#ifdef ELLC_STAMPS
// diagnostic build only: copies the cycle stamps of block (0,0) of the last fused launch
ellc_status ellc_debug_stamps(ellc_ctx* c, unsigned long long* out64) {
  ELLC_ENTER(c);
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  ELLC_HIP(c, hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_stamps), 64 * sizeof(unsigned long long)));
  return ELLC_OK;
}
// the last resident launch's rounds as eight of its blocks saw them (g_ptrace): 8 x 64 x 12 words
ellc_status ellc_debug_persist_trace(ellc_ctx* c, unsigned long long* out) {
  ELLC_ENTER(c);
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  ELLC_HIP(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ptrace), sizeof(unsigned long long) * 8 * 64 * 12));
  unsigned long long* z = (unsigned long long*)calloc(8 * 64 * 12, sizeof(unsigned long long));
  if (z) { ELLC_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(g_ptrace), z, sizeof(unsigned long long) * 8 * 64 * 12)); free(z); }
  return ELLC_OK;
}
ellc_status ellc_debug_block_stamps(ellc_ctx* c, unsigned long long* out, int nblocks) {
  ELLC_ENTER(c);
  ELLC_HIP(c, hipStreamSynchronize(c->stream));
  ELLC_HIP(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_stamps), (size_t)nblocks * 4 * sizeof(unsigned long long)));
  return ELLC_OK;
}
#endif

// device self-test of div_pair_ieee against the compiler's `/` (both quotient arrays are returned)
__global__ void selftest_div_pair(const float* a, const float* b, float* q_pair, float* q_ref, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float q0, q1;
  ellc::div_pair_ieee(a[2 * i], b[2 * i], a[2 * i + 1], b[2 * i + 1], q0, q1);
  q_pair[2 * i] = q0;
  q_pair[2 * i + 1] = q1;
  q_ref[2 * i] = a[2 * i] / b[2 * i];
  q_ref[2 * i + 1] = a[2 * i + 1] / b[2 * i + 1];
}
ellc_status ellc_selftest_div_pair(ellc_ctx* c, int n, const float* a, const float* b, float* q_pair, float* q_ref) {
  ELLC_ENTER(c);
  if (!c || n < 2 || (n & 1) || !a || !b || !q_pair || !q_ref) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  float* d = nullptr;
  ELLC_HIP(c, hipMalloc(&d, (size_t)4 * n * sizeof(float)));
  hipError_t e = hipMemcpyAsync(d, a, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d + n, b, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(selftest_div_pair, dim3((n / 2 + 255) / 256), dim3(256), 0, c->stream, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(q_pair, d + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(q_ref, d + 3 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(c, ELLC_ERR_HIP, std::string("selftest_div_pair: ") + hipGetErrorString(e));
  return ELLC_OK;
}

// device self-test of the 6x6 LU inverse: n symmetric matrices given by their 21 upper-triangular entries (f64), one wave each
__global__ void selftest_lu(const double* tri21, float* inv36, int n) {
  const int m = blockIdx.x, lane = threadIdx.x;
  if (m >= n) return;
  float Hm[36];
  int q = 0;
  for (int r = 0; r < 6; r++)
    for (int cc = r; cc < 6; cc++) {
      const float v = (float)tri21[(size_t)m * 21 + q++];
      Hm[r * 6 + cc] = v;
      Hm[cc * 6 + r] = v;
    }
  float x[6];
  ellc::lu_inverse6_lanes(Hm, lane < 6 ? lane : 0, x);
  if (lane < 6)
    for (int r = 0; r < 6; r++) inv36[(size_t)m * 36 + r * 6 + lane] = x[r];
}
ellc_status ellc_selftest_lu(ellc_ctx* c, int n, const double* tri21, float* inv36) {
  ELLC_ENTER(c);
  if (!c || n < 1 || !tri21 || !inv36) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  double* d = nullptr;
  float* o = nullptr;
  ELLC_HIP(c, hipMalloc(&d, (size_t)n * 21 * sizeof(double)));
  hipError_t e = hipMalloc(&o, (size_t)n * 36 * sizeof(float));
  if (e == hipSuccess) e = hipMemcpyAsync(d, tri21, (size_t)n * 21 * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(selftest_lu, dim3(n), dim3(64), 0, c->stream, d, o, n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(inv36, o, (size_t)n * 36 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (o) (void)hipFree(o);
  if (e != hipSuccess) return fail(c, ELLC_ERR_HIP, std::string("selftest_lu: ") + hipGetErrorString(e));
  return ELLC_OK;
}

ellc_status ellc_profile_calibrate_read(ellc_ctx* c, size_t bytes, int reps, float* avg_ms) {
  ELLC_ENTER(c);
  if (!c || reps < 1 || bytes < 1024) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  float* buf = nullptr;
  ELLC_HIP(c, hipMalloc((void**)&buf, bytes));
  ELLC_HIP(c, hipMemsetAsync(buf, 0, bytes, c->stream));
  const size_t n = bytes / 4;
  hipLaunchKernelGGL(calib_read_f32, dim3(2048), dim3(256), 0, c->stream, buf, n, c->scratch_a);
  ELLC_HIP(c, hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(calib_read_f32, dim3(2048), dim3(256), 0, c->stream, buf, n, c->scratch_a);
  ELLC_HIP(c, hipEventRecord(c->ev1, c->stream));
  ELLC_HIP(c, hipEventSynchronize(c->ev1));
  float ms = 0;
  ELLC_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if (avg_ms) *avg_ms = ms / reps;
  (void)hipFree(buf);
  return ELLC_OK;
}

__global__ __launch_bounds__(256) void stream_read_f32x4(const float4* __restrict__ p, size_t n, float* __restrict__ sink) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = p[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  if (acc == 1.2345e-30f) sink[0] = acc;   // keeps the loads alive
}

// Streaming-read rate of this device with 16-byte lanes (the widest global load): what "HBM peak" amounts to in practice
// for a kernel that does nothing but read. bench.py reports it beside the 8 TB/s the roofline is priced against.
ellc_status ellc_profile_stream_read(ellc_ctx* c, size_t bytes, int reps, float* avg_ms) {
  ELLC_ENTER(c);
  if (!c || reps < 1 || bytes < 4096) return fail(c, ELLC_ERR_BAD_ARG, "bad argument");
  float4* buf = nullptr;
  ELLC_HIP(c, hipMalloc((void**)&buf, bytes));
  hipError_t e = hipMemsetAsync(buf, 0, bytes, c->stream);
  const size_t n = bytes / 16;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(stream_read_f32x4, dim3(4096), dim3(256), 0, c->stream, buf, n, c->scratch_a);
    e = hipEventRecord(c->ev0, c->stream);
  }
  if (e == hipSuccess) {
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(stream_read_f32x4, dim3(4096), dim3(256), 0, c->stream, buf, n, c->scratch_a);
    e = hipEventRecord(c->ev1, c->stream);
  }
  if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(c, ELLC_ERR_HIP, std::string("ellc_profile_stream_read: ") + hipGetErrorString(e));
  if (avg_ms) *avg_ms = ms / reps;
  return ELLC_OK;
}
#endif   // ELLC_DIAG_ABI

}  // extern "C"

#include "ellc_depth_impl.hpp"
#include "ellc_ingest_impl.hpp"
