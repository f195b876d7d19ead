// The Gauss-Newton schedule as it runs in production: one kernel covers a run of consecutive iterations — a single
// iteration of a fine level, or ALL iterations of one or more coarse levels — for a batch of alignments
// (reference: the level / iteration loops of GetImagePoseEstimate, ImageFunc.cpp:150-292, around
// PixelWisePyramid::calculatePixelWiseParallel, PixelWisePyramid.cpp:416-455, or ...InvCompositional, :917-974).
//
// Grid (blocks per alignment, B). Per iteration every block of an alignment runs the pixel pass over its chunk of the
// keyframe's compact pixel list and leaves one record of partial sums; the block whose arrival ticket comes LAST combines
// the records in a fixed order (f64, the same combine as ever: the result does not depend on which block it is), solves the
// 6x6 system and updates the pose — once per alignment and iteration instead of once per block. If the run continues, it
// publishes the new exp(pose) under a generation number and the other blocks of the alignment, which stay resident and
// poll that number, go on with the next iteration: no kernel boundary between the iterations of a coarse level. A run of
// one iteration needs no waiting at all (the blocks that are not last simply exit), so it has no residency requirement;
// longer runs are only enqueued when all their blocks fit on the device together with those of the other batches in
// flight (ellc_hip.hip: run_fits), every wait is bounded and a timeout ends the run with an error word set.
//
// Inter-block visibility (gfx950: one L2 per XCD, not coherent with each other; MI355X guide, "inter-workgroup
// visibility", the write-through form): everything handed from block to block inside a launch — partial records, the
// published state — is written with agent-scope relaxed atomic stores (global_store ... sc1, write-through), drained with
// s_waitcnt vmcnt(0) by the storing wave before ONE lane signals (ticket add / generation store), and read with agent-scope
// relaxed atomic loads (global_load ... sc1) by the wave that took the last ticket or saw the generation, the other waves
// of the block behind a workgroup barrier. Nothing else a block reads is written during the launch.
#pragma once
#include "ellc_kernels_gn.hpp"

namespace ellc {

#define ELLC_RUN_TIMEOUT_POLLS (1u << 22)   // ~1 s of polling with s_sleep: a bound, never reached by a healthy run

// Per-alignment synchronisation words, 256 bytes, three cache lines used: arrivals, generation / error, published state.
struct RunSync {
  unsigned ticket;      // arrivals, monotonic over the launches of one schedule (zeroed by the schedule's first kernel); a
                        // launch's blocks count from AlignState::ticket_base, which the previous launch's last solver left
  unsigned pad0[15];
  unsigned gen;         // solves published, monotonic likewise (base: AlignState::gen_base)
  unsigned error;       // non-zero: a wait timed out; sticky until the host clears it
  unsigned pad1[14];
  float S[12];          // exp(pose) after `gen` solves
  int level_done;
  unsigned pad2[19];
};
static_assert(sizeof(RunSync) == 256, "RunSync layout");

struct RunArgs {
  const LevelGeom* geom;
  const KfLevelDev* kf_tab;
  const FrLevelDev* fr_tab;
  const int* kf_slot;
  const int* fr_slot;
  AlignState* state;          // [B], one buffer: read when a launch starts, written by its last solver
  float* partials;            // [B][ELLC_NBLK_MAX][ELLC_PART_STRIDE]
  RunSync* sync;              // [B]
  AlignResult* res;           // host-visible result records, written when `final`
  int max_kf, max_fr;
  int lvl_hi, lvl_lo;         // the run covers levels lvl_hi .. lvl_lo (descending), iters[l] iterations each
  int iters[ELLC_MAX_LEVELS];
  int nblk[ELLC_MAX_LEVELS];  // blocks of an alignment that take part at level l (<= gridDim.x; the others only follow the state)
  int early_exit;
  int save_w;
  int final;                  // this launch ends the schedule: export the results
};

template <class T>
__device__ __forceinline__ void store_sc1(T* p, T v) { __hip_atomic_store(as_global_rw(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T load_sc1(const T* p) { return __hip_atomic_load(as_global(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// partial_group_sum over records another block wrote during this launch: agent-scope loads
__device__ __forceinline__ double partial_group_sum_sc1(const float* partials, int nblk) {
  const int comp = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const float* p = partials + comp;
  double s = 0.0;
  for (int base = 0; base < nblk; base += 8 * (ELLC_SOLVE_THREADS / 32)) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = base + grp + j * (ELLC_SOLVE_THREADS / 32);
      v[j] = load_sc1(p + (unsigned)min(k, nblk - 1) * ELLC_PART_STRIDE);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = base + grp + j * (ELLC_SOLVE_THREADS / 32);
      s += (k < nblk) ? (double)v[j] : 0.0;
    }
  }
  return s;
}

// block reduction of NV per-thread accumulators into the block's partial record, written through (sc1) and drained;
// then thread 0 takes the alignment's arrival ticket. Returns the ticket in every thread.
template <int NV>
__device__ __forceinline__ unsigned reduce_publish_ticket(float (&acc)[NV], float* out, unsigned* ticket_word) {
  __shared__ float red[ELLC_GN_THREADS / 64][32];
  __shared__ unsigned s_ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_sum_all<NV>(acc);
  if (lane == 63) {
#pragma unroll
    for (int j = 0; j < NV; j++) red[wave][j] = acc[j];
  }
  __syncthreads();
  if (wave == 0) {   // the one storing wave: stores, drain, ticket — in this order (the ticket signals for the stores)
    if (lane < NV) {
      float s = red[0][lane];
#pragma unroll
      for (int w = 1; w < ELLC_GN_THREADS / 64; w++) s += red[w][lane];
      store_sc1(out + lane, s);
    }
    drain_stores();
    if (lane == 0) s_ticket = __hip_atomic_fetch_add(as_global_rw(ticket_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return s_ticket;
}

// One pass of the run kernel's pixel loop for the three pixel kinds.
template <bool ICA, bool FAST, bool DIVC>
__device__ __forceinline__ void run_pixel_pass(const GnArgs& ga, const KfLevelDev& K, const LevelGeom& g, g_u8 cur, const float* S, int begin,
                                               int end, float (&sums)[ICA ? 6 : 27]) {
  constexpr int stride = ELLC_GN_THREADS;
  int i = begin + (int)threadIdx.x;
  if constexpr (ICA) {
    float acc[6];
#pragma unroll
    for (int q = 0; q < 6; q++) acc[q] = 0.0f;
    for (; i < end; i += stride) ica_accumulate_pixel<FAST>(acc, ica_load_any<FAST>(K, (unsigned)i), g, cur, S);
#pragma unroll
    for (int q = 0; q < 6; q++) sums[q] = acc[q];   // the b sums; H^-1 of the level comes from the keyframe slot
  } else {
    FcaAcc acc;
    fca_acc_zero(acc);
    // software pipeline: the record of pixel i + 256 is requested behind pixel i's tap loads (tap_point); the two record
    // slots alternate through an explicitly unrolled loop body (a register copy of a slot would wait for the load that fills it)
    if (i < end) {
      if constexpr (FAST) {
        FcaInF r0 = fcaf_load(K, (unsigned)i), r1 = r0;
        auto step = [&](const FcaInF& in, FcaInF& fill) {
          const int i1 = i + stride;
          auto prefetch = [&]() { fill = fcaf_load(K, (unsigned)min(i1, end - 1)); };
          fca_accumulate_pixel(acc, fcaf_pixel<false>(ga, K, g, cur, S, (unsigned)i, in, prefetch));
          i += stride;
        };
        for (;;) {
          step(r0, r1);
          if (i >= end) break;
          step(r1, r0);
          if (i >= end) break;
        }
      } else {
        FcaIn r0 = fca_load(K, (unsigned)i), r1 = r0;
        auto step = [&](const FcaIn& in, FcaIn& fill) {
          const int i1 = i + stride;
          auto prefetch = [&]() { fill = fca_load(K, (unsigned)min(i1, end - 1)); };
          fca_accumulate_pixel(acc, fca_pixel_in<false, DIVC>(ga, K, g, cur, S, (unsigned)i, in, prefetch));
          i += stride;
        };
        for (;;) {
          step(r0, r1);
          if (i >= end) break;
          step(r1, r0);
          if (i >= end) break;
        }
      }
    }
    fca_acc_unpack<FAST>(acc, sums);
  }
}

// The solve of one iteration by the block that arrived last. Not inlined: it runs once per alignment and iteration, and
// as a call its registers (the 6x6 elimination, the double-precision se(3) update) do not weigh on the pixel loop's
// allocation.
template <bool ICA, bool FAST>
__device__ __attribute__((noinline)) void run_solve(SolveShared& sh, const float* all_partials, int nb, int level, int early_exit,
                                                    const AlignState& cur, const float* hinv) {
  solve_step<FAST>(sh, partial_group_sum_sc1(all_partials, nb), ICA ? 2 : 0, level, early_exit, cur, nullptr, hinv);
}

template <bool ICA, bool FAST, bool DIVC>
__global__ __launch_bounds__(ELLC_GN_THREADS, 4) void gn_run(RunArgs a) {
  const int b = blockIdx.y, sub = blockIdx.x, t = threadIdx.x;
  AlignState& st = a.state[b];
  RunSync& sy = a.sync[b];
  __shared__ SolveShared sh;
  __shared__ AlignState cur;     // the alignment's state as this block knows it (S, pose, level_done, iters, weighted)
  __shared__ int s_flag;
  // ---- state of the alignment as the previous launch left it (plain loads: nothing in this record is written during the
  // launch before every block of the alignment has read it — the last solver writes it after the final ticket)
  if (t < 12) cur.S[t] = st.S[t];
  if (t < 6) cur.pose[t] = st.pose[t];
  if (t < ELLC_MAX_LEVELS) cur.iters[t] = st.iters[t];
  if (t == 0) { cur.weighted = st.weighted; cur.level_done = st.level_done; }
  const int slot = a.kf_slot[b];
  const int frs = a.fr_slot[b];
  __syncthreads();
  GnArgs ga;                     // the pixel functions' view of the arguments
  ga.geom = a.geom; ga.kf_tab = a.kf_tab; ga.fr_tab = a.fr_tab; ga.kf_slot = a.kf_slot; ga.fr_slot = a.fr_slot;
  ga.state = a.state; ga.partials = a.partials; ga.planes = nullptr; ga.level = 0; ga.max_kf = a.max_kf; ga.max_fr = a.max_fr;
  ga.nblk = 0; ga.save_w = a.save_w;
  float* my_partial = a.partials + ((size_t)b * ELLC_NBLK_MAX + sub) * ELLC_PART_STRIDE;
  const float* all_partials = a.partials + (size_t)b * ELLC_NBLK_MAX * ELLC_PART_STRIDE;
  const unsigned ticket_base = st.ticket_base, gen_base = st.gen_base;   // where the previous launches of the schedule left the counters
  unsigned executed = 0;         // solves of this alignment in this launch so far (= the generation this block has seen)
  unsigned tickets_before = 0;   // arrivals of the iterations already solved
  bool i_solved_last = false;    // this block performed the most recent solve
  bool alive = true;             // false after a timeout: leave without touching anything else
  for (int level = a.lvl_hi; level >= a.lvl_lo && alive; level--) {
    const LevelGeom g = a.geom[level];
    const KfLevelDev K = a.kf_tab[level * a.max_kf + slot];
    const FrLevelDev& F = a.fr_tab[level * a.max_fr + frs];
    g_u8 curimg = as_global(F.img);
    const int nb = a.nblk[level];
    const int V = *as_global(K.count);
    const int chunk = (V + nb - 1) / nb;
    const int begin = sub * chunk;
    const int end = min(V, begin + chunk);
    const bool takes_part = sub < nb;
    const float* hinv = ICA ? K.hinv : nullptr;
    for (int it = 0; it < a.iters[level]; it++) {
      if (cur.level_done == level) break;   // the level ended early (ImageFunc.cpp:251-252): uniform over the alignment's blocks
      unsigned ticket = 0xffffffffu;
      if (takes_part) {
        float S[12];
#pragma unroll
        for (int i = 0; i < 12; i++) S[i] = cur.S[i];
        float sums[ICA ? 6 : 27];
        run_pixel_pass<ICA, FAST, DIVC>(ga, K, g, curimg, S, begin, end, sums);
        ticket = reduce_publish_ticket<ICA ? 6 : 27>(sums, my_partial + (ICA ? 21 : 0), &sy.ticket);   // ICA: the b slots of the record
      }
      const bool last = takes_part && ticket == ticket_base + tickets_before + (unsigned)nb - 1u;   // block-uniform (ticket comes from LDS)
      const bool final_step = (level == a.lvl_lo) && (it + 1 == a.iters[level]);
      i_solved_last = false;
      if (last) {
        // every partial record of this iteration has been written through and drained before its block's ticket add, and
        // this block's add returned last: the records are complete. All waves load them behind the barrier inside
        // reduce_publish_ticket that followed the add.
        run_solve<ICA, FAST>(sh, all_partials, nb, level, a.early_exit, cur, hinv);
        // sh.newS / newpose / weighted / level_done hold the update (solve_step ends with a barrier)
        if (t < 12) cur.S[t] = sh.newS[t];
        if (t < 6) cur.pose[t] = sh.newpose[t];
        if (t == 0) { cur.weighted = sh.weighted; cur.level_done = sh.level_done; cur.iters[level] += 1; }
        if (!final_step && t < 64) {   // the other blocks are waiting: state words written through, drained, then the generation (one lane)
          if (t < 12) store_sc1(&sy.S[t], sh.newS[t]);
          if (t == 12) store_sc1(&sy.level_done, sh.level_done);
          drain_stores();
          if (t == 0) store_sc1(&sy.gen, gen_base + executed + 1u);
        }
        i_solved_last = true;
        __syncthreads();
      } else if (!final_step) {
        // wait for the generation that follows this iteration's solve, then take the published state
        if (t < 64) {
          int ok = 1;
          if (t == 0) {
            unsigned polls = 0;
            while (load_sc1(&sy.gen) < gen_base + executed + 1u) {
              __builtin_amdgcn_s_sleep(2);
              if (++polls > ELLC_RUN_TIMEOUT_POLLS || load_sc1(&sy.error) != 0u) { ok = 0; break; }
            }
            if (!ok) store_sc1(&sy.error, 1u);
          }
          ok = __shfl(ok, 0);
          asm volatile("" ::: "memory");   // the loads below stay behind the poll
          if (ok) {
            if (t < 12) cur.S[t] = load_sc1(&sy.S[t]);
            if (t == 12) cur.level_done = load_sc1(&sy.level_done);
          }
          if (t == 0) s_flag = ok;
        }
        __syncthreads();
        if (!s_flag) { alive = false; break; }
        if (t == 0) cur.iters[level] += 1;   // bookkeeping only (the exporting block is the last solver, whose count is exact)
        __syncthreads();
      } else {
        return;   // final iteration of the launch and not the solver: nothing left to do
      }
      executed += 1u;
      tickets_before += (unsigned)nb;
    }
  }
  if (!alive) return;
  // ---- end of the launch for this alignment. The block that performed the last solve (or, if the launch had nothing to
  // do for this alignment, block 0) leaves the state for the next launch and, at the end of the schedule, the result.
  const bool closer = executed > 0 ? i_solved_last : (sub == 0);
  if (!closer) return;
  if (FAST && a.final) {   // tolerance mode carries exp(pose) through the schedule; the twist is its log, taken once here
    if (t == 0) {
      float S[12], np[6];
      for (int i = 0; i < 12; i++) S[i] = cur.S[i];
      log_se3_f32(S, np);
      for (int i = 0; i < 6; i++) cur.pose[i] = np[i];
    }
    __syncthreads();
  }
  if (executed > 0) {
    if (t < 12) st.S[t] = cur.S[t];
    if (t < 6) st.pose[t] = cur.pose[t];
    if (t < ELLC_MAX_LEVELS) st.iters[t] = cur.iters[t];
    if (t == 0) {
      st.weighted = cur.weighted;
      st.level_done = cur.level_done;
      st.ticket_base = ticket_base + tickets_before;   // the counters are never reset while a block may still poll them
      st.gen_base = gen_base + executed;
    }
  }
  if (a.final && a.res) {
    AlignResult* r = a.res + b;
    if (t < 6) r->pose[t] = cur.pose[t];
    if (t < ELLC_MAX_LEVELS) r->iters[t] = cur.iters[t];
    if (t == 0) { r->weighted = cur.weighted; r->pad = 0; }
  }
}

}  // namespace ellc
