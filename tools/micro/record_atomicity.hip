// r06 (r05 verdict, weak 4b): gn_fca_persist passes partial sums between blocks as tagged 128-byte records — 32 consecutive lanes store
// the line with relaxed agent-scope stores (28 payload words, the tag in the last word of each of the four 32-byte sectors), a reader
// loads the whole line with relaxed agent-scope loads and TAKES it when all four tags name the round it waits for. No fence, no
// acquire / release. What that relies on: within one 32-byte sector, a reader never sees the NEW tag beside an OLD payload word (or
// the other way round) — i.e. a wave's store of a sector becomes visible as a unit. This program hammers exactly that: writer blocks
// on XCDs 0-3 rewrite their line back to back with sequence numbers 1 .. N (payload word w of sequence s holds s * 64 + w, so every
// word names its sequence), reader blocks on XCDs 4-7 (block b + 4: workgroups are dealt round-robin over the 8 XCDs) load it as
// the product does and check every line whose four tags agree. Counts: lines taken (four equal tags), lines whose payload
// disagreed with their tags (VIOLATIONS: must be 0), torn lines (tags differ: expected, retried by the product), distinct sequences seen.
// build: hipcc --offload-arch=gfx950 -O3 -o build/record_atomicity tools/micro/record_atomicity.hip ; run: build/record_atomicity [N per pair]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
struct Counts { unsigned long long taken, violations, torn, distinct, polls; };
__device__ __forceinline__ unsigned payload(unsigned seq, int w) { return seq * 64u + (unsigned)w; }
__global__ __launch_bounds__(64) void hammer(unsigned* lines, unsigned nseq, Counts* counts, int pairs) {
  const int blk = blockIdx.x, lane = threadIdx.x, w = lane & 31;
  const int xcd = blk & 7, grp = blk >> 3;
  const bool writer = xcd < 4;
  const int pair = grp * 4 + (xcd & 3);
  if (pair >= pairs) return;
  unsigned* line = lines + (size_t)pair * 32;
  const bool is_tag = (w & 7) == 7;
  if (writer) {
    for (unsigned s = 1; s <= nseq; s++) {
      if (lane < 32) __hip_atomic_store(line + w, is_tag ? s : payload(s, w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the product's writers do ~1 us of arithmetic between two records of one slot; back to back is the harsher case)
    }
  } else {
    unsigned long long taken = 0, violations = 0, torn = 0, distinct = 0, polls = 0;
    unsigned last = 0;
    const unsigned long long limit = 200ull * nseq + 100000000ull;
    for (;;) {
      const unsigned v = __hip_atomic_load(line + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      polls++;
      const unsigned t0 = (unsigned)__shfl((int)v, 7, 64);
      const bool tags_agree = __ballot(is_tag && v != t0) == 0ull;
      if (tags_agree && t0 != 0u) {
        taken++;
        if (__ballot(!is_tag && v != payload(t0, w)) != 0ull) violations++;   // every payload word must name the tags' sequence
        if (t0 != last) { distinct++; last = t0; }
        if (t0 == nseq) break;
      } else if (!tags_agree) {
        torn++;
      }
      if (polls > limit) break;
    }
    if (lane == 0) {
      atomicAdd(&counts->taken, taken); atomicAdd(&counts->violations, violations); atomicAdd(&counts->torn, torn);
      atomicAdd(&counts->distinct, distinct); atomicAdd(&counts->polls, polls);
    }
  }
}
int main(int argc, char** argv) {
  const unsigned nseq = argc > 1 ? (unsigned)atoll(argv[1]) : 2000000u;
  const int pairs = 128, blocks = pairs * 2;
  unsigned* lines; Counts* counts;
  (void)hipMalloc(&lines, (size_t)pairs * 128); (void)hipMemset(lines, 0, (size_t)pairs * 128);
  (void)hipMalloc(&counts, sizeof(Counts)); (void)hipMemset(counts, 0, sizeof(Counts));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(hammer, dim3(blocks), dim3(64), 0, 0, lines, nseq, counts, pairs);
  (void)hipEventRecord(e1, 0);
  const hipError_t e = hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  Counts c; (void)hipMemcpy(&c, counts, sizeof(c), hipMemcpyDeviceToHost);
  std::printf("{\"pairs\": %d, \"records_written\": %llu, \"lines_taken_with_four_equal_tags\": %llu, \"distinct_sequences_seen\": %llu, \"torn_lines_seen\": %llu, "
              "\"polls\": %llu, \"violations\": %llu, \"ms\": %.1f, \"hip\": \"%s\"}\n",
              pairs, (unsigned long long)pairs * nseq, c.taken, c.distinct, c.torn, c.polls, c.violations, ms, hipGetErrorString(e));
  return (e == hipSuccess && c.violations == 0) ? 0 : 1;
}
