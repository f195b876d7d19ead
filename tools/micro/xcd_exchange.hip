// r06: what one round of gn_fca_persist's record exchange costs by WHERE its blocks sit and HOW the record is stored. K blocks each
// store a tagged 128-byte record (32 lanes, tag in the last word of each 32-byte sector, as persist_store_record) and then gather
// all K records of the round (256 threads, 8 loads in flight per thread, retried until the four tags match, as persist_group_sum);
// R rounds back to back, two buffers alternating by round. Forms:
//   spread / sc1 : the K blocks are blocks 0 .. K-1 of the grid (dealt round-robin over the 8 XCDs), agent-scope stores and loads
//                  (what the product does)
//   home   / sc1 : the K blocks are blocks 0, 8, 16, ... (ONE XCD: checked here through HW_REG_XCC_ID), same accesses
//   (each with only the K writers gathering, and with all 256 blocks of the grid gathering, as the product's blocks all follow the solve)
//   home   / plain: the same blocks, the record stored with plain stores (the line stays in the XCD's L2), agent-scope loads (served
//                  by that L2) — coherent only because writer and reader share the L2
// Prints µs per round (device clock of block 0) and the payload mismatches (every word names its round and writer).
// build: hipcc --offload-arch=gfx950 -O3 -o build/xcd_exchange tools/micro/xcd_exchange.hip ; run: build/xcd_exchange [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define REC_STRIDE 32
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }
template <int PLAIN>
__global__ __launch_bounds__(256) void exch(unsigned* recs, int K, int stride, int rounds, int work_sleep, int all_read, unsigned long long* out, unsigned* xcc_of, unsigned* bad) {
  const int blk = blockIdx.x, t = threadIdx.x;
  if (t == 0) xcc_of[blk] = xcc_id();
  const bool writer = blk % stride == 0 && blk / stride < K;
  if (!writer && !all_read) return;   // all_read: the other blocks of the grid gather every round too (the product's blocks all follow the solve)
  const int me = blk / stride;
  const int comp = t & 31, grp = t >> 5, lane = t & 63;
  const bool is_tag = (comp & 7) == 7;
  const unsigned long long half = (lane < 32) ? 0x00000000ffffffffull : 0xffffffff00000000ull;
  unsigned mism = 0, spins = 0;
  __shared__ int dummy;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < rounds; r++) {
    const unsigned tag = (unsigned)(r + 1);
    unsigned* buf = recs + (size_t)(r & 1) * 256 * REC_STRIDE;
    if (writer && t < 32) {
      const unsigned v = is_tag ? tag : (tag * 4096u + (unsigned)me * 32u + (unsigned)t);
      if (PLAIN) *(volatile unsigned*)(buf + (size_t)me * REC_STRIDE + t) = v;
      else __hip_atomic_store(buf + (size_t)me * REC_STRIDE + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int base = 0; base < K; base += 64) {
      unsigned w[8];
      unsigned okm = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) { w[j] = 0u; if (base + grp + 8 * j >= K) okm |= 1u << j; }
      for (;;) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int k = base + grp + 8 * j;
          if (!((okm >> j) & 1u)) w[j] = __hip_atomic_load(buf + (size_t)k * REC_STRIDE + comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const bool miss = !((okm >> j) & 1u);
          const unsigned long long badm = __ballot(miss && is_tag && w[j] != tag);
          if (miss && (badm & half) == 0ull) okm |= 1u << j;
        }
        if (__ballot(okm != 0xffu) == 0ull) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0u) {   // an exit every wave reaches: a form whose stores never arrive must not hang the device
          if (spins > (1u << 18) && lane == 0) __hip_atomic_store(bad + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__hip_atomic_load(bad + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = base + grp + 8 * j;
        if (k < K && !is_tag && w[j] != tag * 4096u + (unsigned)k * 32u + (unsigned)comp) mism++;
      }
    }
    __syncthreads();
    if (work_sleep > 0) {   // stands for the solve + the pixel pass between two exchanges
      for (int i = 0; i < work_sleep; i++) __builtin_amdgcn_s_sleep(8);
      if (t == 0) dummy = r;
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (mism) atomicAdd(bad, mism);
  if (blk == 0 && t == 0) out[0] = t1 - t0;
}
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned* recs; unsigned long long* out; unsigned* xcc; unsigned* bad;
  (void)hipMalloc(&recs, 2 * 256 * REC_STRIDE * 4); (void)hipMalloc(&out, 8); (void)hipMalloc(&xcc, 256 * 4); (void)hipMalloc(&bad, 8);
  unsigned hx[256];
  printf("{\"rounds\": %d, \"forms\": [\n", rounds);
  bool first = true;
  for (int pass = 0; pass < 4; pass++) {
    const int work = (pass & 1) ? 24 : 0, all_read = pass >> 1;
    const int Ks[4] = {7, 30, 32, 120};
    for (int ki = 0; ki < 4; ki++) {
      for (int form = 0; form < 3; form++) {
        const int K = Ks[ki];
        const int stride = form == 0 ? 1 : 8;
        if (stride == 8 && K > 32) continue;
        (void)hipMemset(recs, 0, 2 * 256 * REC_STRIDE * 4); (void)hipMemset(bad, 0, 8);
        double best = 1e30;
        unsigned hb = 0;
        int same = 1;
        for (int rep = 0; rep < 3; rep++) {
          (void)hipMemset(recs, 0, 2 * 256 * REC_STRIDE * 4);
          if (form == 2) hipLaunchKernelGGL(exch<1>, dim3(256), dim3(256), 0, 0, recs, K, stride, rounds, work, all_read, out, xcc, bad);
          else hipLaunchKernelGGL(exch<0>, dim3(256), dim3(256), 0, 0, recs, K, stride, rounds, work, all_read, out, xcc, bad);
          if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
          unsigned long long ticks;
          (void)hipMemcpy(&ticks, out, 8, hipMemcpyDeviceToHost);
          (void)hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
          for (int b = 8; b < 256; b++) if (hx[b] != hx[b & 7]) same = 0;   // blocks b and b + 8 share an XCD?
          best = ticks / 100.0 / rounds < best ? ticks / 100.0 / rounds : best;
        }
        unsigned hb2[2];
        (void)hipMemcpy(hb2, bad, 8, hipMemcpyDeviceToHost);
        hb = hb2[0];
        if (hb2[1]) best = -1.0;   // gave up: records that never arrived
        printf("%s  {\"work_sleeps\": %d, \"every_block_reads\": %d, \"K\": %d, \"placement\": \"%s\", \"stores\": \"%s\", \"us_per_round\": %.3f, \"payload_mismatches\": %u, \"b_and_b_plus_8_share_an_xcd\": %s, \"xcc_of_blocks_0_to_7\": [%u,%u,%u,%u,%u,%u,%u,%u]}",
               first ? "" : ",\n", work, all_read, K, stride == 1 ? "spread" : "home", form == 2 ? "plain" : "sc1", best, hb, same ? "true" : "false",
               hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7]);
        first = false;
      }
    }
  }
  printf("\n]}\n");
  return 0;
}
