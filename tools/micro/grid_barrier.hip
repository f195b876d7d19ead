// What a grid-wide barrier costs inside one persistent kernel, against a kernel boundary (r04): the tracked frame's alignment is ~16
// dependent launches of a few microseconds of work each, 6.5 us apart. G blocks of 256 threads run ITER rounds; a round: every block
// stores 27 partial sums (agent-scope write-through stores), arrives at a counter (agent-scope atomic), spins until all G have, and
// reads all G x 27 partials back past its own L2 (what the next Gauss-Newton iteration's reduce does). Variants: F = __threadfence()
// + plain stores / loads instead of the per-access scope. Also: the same work as ITER dependent launches of one stream.
// build: hipcc --offload-arch=gfx950 -O3 -o build/grid_barrier tools/micro/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <bool FENCE>
__global__ __launch_bounds__(256) void persistent(float* part, unsigned* counter, int G, int iters, float* out) {
  float acc = 0.0f;
  for (int it = 0; it < iters; it++) {
    float* p = part + (size_t)(it & 1) * G * 32;
    if (threadIdx.x < 27) {
      const float v = (float)(blockIdx.x + it + threadIdx.x) + acc * 1e-9f;
      if (FENCE) p[blockIdx.x * 32 + threadIdx.x] = v;
      else __hip_atomic_store(&p[blockIdx.x * 32 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (FENCE) __threadfence();
    __syncthreads();   // (every wave's stores are complete before the arrival: s_waitcnt vmcnt(0) precedes the barrier)
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, FENCE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)G * (unsigned)(it + 1);
      while (__hip_atomic_load(counter, FENCE ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (FENCE) __threadfence();
    float s = 0.0f;
    for (int k = threadIdx.x; k < G * 27; k += 256) {
      const int b = k / 27, j = k - b * 27;
      s += FENCE ? p[b * 32 + j] : __hip_atomic_load(&p[b * 32 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    acc += s;
  }
  if (acc == -1.0f) out[0] = acc;
}
__global__ __launch_bounds__(256) void one_round(float* part, int G, int it, float* out) {
  const float* q = part + (size_t)((it + 1) & 1) * G * 32;
  float s = 0.0f;
  for (int k = threadIdx.x; k < G * 27; k += 256) { const int b = k / 27, j = k - b * 27; s += q[b * 32 + j]; }
  float* p = part + (size_t)(it & 1) * G * 32;
  if (threadIdx.x < 27) p[blockIdx.x * 32 + threadIdx.x] = (float)(blockIdx.x + it + threadIdx.x) + s * 1e-9f;
  if (s == -1.0f) out[0] = s;
}
int main() {
  const int iters = 2000;
  float *part, *out; unsigned* counter;
  CHECK(hipMalloc(&part, 2 * 1024 * 32 * 4)); CHECK(hipMalloc(&out, 4)); CHECK(hipMalloc(&counter, 4));
  CHECK(hipMemset(part, 0, 2 * 1024 * 32 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int G : {16, 64, 128, 256, 512}) {
    float ms[3] = {0, 0, 0};
    for (int v = 0; v < 2; v++) {
      for (int rep = 0; rep < 2; rep++) {
        CHECK(hipMemset(counter, 0, 4));
        CHECK(hipEventRecord(e0));
        if (v == 0) hipLaunchKernelGGL(persistent<false>, dim3(G), dim3(256), 0, 0, part, counter, G, iters, out);
        else hipLaunchKernelGGL(persistent<true>, dim3(G), dim3(256), 0, 0, part, counter, G, iters, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[v], e0, e1));
      }
    }
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0));
      for (int it = 0; it < iters; it++) hipLaunchKernelGGL(one_round, dim3(G), dim3(256), 0, 0, part, G, it, out);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms[2], e0, e1));
    }
    printf("G %4d blocks: persistent round, scoped accesses %.2f us; with __threadfence %.2f us; one launch per round %.2f us\n", G, 1e3 * ms[0] / iters,
           1e3 * ms[1] / iters, 1e3 * ms[2] / iters);
  }
  return 0;
}
