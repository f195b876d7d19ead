// How many dependent small kernel launches per second does the device take from 1..4 streams? Each stream replays a
// linear hipGraph of K launches (grid G blocks x 256 threads, a few hundred ns of work each), R times. If the aggregate
// rate stops growing with the number of streams, a pipeline of short launches is bound by packet processing, not by CUs.
// build: hipcc --offload-arch=gfx950 -O2 -o build/dispatch_rate tools/micro/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float* p, int spin) {
  float v = p[blockIdx.x];
  for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
  if (v == 1.2345f) p[blockIdx.x] = v;
}
int main(int argc, char** argv) {
  const int K = argc > 1 ? std::atoi(argv[1]) : 36, G = argc > 2 ? std::atoi(argv[2]) : 224, R = argc > 3 ? std::atoi(argv[3]) : 100;
  const int spin = argc > 4 ? std::atoi(argv[4]) : 200;
  const int SMAX = argc > 5 ? std::atoi(argv[5]) : 4;
  for (int S = 1; S <= SMAX; S++) {
    std::vector<hipStream_t> st(S);
    std::vector<hipGraphExec_t> ex(S);
    std::vector<float*> buf(S);
    for (int s = 0; s < S; s++) {
      CK(hipStreamCreate(&st[s]));
      CK(hipMalloc((void**)&buf[s], 65536 * 4));
      CK(hipMemsetAsync(buf[s], 0, 65536 * 4, st[s]));
      hipGraph_t g;
      CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < K; k++) hipLaunchKernelGGL(tiny, dim3(G), dim3(256), 0, st[s], buf[s], spin);
      CK(hipStreamEndCapture(st[s], &g));
      CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
      CK(hipGraphDestroy(g));
      CK(hipGraphLaunch(ex[s], st[s]));
    }
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; r++)
      for (int s = 0; s < S; s++) CK(hipGraphLaunch(ex[s], st[s]));
    CK(hipDeviceSynchronize());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double n = (double)S * R * K;
    std::printf("streams %d: %.0f launches in %.3f ms -> %.2f us per launch per stream, aggregate %.2f M launches/s\n", S, n, dt * 1e3,
                dt * 1e6 / ((double)R * K), n / dt / 1e6);
    for (int s = 0; s < S; s++) { (void)hipGraphExecDestroy(ex[s]); (void)hipFree(buf[s]); (void)hipStreamDestroy(st[s]); }
  }
  return 0;
}
