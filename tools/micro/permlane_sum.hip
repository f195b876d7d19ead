// Checks the halving transpose reduction of csrc/ellc_kernels_gn.hpp (wave_sum_rows) lane by lane against a plain sum.
// build: hipcc --offload-arch=gfx950 -O3 -I egomotion_with_local_loop_closures_amd/csrc -o build/permlane_sum tools/micro/permlane_sum.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ellc_kernels_gn.hpp"
template <int NV>
__global__ void k(const float* in, float* out) {
  float v[NV];
  for (int j = 0; j < NV; j++) v[j] = in[threadIdx.x * NV + j];
  float r[ellc::WaveRows<NV>::N2];
  ellc::wave_sum_rows<NV>(v, r);
  for (int j = 0; j < ellc::WaveRows<NV>::N2; j++) out[j * 64 + threadIdx.x] = r[j];
}
template <int NV>
int run() {
  float h[64 * NV], *d, *o, ho[64 * 8];
  for (int l = 0; l < 64; l++) for (int j = 0; j < NV; j++) h[l * NV + j] = (float)(1 + l) * (j + 1);
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<NV>, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int j = 0; j < ellc::WaveRows<NV>::N2; j++)
    for (int q = 0; q < 4; q++) {
      const int val = 4 * j + 2 * (q & 1) + (q >> 1);
      const float expect = val < NV ? 2080.0f * (val + 1) : -1;
      for (int l = 0; l < 16; l++) {
        const float got = ho[j * 64 + q * 16 + l];
        if (val < NV && got != expect) { if (bad < 8) std::printf("NV %d reg %d row %d lane %d: got %g expect %g\n", NV, j, q, l, got, expect); bad++; }
      }
    }
  if (bad) { std::printf("NV %d reg 0 all lanes:", NV); for (int l = 0; l < 64; l++) std::printf(" %g", ho[l]); std::printf("\n"); }
  std::printf("NV %d: %s\n", NV, bad ? "MISMATCH" : "ok");
  return bad;
}
int main() { return run<1>() + run<6>() + run<27>(); }
