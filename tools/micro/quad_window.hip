// r06: what the taps of 256 neighbouring pixels of a dense level cost the CU's vector cache by the way a wave asks for them.
// One lane per pixel (4 wave-steps for 256 pixels: the r05 kernels) against one lane per FOUR adjacent pixels (one wave-step: a
// window per tap row that serves all four neighbourhoods). All L2-resident (1.2 MB image), no arithmetic: the time is the memory
// pipeline's. 256 CUs x 16 waves. Same harness as gather_rate.hip; the figure to compare is "cycles per 256 pixels per CU".
// build: hipcc --offload-arch=gfx950 -O3 -o build/quad_window tools/micro/quad_window.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
typedef uint32_t u32a1 __attribute__((aligned(1)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 u32x2a1 __attribute__((aligned(1)));
typedef u32x3 u32x3a1 __attribute__((aligned(1)));
typedef u32x2 u32x2a4 __attribute__((aligned(4)));
typedef u32x3 u32x3a4 __attribute__((aligned(4)));
typedef u32x4 u32x4a4 __attribute__((aligned(4)));
enum { P_PIX_DWORD, P_QUAD_X2_U, P_QUAD_X2_U5, P_QUAD_X3_U, P_QUAD_X3_U5, P_QUAD_X2_A, P_QUAD_X3_A, P_QUAD_X3_A5, P_QUAD_X4_A, P_COLPACK_X4, P_PIX_DWORD_JIT, P_QUAD_X3_JIT, P_N };
template <int PAT>
__global__ __launch_bounds__(1024) void k(const uint8_t* __restrict__ img, int pitch, int rows, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  unsigned y = (wave * 7) % (rows - 8) + 2;
  unsigned xb = 3;   // first pixel of the wave's 256
  const unsigned jit = (lane * 2654435761u >> 28) & 1u;   // a per-lane row jitter of 0 / 1 (depth noise moves a point across a row boundary)
  for (int it = 0; it < ITER; it++) {
    if (PAT == P_PIX_DWORD || PAT == P_PIX_DWORD_JIT) {   // the product's taps: lane = pixel, 4 rows of one unaligned dword, 4 wave-steps
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned o = (y + (PAT == P_PIX_DWORD_JIT ? jit : 0u)) * pitch + xb + 64 * q + lane;
        acc ^= *(const u32a1*)(img + o - 1 - pitch) ^ *(const u32a1*)(img + o - 1) ^ *(const u32a1*)(img + o - 1 + pitch) ^ *(const u32a1*)(img + o - 1 + 2 * pitch);
      }
    } else if (PAT == P_COLPACK_X4) {   // column-packed image (one dword per pixel = 4 rows): ONE 16-byte load per pixel at a 4-aligned address
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const size_t o = ((size_t)y * pitch + xb + 64 * q + lane) * 4 % ((size_t)pitch * rows * 4);
        const u32x4 a = *(const u32x4a4*)(img + o);
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
      }
    } else {   // lane = four adjacent pixels: one window per row from column x0 - 1 of the first
      const unsigned o = (y + ((PAT == P_QUAD_X3_JIT) ? jit : 0u)) * pitch + xb + 4 * lane + (it & 3) - 1;   // (the windows start at any byte)
      const int nrows = (PAT == P_QUAD_X2_U5 || PAT == P_QUAD_X3_U5 || PAT == P_QUAD_X3_A5) ? 5 : 4;
#pragma unroll
      for (int r = 0; r < nrows; r++) {
        const uint8_t* p = img + o + (r - 1) * pitch;
        if (PAT == P_QUAD_X2_U || PAT == P_QUAD_X2_U5) { const u32x2 a = *(const u32x2a1*)p; acc ^= a.x ^ a.y; }
        else if (PAT == P_QUAD_X3_U || PAT == P_QUAD_X3_U5 || PAT == P_QUAD_X3_JIT) { const u32x3 a = *(const u32x3a1*)p; acc ^= a.x ^ a.y ^ a.z; }
        else if (PAT == P_QUAD_X2_A) { const u32x2 a = *(const u32x2a4*)((uintptr_t)p & ~(uintptr_t)3); acc ^= a.x ^ a.y; }
        else if (PAT == P_QUAD_X3_A || PAT == P_QUAD_X3_A5) { const u32x3 a = *(const u32x3a4*)((uintptr_t)p & ~(uintptr_t)3); acc ^= a.x ^ a.y ^ a.z; }
        else if (PAT == P_QUAD_X4_A) { const u32x4 a = *(const u32x4a4*)((uintptr_t)p & ~(uintptr_t)3); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
      }
    }
    xb += 256;
    if (xb + 256 + 16 >= (unsigned)pitch) { xb = 3; y += 1; if (y >= (unsigned)rows - 6) y = 2; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int PAT>
void run(const char* name, const uint8_t* img, int pitch, int rows, uint32_t* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int waves_per_cu = 16, cus = 256;
  dim3 grid(cus * waves_per_cu / 4), block(256);
  hipLaunchKernelGGL(k<PAT>, grid, block, 0, 0, img, pitch, rows, out);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<PAT>, grid, block, 0, 0, img, pitch, rows, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double steps_per_cu = (double)waves_per_cu * ITER;
  const double cyc = best * 1e-3 * 2.35e9;   // ~2.35 GHz under load (tools/micro/valu_rates.hip)
  std::printf("%-64s %8.1f us  %7.1f cycles per 256 pixels per CU\n", name, best * 1e3, cyc / steps_per_cu);
}
int main() {
  const int pitch = 1280, rows = 960;
  uint8_t* img; uint32_t* out;
  (void)hipMalloc(&img, 4 * ((size_t)pitch * rows + 1024) + 4096); (void)hipMalloc(&out, 64);
  (void)hipMemset(img, 0x5a, 4 * ((size_t)pitch * rows + 1024) + 4096);
  run<P_PIX_DWORD>("lane = pixel: 4 x (4 rows, unaligned dword)", img, pitch, rows, out);
  run<P_PIX_DWORD_JIT>("lane = pixel: the same, lanes on two rows at random", img, pitch, rows, out);
  run<P_QUAD_X2_U>("lane = 4 pixels: 4 rows, 8 bytes at any byte", img, pitch, rows, out);
  run<P_QUAD_X2_U5>("lane = 4 pixels: 5 rows, 8 bytes at any byte", img, pitch, rows, out);
  run<P_QUAD_X3_U>("lane = 4 pixels: 4 rows, 12 bytes at any byte", img, pitch, rows, out);
  run<P_QUAD_X3_U5>("lane = 4 pixels: 5 rows, 12 bytes at any byte", img, pitch, rows, out);
  run<P_QUAD_X3_JIT>("lane = 4 pixels: 4 rows, 12 bytes, lanes on two rows at random", img, pitch, rows, out);
  run<P_QUAD_X2_A>("lane = 4 pixels: 4 rows, 8 bytes 4-aligned", img, pitch, rows, out);
  run<P_QUAD_X3_A>("lane = 4 pixels: 4 rows, 12 bytes 4-aligned", img, pitch, rows, out);
  run<P_QUAD_X3_A5>("lane = 4 pixels: 5 rows, 12 bytes 4-aligned", img, pitch, rows, out);
  run<P_QUAD_X4_A>("lane = 4 pixels: 4 rows, 16 bytes 4-aligned", img, pitch, rows, out);
  run<P_COLPACK_X4>("lane = pixel: 4 x (one 16-byte load, column-packed image)", img, pitch, rows, out);
  return 0;
}
