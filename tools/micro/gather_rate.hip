// What a vector-memory gather costs the CU's vector cache (TCP) by its address pattern (r04). Every wave loops over ITER steps; a step
// issues NL loads whose lane addresses follow one of the patterns below (all L2-resident: the image is 1.2 MB) and folds the
// results into one register. No arithmetic to speak of: the time is the memory pipeline's. 256 CUs x 16 waves.
// build: hipcc --offload-arch=gfx950 -O3 -o build/gather_rate tools/micro/gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
typedef uint32_t u32a1 __attribute__((aligned(1)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
enum { P_TAP_UNALIGNED, P_TAP_ALIGNED, P_ROW_COALESCED, P_TAP_BYTE, P_TAP_X2, P_REC_X4, P_TAP_SCATTER, P_TAP_1ROW, P_TAP_2ROWS, P_SAMEADDR, P_TAP_STRIDE2, P_TAP_4COPIES, P_SCATTER_4COPIES, P_N };
template <int PAT>
__global__ __launch_bounds__(1024) void k(const uint8_t* __restrict__ img, int pitch, int rows, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  const unsigned copy_stride = (unsigned)pitch * (unsigned)rows + 1024u;   // (the four shifted copies of P_*_4COPIES)
  // a wave walks along a row band: 64 neighbouring pixels per step, like the taps of 64 neighbouring pixels of a dense level
  unsigned y = (wave * 7) % (rows - 8) + 2;
  unsigned x = 3 + lane;
  for (int it = 0; it < ITER; it++) {
    const unsigned o = y * pitch + x;
    if (PAT == P_TAP_UNALIGNED) {   // 4 rows, one dword per row at byte address o - 1 (the product's taps)
      acc ^= *(const u32a1*)(img + o - 1 - pitch) ^ *(const u32a1*)(img + o - 1) ^ *(const u32a1*)(img + o - 1 + pitch) ^ *(const u32a1*)(img + o - 1 + 2 * pitch);
    } else if (PAT == P_TAP_ALIGNED) {   // the same, addresses rounded down to 4 (lanes share dwords in fours)
      const unsigned oa = (o - 1) & ~3u;
      acc ^= *(const uint32_t*)(img + oa - pitch) ^ *(const uint32_t*)(img + oa) ^ *(const uint32_t*)(img + oa + pitch) ^ *(const uint32_t*)(img + oa + 2 * pitch);
    } else if (PAT == P_ROW_COALESCED) {   // 4 rows, lane k reads dword k of the row segment (fully coalesced: 256 B per row)
      const unsigned oa = y * pitch + ((x - lane) & ~3u) + 4 * lane;
      acc ^= *(const uint32_t*)(img + oa - pitch) ^ *(const uint32_t*)(img + oa) ^ *(const uint32_t*)(img + oa + pitch) ^ *(const uint32_t*)(img + oa + 2 * pitch);
    } else if (PAT == P_TAP_BYTE) {   // 4 byte loads (one row's worth of bytes; what 16 byte loads would cost is 4 x this)
      acc ^= img[o - 1 - pitch] ^ img[o - 1] ^ img[o - 1 + pitch] ^ img[o - 1 + 2 * pitch];
    } else if (PAT == P_TAP_X2) {   // 4 rows, 8 bytes per row, 4-aligned
      const unsigned oa = (o - 1) & ~3u;
      const u32x2 a = *(const u32x2*)(img + oa - pitch), b = *(const u32x2*)(img + oa), c = *(const u32x2*)(img + oa + pitch), d = *(const u32x2*)(img + oa + 2 * pitch);
      acc ^= a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y;
    } else if (PAT == P_REC_X4) {   // one 16-byte record per lane, consecutive (the record stream's shape; from L2 here)
      const u32x4 a = *(const u32x4*)(img + (((size_t)wave * ITER + it) * 1024 + lane * 16) % (size_t)(pitch * (rows - 1)));
      acc ^= a.x ^ a.y ^ a.z ^ a.w;
    } else if (PAT == P_TAP_SCATTER) {   // 4 rows, lanes 5 pixels apart and on 4 different rows (a semi-dense wave)
      const unsigned os = (y + (lane & 3)) * pitch + 3 + lane * 5 + (x - 3 - lane);
      acc ^= *(const u32a1*)(img + os - 1 - pitch) ^ *(const u32a1*)(img + os - 1) ^ *(const u32a1*)(img + os - 1 + pitch) ^ *(const u32a1*)(img + os - 1 + 2 * pitch);
    } else if (PAT == P_TAP_1ROW) {
      acc ^= *(const u32a1*)(img + o - 1);
    } else if (PAT == P_TAP_2ROWS) {
      acc ^= *(const u32a1*)(img + o - 1) ^ *(const u32a1*)(img + o - 1 + pitch);
    } else if (PAT == P_SAMEADDR) {   // every lane the same dword, 4 rows
      const unsigned oa = y * pitch + ((x - lane) & ~3u);
      acc ^= *(const uint32_t*)(img + oa - pitch) ^ *(const uint32_t*)(img + oa) ^ *(const uint32_t*)(img + oa + pitch) ^ *(const uint32_t*)(img + oa + 2 * pitch);
    } else if (PAT == P_TAP_4COPIES || PAT == P_SCATTER_4COPIES) {
      // four copies of the image shifted by 0..3 bytes: the window [x - 1, x + 2] of a row is an ALIGNED dword of copy (x - 1) & 3
      // (r04 idea; lanes 1 px apart / the semi-dense pattern of P_TAP_SCATTER)
      const unsigned os = (PAT == P_TAP_4COPIES) ? o : (y + (lane & 3)) * pitch + 3 + lane * 5 + (x - 3 - lane);
      const unsigned s0 = os - 1, c = s0 & 3u, oa = (s0 & ~3u) + c * copy_stride;
      acc ^= *(const uint32_t*)(img + oa - pitch) ^ *(const uint32_t*)(img + oa) ^ *(const uint32_t*)(img + oa + pitch) ^ *(const uint32_t*)(img + oa + 2 * pitch);
    } else if (PAT == P_TAP_STRIDE2) {   // lanes 2 pixels apart
      const unsigned os = y * pitch + 3 + lane * 2 + (x - 3 - lane);
      acc ^= *(const u32a1*)(img + os - 1 - pitch) ^ *(const u32a1*)(img + os - 1) ^ *(const u32a1*)(img + os - 1 + pitch) ^ *(const u32a1*)(img + os - 1 + 2 * pitch);
    }
    x += 64;
    if (x + 64 * 5 + 8 >= (unsigned)pitch) { x = 3 + lane; y += 1; if (y >= (unsigned)rows - 6) y = 2; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int PAT>
void run(const char* name, int nloads, const uint8_t* img, int pitch, int rows, uint32_t* out) {
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
  std::printf("%-44s %8.1f us  %6.1f cycles per wave-step per CU  %6.1f per load instruction  (%.2f lane-accesses per cycle)\n", name, best * 1e3,
              cyc / steps_per_cu, cyc / steps_per_cu / nloads, 64.0 * nloads * steps_per_cu / cyc);
}
int main() {
  const int pitch = 1280, rows = 960;
  uint8_t* img; uint32_t* out;
  (void)hipMalloc(&img, 4 * ((size_t)pitch * rows + 1024) + 4096); (void)hipMalloc(&out, 64);
  (void)hipMemset(img, 0x5a, 4 * ((size_t)pitch * rows + 1024) + 4096);
  run<P_TAP_UNALIGNED>("4 rows, unaligned dword, lanes 1 px apart", 4, img, pitch, rows, out);
  run<P_TAP_ALIGNED>("4 rows, dword rounded to 4 (shared by 4)", 4, img, pitch, rows, out);
  run<P_ROW_COALESCED>("4 rows, lane k = dword k (coalesced)", 4, img, pitch, rows, out);
  run<P_SAMEADDR>("4 rows, every lane the same dword", 4, img, pitch, rows, out);
  run<P_TAP_BYTE>("4 rows, one byte, lanes 1 px apart", 4, img, pitch, rows, out);
  run<P_TAP_X2>("4 rows, 8 bytes 4-aligned", 4, img, pitch, rows, out);
  run<P_TAP_STRIDE2>("4 rows, unaligned dword, lanes 2 px apart", 4, img, pitch, rows, out);
  run<P_TAP_SCATTER>("4 rows, unaligned dword, 5 px apart, 4 rows", 4, img, pitch, rows, out);
  run<P_TAP_4COPIES>("4 rows, aligned dword of 4 shifted copies, 1 px apart", 4, img, pitch, rows, out);
  run<P_SCATTER_4COPIES>("4 rows, aligned dword of 4 shifted copies, scattered", 4, img, pitch, rows, out);
  run<P_TAP_1ROW>("1 row, unaligned dword", 1, img, pitch, rows, out);
  run<P_TAP_2ROWS>("2 rows, unaligned dword", 2, img, pitch, rows, out);
  run<P_REC_X4>("16-byte records, consecutive lanes", 1, img, pitch, rows, out);
  return 0;
}
