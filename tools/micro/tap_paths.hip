// What does one bilinear-tap neighbourhood (4 rows x 4 bytes at a byte-unaligned column) cost, by the way it is fetched?
//   mode 0: four unaligned global dword loads per lane (the r02 tap path)
//   mode 1: four aligned global 8-byte loads + v_alignbyte_b32
//   mode 2: window staged in LDS once per block, four unaligned ds_read_b32
//   mode 3: window staged in LDS, four ds_read2_b32 + v_alignbyte_b32
//   mode 4: arithmetic only (the VALU filler alone)
// Every lane of a wave samples column x0 + lane (a dense row, as the level-0 pass does), rows y .. y+3 of a 1280-wide image;
// `fill` fused multiply-adds per tap stand in for the pixel pass's arithmetic (132 VALU instructions per pixel in r03).
// Second table: the same taps (mode 0) beside a RECORD STREAM — 16 bytes per lane and iteration from a buffer far larger than
// the caches, as the level-0 pass reads its compact list at 1280x960 dense — fetched (a) by a global_load_dwordx4 one
// iteration ahead (the r02 loop), (b) two ahead, (c) by the block as LDS-DMA bands of 8 iterations (global_load_lds_dwordx4,
// issued one band ahead), (d) not at all. Does a cache-hit gather wait behind the misses of the stream?
// build: hipcc --offload-arch=gfx950 -O3 -o build/tap_paths tools/micro/tap_paths.hip ; run: build/tap_paths [fill] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32a1 __attribute__((aligned(1)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GL __attribute__((address_space(1)))
constexpr int W = 1280, ROWS = 24, PITCH = 1280 + 16;

template <int MODE, int FILL>
__global__ __launch_bounds__(256, 4) void taps(const uint8_t* img, float* out, int iters, int h) {
  __shared__ __attribute__((aligned(16))) uint8_t win[MODE == 2 || MODE == 3 ? ROWS * PITCH : 16];
  const int t = threadIdx.x;
  const int band = (blockIdx.x * 7) % (h - ROWS);   // this block's rows
  if (MODE == 2 || MODE == 3) {
    for (int i = t; i < ROWS * (W / 16); i += 256) {
      const int r = i / (W / 16), c = i - r * (W / 16);
      *(uint4*)(win + r * PITCH + c * 16) = *(const uint4*)(img + (size_t)(band + r) * W + c * 16);
    }
    __syncthreads();
  }
  const GL uint8_t* g = (const GL uint8_t*)img;
  float acc = 0.0f, f = 1.0f + 1e-7f * t;
  for (int it = 0; it < iters; it++) {
    const int x = 1 + ((it * 256 + t) % (W - 8)), y = (it * 256 + t) / (W - 8) % (ROWS - 4);
    uint32_t w[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (MODE == 0) w[r] = *(const GL u32a1*)(g + (unsigned)((band + y + r) * W + x));
      if (MODE == 1) {
        const unsigned o = (unsigned)((band + y + r) * W + x);
        const u32x2 v = *(const GL u32x2*)(g + (o & ~3u));
        w[r] = __builtin_amdgcn_alignbyte(v.y, v.x, o & 3u);
      }
      if (MODE == 2) w[r] = *(const u32a1*)(win + (y + r) * PITCH + x);
      if (MODE == 3) {
        const unsigned o = (unsigned)((y + r) * PITCH + x);
        const uint32_t* p = (const uint32_t*)(win + (o & ~3u));
        w[r] = __builtin_amdgcn_alignbyte(p[1], p[0], o & 3u);
      }
      if (MODE == 4) w[r] = (uint32_t)(x * 2654435761u + r);
    }
    float v = (float)(w[0] & 255u) + (float)((w[1] >> 8) & 255u) + (float)((w[2] >> 16) & 255u) + (float)(w[3] >> 24);
#pragma unroll
    for (int k = 0; k < FILL; k++) v = __builtin_fmaf(v, f, 0.25f);
    acc += v;
  }
  if (acc == 1.2345f) out[t] = acc;
}

// REC: 0 none, 1 register prefetch distance 1, 2 distance 2, 3 LDS-DMA bands
template <int REC, int FILL>
__global__ __launch_bounds__(256, 4) void taps_rec(const uint8_t* img, const u32x4* rec, float* out, int iters, int h, size_t nrec) {
  constexpr int NB = 8;   // iterations per band
  __shared__ __attribute__((aligned(16))) u32x4 band[REC == 3 ? 2 * NB * 256 : 1];
  const int t = threadIdx.x;
  const int rows = (blockIdx.x * 7) % (h - ROWS);
  const GL uint8_t* g = (const GL uint8_t*)img;
  const GL u32x4* r = (const GL u32x4*)rec + ((size_t)blockIdx.x * iters * 256) % (nrec - (size_t)iters * 256 - 4096);
  float acc = 0.0f, f = 1.0f + 1e-7f * t;
  u32x4 r0 = (u32x4)(0u), r1 = r0, r2 = r0;
  if (REC == 1 || REC == 2) r0 = r[t];
  if (REC == 2) r1 = r[256 + t];
  auto dma_band = [&](int b) {   // band b -> LDS buffer b & 1: every wave moves its own 64 records of each iteration
    if (b * NB >= iters) return;
    for (int k = 0; k < NB; k++) {
      const GL u32x4* src = r + (size_t)(b * NB + k) * 256 + t;
      u32x4* dst = band + ((b & 1) * NB + k) * 256 + (t & ~63);   // wave-uniform LDS base; lane l lands at base + 16 l
      __builtin_amdgcn_global_load_lds((const GL void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  if (REC == 3) { dma_band(0); }
  for (int it = 0; it < iters; it++) {
    if (REC == 3 && it % NB == 0) {
      dma_band(it / NB + 1);                                   // next band, a whole band ahead
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");   // the current band has landed (the NB youngest = the next band)
      __syncthreads();
    }
    u32x4 rc;
    if (REC == 1) { rc = r0; r0 = r[(size_t)min(it + 1, iters - 1) * 256 + t]; }
    if (REC == 2) { rc = r0; r0 = r1; r1 = r[(size_t)min(it + 2, iters - 1) * 256 + t]; }
    if (REC == 3) rc = band[((it / NB & 1) * NB + it % NB) * 256 + t];
    if (REC == 0) rc = (u32x4){(unsigned)it, 1u, 2u, 3u};
    const int x = 1 + ((it * 256 + t + (rc.x & 1)) % (W - 8)), y = (it * 256 + t) / (W - 8) % (ROWS - 4);
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) w[q] = *(const GL u32a1*)(g + (unsigned)((rows + y + q) * W + x));
    float v = (float)(w[0] & 255u) + (float)((w[1] >> 8) & 255u) + (float)((w[2] >> 16) & 255u) + (float)(w[3] >> 24) + (float)(rc.y ^ rc.z ^ rc.w);
#pragma unroll
    for (int k = 0; k < FILL; k++) v = __builtin_fmaf(v, f, 0.25f);
    acc += v;
  }
  if (acc == 1.2345f) out[t] = acc + (float)r2.x;
}

// LDS-staged taps (window of ROWS image rows per block, restaged every `restage` iterations through a barrier pair) + the record
// stream by register prefetch DIST iterations ahead: the computing waves' only vector-memory loads are the records.
template <int DIST, int FILL>
__global__ __launch_bounds__(256, 4) void taps_lds_rec(const uint8_t* img, const u32x4* rec, float* out, int iters, int h, size_t nrec, int restage) {
  __shared__ __attribute__((aligned(16))) uint8_t win[ROWS * PITCH];
  const int t = threadIdx.x;
  int rows = (blockIdx.x * 7) % (h - 2 * ROWS);
  const GL u32x4* r = (const GL u32x4*)rec + ((size_t)blockIdx.x * iters * 256) % (nrec - (size_t)iters * 256 - 4096);
  float acc = 0.0f, f = 1.0f + 1e-7f * t;
  u32x4 q[DIST];
#pragma unroll
  for (int k = 0; k < DIST; k++) q[k] = r[(size_t)min(k, iters - 1) * 256 + t];
  for (int it0 = 0; it0 < iters; it0 += restage) {
    if (it0) __syncthreads();
    for (int i = t; i < ROWS * (W / 16); i += 256) {
      const int rr = i / (W / 16), c = i - rr * (W / 16);
      *(uint4*)(win + rr * PITCH + c * 16) = *(const uint4*)(img + (size_t)(rows + rr + (it0 & 7)) * W + c * 16);
    }
    __syncthreads();
    for (int it = it0; it < min(iters, it0 + restage); it += DIST) {
#pragma unroll
      for (int k = 0; k < DIST; k++) {
        if (it + k >= iters) break;
        const u32x4 rc = q[k];
        q[k] = r[(size_t)min(it + k + DIST, iters - 1) * 256 + t];
        const int x = 1 + (((it + k) * 256 + t + (rc.x & 1)) % (W - 8)), y = ((it + k) * 256 + t) / (W - 8) % (ROWS - 4);
        uint32_t w[4];
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
          const unsigned o = (unsigned)((y + qq) * PITCH + x);
          const uint32_t* p = (const uint32_t*)(win + (o & ~3u));
          w[qq] = __builtin_amdgcn_alignbyte(p[1], p[0], o & 3u);
        }
        float v = (float)(w[0] & 255u) + (float)((w[1] >> 8) & 255u) + (float)((w[2] >> 16) & 255u) + (float)(w[3] >> 24) + (float)(rc.y ^ rc.z ^ rc.w);
#pragma unroll
        for (int kk = 0; kk < FILL; kk++) v = __builtin_fmaf(v, f, 0.25f);
        acc += v;
      }
    }
  }
  if (acc == 1.2345f) out[t] = acc;
}

template <int DIST, int FILL>
static int run_lds_rec(const uint8_t* img, const u32x4* rec, size_t nrec, float* out, int iters, int h, int blocks, int restage) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((taps_lds_rec<DIST, FILL>), dim3(blocks), dim3(256), 0, 0, img, rec, out, iters, h, nrec, restage);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((taps_lds_rec<DIST, FILL>), dim3(blocks), dim3(256), 0, 0, img, rec, out, iters, h, nrec, restage);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double px = 5.0 * blocks * 256.0 * iters;
  std::printf("LDS-staged taps (restaged every %2d it.) + records %d ahead  fill %3d, %4d blocks: %8.1f us per launch, %6.2f Gpx/s, record stream %5.2f TB/s\n",
              restage, DIST, FILL, blocks, 1e3 * ms / 5, px / (ms * 1e6), px * 16 / (ms * 1e-3) / 1e12);
  return 0;
}

// REC 4: a fifth wave of the block does nothing but stream the records into LDS (LDS-DMA, one band of NB iterations ahead, two
// band buffers); the four computing waves only ever issue tap loads, so their vmcnt waits never cover a record load.
template <int FILL, int NB>
__global__ __launch_bounds__(320, 1) void taps_loader(const uint8_t* img, const u32x4* rec, float* out, int iters, int h, size_t nrec) {
  __shared__ __attribute__((aligned(16))) u32x4 band[2 * NB * 256];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int rows = (blockIdx.x * 7) % (h - ROWS);
  const GL uint8_t* g = (const GL uint8_t*)img;
  const GL u32x4* r = (const GL u32x4*)rec + ((size_t)blockIdx.x * iters * 256) % (nrec - (size_t)iters * 256 - 4096);
  const int nbands = (iters + NB - 1) / NB;
  float acc = 0.0f, f = 1.0f + 1e-7f * t;
  if (wave == 4) {   // loader
    for (int b = 0; b <= nbands; b++) {
      if (b < nbands) {
        for (int k = 0; k < NB * 4; k++) {   // NB iterations x 4 compute waves, 1 KiB per instruction
          const GL u32x4* src = r + (size_t)b * NB * 256 + (size_t)k * 64 + lane;
          u32x4* dst = band + (b & 1) * NB * 256 + k * 64;
          __builtin_amdgcn_global_load_lds((const GL void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      }
      // band b - 1 ... wait: before the computing waves start band b, band b must have landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // A: band b is in LDS (and the computing waves have finished band b - 1: its buffer (b + 1) & 1 is free)
    }
    return;
  }
  for (int b = 0; b < nbands; b++) {
    __syncthreads();     // A
    for (int it = b * NB; it < min(iters, (b + 1) * NB); it++) {
      const u32x4 rc = band[((b & 1) * NB + (it - b * NB)) * 256 + t];
      const int x = 1 + ((it * 256 + t + (rc.x & 1)) % (W - 8)), y = (it * 256 + t) / (W - 8) % (ROWS - 4);
      uint32_t w[4];
#pragma unroll
      for (int q = 0; q < 4; q++) w[q] = *(const GL u32a1*)(g + (unsigned)((rows + y + q) * W + x));
      float v = (float)(w[0] & 255u) + (float)((w[1] >> 8) & 255u) + (float)((w[2] >> 16) & 255u) + (float)(w[3] >> 24) + (float)(rc.y ^ rc.z ^ rc.w);
#pragma unroll
      for (int k = 0; k < FILL; k++) v = __builtin_fmaf(v, f, 0.25f);
      acc += v;
    }
  }
  __syncthreads();   // the loader's last A
  if (acc == 1.2345f) out[t] = acc;
}

template <int FILL, int NB>
static int run_loader(const uint8_t* img, const u32x4* rec, size_t nrec, float* out, int iters, int h, int blocks, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((taps_loader<FILL, NB>), dim3(blocks), dim3(320), 0, 0, img, rec, out, iters, h, nrec);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((taps_loader<FILL, NB>), dim3(blocks), dim3(320), 0, 0, img, rec, out, iters, h, nrec);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double px = 5.0 * blocks * 256.0 * iters;
  std::printf("%-44s NB %2d fill %3d, %4d blocks: %8.1f us per launch, %6.2f Gpx/s, record stream %5.2f TB/s\n", name, NB, FILL, blocks, 1e3 * ms / 5, px / (ms * 1e6),
              px * 16 / (ms * 1e-3) / 1e12);
  return 0;
}

template <int REC, int FILL>
static int run_rec(const uint8_t* img, const u32x4* rec, size_t nrec, float* out, int iters, int h, int blocks, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((taps_rec<REC, FILL>), dim3(blocks), dim3(256), 0, 0, img, rec, out, iters, h, nrec);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((taps_rec<REC, FILL>), dim3(blocks), dim3(256), 0, 0, img, rec, out, iters, h, nrec);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double px = 5.0 * blocks * 256.0 * iters;
  std::printf("%-52s fill %3d, %4d blocks: %8.1f us per launch, %6.2f Gpx/s, record stream %5.2f TB/s\n", name, FILL, blocks, 1e3 * ms / 5, px / (ms * 1e6),
              REC ? px * 16 / (ms * 1e-3) / 1e12 : 0.0);
  return 0;
}

template <int MODE, int FILL>
static int run(const uint8_t* img, float* out, int iters, int h, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 1024;
  hipLaunchKernelGGL((taps<MODE, FILL>), dim3(blocks), dim3(256), 0, 0, img, out, iters, h);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((taps<MODE, FILL>), dim3(blocks), dim3(256), 0, 0, img, out, iters, h);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double px = 5.0 * blocks * 256.0 * iters;
  std::printf("%-44s fill %3d: %8.1f us per launch, %6.2f Gpx/s, %5.1f ns per wave-iteration per CU\n", name, FILL, 1e3 * ms / 5, px / (ms * 1e6),
              ms * 1e6 / 5 / ((double)blocks * 4 * iters / 256.0));
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 64, h = 960;
  uint8_t* img; float* out;
  CK(hipMalloc(&img, (size_t)W * h + 64)); CK(hipMemset(img, 7, (size_t)W * h + 64)); CK(hipMalloc(&out, 4096));
#define ALL(F) \
  run<0, F>(img, out, iters, h, "global, 4 unaligned dword loads"); \
  run<1, F>(img, out, iters, h, "global, 4 aligned 8-byte loads + alignbyte"); \
  run<2, F>(img, out, iters, h, "LDS window, 4 unaligned ds_read_b32"); \
  run<3, F>(img, out, iters, h, "LDS window, 4 ds_read2_b32 + alignbyte"); \
  run<4, F>(img, out, iters, h, "no loads");
  ALL(120)
  const size_t nrec = (size_t)96 << 20;   // 1.5 GiB of records
  u32x4* rec;
  CK(hipMalloc(&rec, nrec * 16)); CK(hipMemset(rec, 1, nrec * 16));
  const int it2 = 600;   // iterations per block, as a level-0 launch at 1280x960 over 64 alignments has them (153 600 px / 256)
#define RECS(F, B) \
  run_rec<0, F>(img, rec, nrec, out, it2, h, B, "taps only"); \
  run_rec<1, F>(img, rec, nrec, out, it2, h, B, "taps + records, register prefetch 1 ahead"); \
  run_rec<2, F>(img, rec, nrec, out, it2, h, B, "taps + records, register prefetch 2 ahead"); \
  run_rec<3, F>(img, rec, nrec, out, it2, h, B, "taps + records by LDS-DMA, bands of 8 iterations");
  RECS(120, 512) RECS(120, 1024)
  run_loader<120, 4>(img, rec, nrec, out, it2, h, 512, "taps + records by a loader wave (LDS-DMA)");
  run_loader<120, 8>(img, rec, nrec, out, it2, h, 512, "taps + records by a loader wave (LDS-DMA)");
  run_loader<120, 4>(img, rec, nrec, out, it2, h, 1024, "taps + records by a loader wave (LDS-DMA)");
  run_loader<120, 8>(img, rec, nrec, out, it2, h, 1024, "taps + records by a loader wave (LDS-DMA)");
  run_loader<120, 2>(img, rec, nrec, out, it2, h, 1024, "taps + records by a loader wave (LDS-DMA)");
  for (int blocks = 512; blocks <= 1024; blocks += 512) {
    run_lds_rec<1, 120>(img, rec, nrec, out, it2, h, blocks, 24);
    run_lds_rec<2, 120>(img, rec, nrec, out, it2, h, blocks, 24);
    run_lds_rec<3, 120>(img, rec, nrec, out, it2, h, blocks, 24);
    run_lds_rec<4, 120>(img, rec, nrec, out, it2, h, blocks, 24);
    run_lds_rec<2, 120>(img, rec, nrec, out, it2, h, blocks, 8);
  }
  return 0;
}
