// Micro-benchmark of the resident-A K loop in isolation: per 16-deep step 12 MFMAs,
// 4 x ds_read_b128 (A fragments, one step ahead) and 4 x global_load_dwordx4 (W fragments,
// RING steps ahead) on synthetic data.  Reports core ticks per MFMA per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_loop.hip -o mfma_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int RING, bool LDSR, bool GLD>
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ w, float* out,
                                            unsigned long long* ticks, int steps, int wstride) {
  extern __shared__ uint4 lds[];   // 64 KB of "A panel"
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_uint4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // every workgroup walks the SAME 1 MB of W (4 KB per step per wave, offset by wave): L2-resident
  const uint4* wl = w + (size_t)wave * wstride + lane;
  uint4 ring[RING][4];
  for (int d = 0; d < RING; ++d)
    for (int q = 0; q < 4; ++q) ring[d][q] = wl[(d * 4 + q) * 64];
  uint4 afn[4];
  for (int q = 0; q < 4; ++q) afn[q] = lds[lane + 64 * q];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s0 = 0; s0 < steps; s0 += RING) {
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      uint4 af[4], bf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        af[q] = afn[q];
        if (LDSR) afn[q] = lds[((s0 + d + 1) & 15) * 256 + lane + 64 * q];
        bf[q] = ring[d][q];
        if (GLD) ring[d][q] = wl[(((s0 + d + RING) & 255) * 4 + q) * 64];
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int pa = t == 1 ? 1 : 0, pb = t == 0 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(f16x8, af[2 * i + pa]), __builtin_bit_cast(f16x8, bf[2 * j + pb]),
                acc[i][j], 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int RING, bool LDSR, bool GLD>
void run(const char* name, int blocks) {
  uint4* w;
  float* out;
  unsigned long long* ticks;
  const int wstride = 1024 * 64 + 64;   // uint4 per wave
  hipMalloc(&w, sizeof(uint4) * (size_t)4 * wstride);
  hipMemset(w, 0x11, sizeof(uint4) * (size_t)4 * wstride);
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&ticks, sizeof(unsigned long long) * blocks);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<RING, LDSR, GLD>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int steps = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<RING, LDSR, GLD><<<blocks, 256, 65536>>>(w, out, ticks, 64, wstride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<RING, LDSR, GLD><<<blocks, 256, 65536>>>(w, out, ticks, steps, wstride);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h;
  hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  const double mf = steps * 12.0;
  const int wps = blocks >= 512 ? 2 : 1;
  printf("%-34s blocks=%4d: %.1f ticks per MFMA per wave, %.1f per SIMD (%d waves/SIMD), %.0f TFLOP/s\n", name,
         blocks, (double)h / mf, (double)h / mf / wps, wps, mf * blocks * 4 * 32768.0 / ms / 1e9);
  hipFree(w);
  hipFree(out);
  hipFree(ticks);
}

int main() {
  run<4, false, false>("MFMA only", 256);
  run<4, false, false>("MFMA only", 512);
  run<4, true, false>("+ LDS fragment reads", 512);
  run<4, false, true>("+ global W loads (ring 4)", 512);
  run<4, true, true>("+ both (ring 4)", 256);
  run<4, true, true>("+ both (ring 4)", 512);
  run<8, true, true>("+ both (ring 8)", 512);
  run<2, true, true>("+ both (ring 2)", 512);
  return 0;
}
