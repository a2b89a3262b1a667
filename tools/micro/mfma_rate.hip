// Micro-benchmark: issue rate of the 32x32x16 bf16 / f16 MFMAs on gfx950, and the
// s_memtime tick rate.  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i);
    b[i] = (__bf16)(seed - i);
    ha[i] = (_Float16)(seed + threadIdx.x * 0.001f + i);
    hb[i] = (_Float16)(seed - i);
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND, int NACC>
void run(const char* name, int blocks, int threads, int iters) {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&ticks, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KIND, NACC><<<blocks, threads>>>(out, ticks, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND, NACC><<<blocks, threads>>>(out, ticks, iters, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h;
  hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * NACC;                    // per wave
  const double waves = (double)blocks * threads / 64;
  const double flops = n_mfma * waves * 32768.0;
  printf("%-28s blocks=%d thr=%d: %.3f ms  %.1f TFLOP/s  ticks/MFMA(per wave)=%.1f  tick rate %.1f MHz\n", name, blocks,
         threads, ms, flops / ms / 1e9, (double)h / n_mfma, (double)h / (ms * 1e3));
  hipFree(out);
  hipFree(ticks);
}

int main() {
  const int it = 20000;
  run<0, 4>("bf16 1 wave/SIMD (1 CU)", 1, 256, it);
  run<1, 4>("f16  1 wave/SIMD (1 CU)", 1, 256, it);
  run<0, 4>("bf16 1 wave/SIMD all CUs", 256, 256, it);
  run<1, 4>("f16  1 wave/SIMD all CUs", 256, 256, it);
  run<0, 4>("bf16 3 waves/SIMD all CUs", 768, 256, it);
  run<1, 4>("f16  3 waves/SIMD all CUs", 768, 256, it);
  run<1, 1>("f16 dependent chain 1 CU", 1, 256, it);
  return 0;
}
