// Micro-benchmark: can a wave's own VALU / SALU / LDS instructions issue in the shadow
// of its MFMAs on gfx950?  Build: hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap
// Each variant runs `iters` rounds of { 4 independent MFMAs, each followed by NV
// filler instructions } on one wave per SIMD of one CU and reports core-clock ticks per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV>   // KIND 0: v_add_f32, 1: s_add_u32, 2: ds_read_b32, 3: v_cvt chain (dependent)
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters, float seed) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = seed;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) {
    ha[i] = (_Float16)(seed + threadIdx.x * 0.001f + i);
    hb[i] = (_Float16)(seed - i);
  }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = seed + i;
  unsigned int sreg = 1;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v & 7]) : "v"(seed));
        if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
        if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(f[v & 7]) : "v"((threadIdx.x & 255) * 4));
        if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(seed));
      }
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)sreg;
  for (int i = 0; i < 8; ++i) s += f[i];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND, int NV>
void run(const char* name, int threads) {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, sizeof(float) * 1024);
  hipMalloc(&ticks, 64);
  const int iters = 5000;
  k<KIND, NV><<<1, threads>>>(out, ticks, 10, 1.0f);
  k<KIND, NV><<<1, threads>>>(out, ticks, iters, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-10s fillers/MFMA=%d  waves/SIMD=%d: %.1f ticks per MFMA (per wave)\n", name, NV, threads / 256,
         (double)h / (iters * 4.0));
  hipFree(out);
  hipFree(ticks);
}

int main() {
  run<0, 0>("none", 256);
  run<0, 2>("v_add", 256);
  run<0, 4>("v_add", 256);
  run<0, 6>("v_add", 256);
  run<0, 8>("v_add", 256);
  run<0, 12>("v_add", 256);
  run<3, 6>("v_add dep", 256);
  run<1, 4>("s_add", 256);
  run<1, 8>("s_add", 256);
  run<1, 16>("s_add", 256);
  run<2, 1>("ds_read", 256);
  run<2, 2>("ds_read", 256);
  run<2, 4>("ds_read", 256);
  run<0, 0>("none", 512);
  run<0, 6>("v_add", 512);
  run<0, 12>("v_add", 512);
  run<1, 16>("s_add", 512);
  return 0;
}
