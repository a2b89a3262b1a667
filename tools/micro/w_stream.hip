// What limits the W fragment stream of the contraction loop (profiles/r05_mfma_ceiling.md: re-loading W from L2 costs a
// third of the MFMA rate even with power to spare)?  The f16x2 loop with W through the register ring (constant operands:
// no power limit), varying (a) how much W a workgroup walks (so how many L2 lines are hot), (b) whether all workgroups
// read the SAME lines or copies of their own (per XCD, per workgroup slot), (c) the ring depth.
// Build: hipcc --offload-arch=gfx950 -O3 w_stream.hip -o w_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int RING>
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ w, float* out, int steps, int walk, int wstride,
                                            int copy_mode, size_t copy_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // copy_mode 0: every workgroup the same W; 1: one copy per XCD (workgroup i runs on XCD i % 8); 2: 32 copies
  // (workgroup i -> copy i % 32: four per XCD); 3: one copy per workgroup (512 copies)
  const int copy = copy_mode == 0 ? 0 : copy_mode == 1 ? (blockIdx.x & 7) : copy_mode == 2 ? (blockIdx.x & 31) : blockIdx.x;
  const uint4* wl = w + (size_t)copy * copy_stride + (size_t)wave * wstride + lane;
  const uint4 one = make_uint4(0x2c002c00u, 0x2c002c00u, 0x2c002c00u, 0x2c002c00u);   // fp16 2^-4
  uint4 ring[RING][4];
  for (int d = 0; d < RING; ++d)
    for (int q = 0; q < 4; ++q) ring[d][q] = wl[((d % walk) * 4 + q) * 64];
  for (int s0 = 0; s0 < steps; s0 += RING) {
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      uint4 bf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf[q] = ring[d][q];
        ring[d][q] = wl[(((s0 + d + RING) % walk) * 4 + q) * 64];
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int pb = t == 0 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, one),
                                                               __builtin_bit_cast(f16x8, bf[2 * j + pb]), acc[i][j], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int RING>
static void run(const uint4* w, float* out, int blocks, int walk, int wstride, int copy_mode, size_t copy_stride) {
  const int steps = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<RING><<<blocks, 256>>>(w, out, 256, walk, wstride, copy_mode, copy_stride);
  hipDeviceSynchronize();
  const int n = 200;
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) k<RING><<<blocks, 256>>>(w, out, steps, walk, wstride, copy_mode, copy_stride);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)n * steps * 12.0 * blocks * 4 * 32768.0 / ms / 1e9;
  const char* cm[4] = {"all workgroups the same W", "one copy per XCD (8)", "32 copies (4 per XCD)", "one copy per workgroup"};
  // bytes a CU pulls per clock at this rate: 2 workgroups x 4 waves x 4 KB per step
  printf("| %d | %4d KB per wave, %5.2f MB per workgroup | %s | %.0f | %.3f |\n", RING, walk * 4, walk * 16.0 / 1024.0,
         cm[copy_mode], tf, tf / 2500.0);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 2;
  const int wstride = 256 * 4 * 64 + 64;                 // uint4 per wave at the longest walk
  const size_t copy_stride = (size_t)4 * wstride;       // uint4 per copy
  const size_t copies = 512;
  uint4* w;
  float* out;
  hipMalloc(&w, copies * copy_stride * 16);
  hipMemset(w, 0x2c, copies * copy_stride * 16);
  hipMalloc(&out, sizeof(float) * blocks * 256);
  printf("| ring depth | W walked | who reads what | TFLOP/s (constant operands: no power limit) | of 2.5 PF |\n|---:|---|---|---:|---:|\n");
  for (int walk : {256, 64, 16, 4}) {
    run<4>(w, out, blocks, walk, wstride, 0, copy_stride);
    run<4>(w, out, blocks, walk, wstride, 1, copy_stride);
    run<4>(w, out, blocks, walk, wstride, 2, copy_stride);
    if (walk <= 64) run<4>(w, out, blocks, walk, wstride, 3, copy_stride);
  }
  run<2>(w, out, blocks, 64, wstride, 0, copy_stride);
  run<8>(w, out, blocks, 64, wstride, 0, copy_stride);
  run<8>(w, out, blocks, 64, wstride, 2, copy_stride);
  return 0;
}
