// Helpers shared by the MFMA contraction kernels (mlp_gemm.hip, mlp_heads.hip).
#pragma once
#include "s4g_common.h"

namespace s4g {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}


__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
  f16x2 v;
  v.x = (_Float16)a;
  v.y = (_Float16)b;
  return __builtin_bit_cast(uint32_t, v);
}

template <bool CLAMP>
__device__ __forceinline__ void split2_h(const float4 v, float s, uint2& h, uint2& l) {
  float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
  float r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if constexpr (CLAMP) x[e] = __builtin_amdgcn_fmed3f(x[e], -65504.f, 65504.f);
    r[e] = x[e] - (float)(_Float16)x[e];
  }
  h.x = pack_h2(x[0], x[1]);
  h.y = pack_h2(x[2], x[3]);
  l.x = pack_h2(r[0], r[1]);
  l.y = pack_h2(r[2], r[3]);
}

__device__ __forceinline__ float amax_slots(const float* __restrict__ slots, int lane) {
  return __uint_as_float(wave_max_u32(__float_as_uint(slots[lane])));
}

// Activation maxima are kept PER SCENE, so that a scene's scales -- and therefore its results --
// do not depend on which other scenes share the batch: slot row s (64 words) belongs to scene
// s = row / rps.  A tile takes the maximum over the scenes its rows [p_lo, p_hi] touch (one scene
// whenever rps is a multiple of the tile height, which holds at every level of the shipped
// configuration) and publishes its own maximum to each of them.
__device__ __forceinline__ float amax_rows(const float* __restrict__ slots, int lane, int p_lo, int p_hi,
                                           int rps) {
  const int s0 = rps > 0 ? p_lo / rps : 0, s1 = rps > 0 ? p_hi / rps : 0;
  float m = amax_slots(slots + (size_t)s0 * 64, lane);
  for (int sc = s0 + 1; sc <= s1; ++sc) m = fmaxf(m, amax_slots(slots + (size_t)sc * 64, lane));
  return m;
}
__device__ __forceinline__ void amax_publish(uint32_t* __restrict__ slots, uint32_t wm, int slot, int p_lo,
                                             int p_hi, int rps) {
  const int s0 = rps > 0 ? p_lo / rps : 0, s1 = rps > 0 ? p_hi / rps : 0;
  for (int sc = s0; sc <= s1; ++sc) atomicMax(slots + (size_t)sc * 64 + (slot & 63), wm);
}


// One 32x32x16 matrix-core step on 16-byte operand registers: PL == 2 fp16 planes, PL == 1 bf16.
template <int PL>
__device__ __forceinline__ f32x16 chain_mfma(const uint4 a, const uint4 b, const f32x16 c) {
  if constexpr (PL == 2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace s4g
