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

// (x0 s, x1 s) as two packed fp16 pairs: h = RN16(x s), l = RN16(x s - h).  s is a power of two, so
// x s and the residual are exact in fp32 and every half is rounded once; the mixed-precision FMA
// forms convert on the way out and read h's halves directly (4 instructions per pair instead of
// the 10 of multiply / convert / convert back / subtract / convert / pack).
//
// ONE asm block per two pairs, interleaved: v_fma_mixhi writes a register HALF (op_sel destination), and gfx940-class
// hardware wants a wait state between such a write and a VALU read of that register (LLVM's dst-sel forwarding hazard).
// The hazard recognizer does not look inside inline asm and is free to place separate asm statements back to back, so the
// distance is built into the block: every read of h01 / h23 sits at least two instructions behind the mixhi that
// completed it, whatever the compiler schedules around the block.
__device__ __forceinline__ void split_quad_h2(float x0, float x1, float x2, float x3, float s0, float s1, float s2, float s3,
                                              uint32_t& h01, uint32_t& l01, uint32_t& h23, uint32_t& l23) {
  uint32_t ha, hb, la, lb;
  asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
      "v_fma_mixhi_f16 %0, %5, %9, 0\n\t"
      "v_fma_mixlo_f16 %1, %6, %10, 0\n\t"
      "v_fma_mixhi_f16 %1, %7, %11, 0\n\t"
      "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %5, %9, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %6, %10, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %7, %11, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(ha), "=&v"(hb), "=&v"(la), "=&v"(lb)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(s0), "v"(s1), "v"(s2), "v"(s3));
  h01 = ha;
  h23 = hb;
  l01 = la;
  l23 = lb;
}

__device__ __forceinline__ void split_quad_h(float x0, float x1, float x2, float x3, float s,
                                             uint32_t& h01, uint32_t& l01, uint32_t& h23, uint32_t& l23) {
  uint32_t ha, hb, la, lb;
  asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
      "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
      "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
      "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
      "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(ha), "=&v"(hb), "=&v"(la), "=&v"(lb)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(s));
  h01 = ha;
  h23 = hb;
  l01 = la;
  l23 = lb;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// (a0 b0 + c0, a1 b1 + c1), each rounded once: one v_pk_fma_f32 (two fp32 FMAs per lane and issue slot)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <bool CLAMP>
__device__ __forceinline__ void split2_h(const float4 v, float s, uint2& h, uint2& l) {
  if constexpr (CLAMP) {
    float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_fmed3f(x[e], -65504.f, 65504.f);
    split_quad_h(x[0], x[1], x[2], x[3], 1.0f, h.x, l.x, h.y, l.y);
  } else {
    split_quad_h(v.x, v.y, v.z, v.w, s, h.x, l.x, h.y, l.y);
  }
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


// 16-byte load with the streaming hint (data read once: do not keep it in L2)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// A position in a fragment-ordered W stream: buffer resource of the tensor + scalar byte offset.
struct WRef {
  __amdgpu_buffer_rsrc_t r;
  uint32_t off;
};
__device__ __forceinline__ WRef wref(const void* base, size_t off) {
  return WRef{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000), (uint32_t)off};
}
__device__ __forceinline__ WRef operator+(const WRef& w, size_t bytes) { return WRef{w.r, w.off + (uint32_t)bytes}; }
// 16 bytes per lane at w + byte_off (scalar) + lane_off (vector; a constant part becomes the
// instruction's immediate)
__device__ __forceinline__ uint4 wref_load(const WRef& w, uint32_t lane_off, uint32_t byte_off) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w.r, (int)lane_off, (int)(w.off + byte_off), 0));
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
