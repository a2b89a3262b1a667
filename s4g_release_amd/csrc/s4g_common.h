// Device-side helpers shared by the gfx950 kernels of libs4g_hip.so.
// Wave = 64 lanes everywhere; this code targets CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s4g_ops.h"

#define S4G_WAVE 64

#define S4G_LAUNCH_CHECK()                        \
  do {                                            \
    hipError_t e_ = hipGetLastError();            \
    if (e_ != hipSuccess) return (int)e_;         \
  } while (0)

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace s4g {
// A/B and test knobs (include/s4g_ops.h lists them): read from the environment ONLY when the process also sets
// S4G_TEST_KNOBS=1 (or the library is a -DS4G_VARIANTS measurement build).  A production process that does not set the
// master switch cannot have its kernel selection changed by a stray variable in some launcher's environment; every knob
// selects an exact path, so the switch changes speed, never results.  Not cached: tests flip it within one process.
inline bool test_knobs_enabled() {
#ifdef S4G_VARIANTS
  return true;
#else
  const char* m = getenv("S4G_TEST_KNOBS");
  return m && m[0] == '1' && m[1] == 0;
#endif
}
inline const char* knob(const char* name) { return test_knobs_enabled() ? getenv(name) : nullptr; }
}  // namespace s4g

namespace s4g {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE setting: one slot per (call site,
// device), set on the first launch there and again whenever a launch needs more than the slot
// holds; a failure is returned to the caller and never memoised.
struct LdsAttrCache {
  std::atomic<int> bytes[64];
};
inline int allow_dynamic_lds(const void* func, size_t lds_bytes, LdsAttrCache& cache) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const bool slot = dev >= 0 && dev < 64;
  if (slot && cache.bytes[dev].load(std::memory_order_relaxed) >= (int)lds_bytes) return 0;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  if (slot) cache.bytes[dev].store((int)lds_bytes, std::memory_order_relaxed);
  return 0;
}

// Squared distance with the arithmetic contract of include/s4g_ops.h.
// The translation units are compiled with -ffp-contract=off, and the
// __f*_rn intrinsics additionally pin each rounding.
template <bool FMAD>
__device__ __forceinline__ float dist2(float x1, float y1, float z1, float x2,
                                       float y2, float z2) {
  const float dx = __fsub_rn(x2, x1);
  const float dy = __fsub_rn(y2, y1);
  const float dz = __fsub_rn(z2, z1);
  if constexpr (FMAD) {
    float t = __fmul_rn(dx, dx);
    t = __fmaf_rn(dy, dy, t);
    t = __fmaf_rn(dz, dz, t);
    return t;
  } else {
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)),
                     __fmul_rn(dz, dz));
  }
}

// ---- DPP helpers (gfx9 encodings) ------------------------------------------
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E,
// row_half_mirror = 0x141, row_mirror = 0x140.
// (old = 0 with bound_ctrl: every lane of these permutations has a valid source, so the value is
// the same as with old = v -- but in this form the compiler folds the move into its consumer,
// `v_max_u32_dpp v, v, v quad_perm:...` = ONE instruction per reduction step instead of
// mov / nop / mov_dpp / max: half the instructions and half the dependent chain of every
// row16_* / wave_* reduction below)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// After these four steps every lane holds the max of its row of 16 lanes.
__device__ __forceinline__ uint32_t row16_max_u32(uint32_t v) {
  v = max(v, dpp_u32<0xB1>(v));
  v = max(v, dpp_u32<0x4E>(v));
  v = max(v, dpp_u32<0x141>(v));
  v = max(v, dpp_u32<0x140>(v));
  return v;
}
__device__ __forceinline__ uint32_t row16_min_u32(uint32_t v) {
  v = min(v, dpp_u32<0xB1>(v));
  v = min(v, dpp_u32<0x4E>(v));
  v = min(v, dpp_u32<0x141>(v));
  v = min(v, dpp_u32<0x140>(v));
  return v;
}

// Wave-wide (64 lanes) reductions; the result is wave-uniform (SGPR).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = row16_max_u32(v);
  const uint32_t a = __builtin_amdgcn_readlane(v, 0);
  const uint32_t b = __builtin_amdgcn_readlane(v, 16);
  const uint32_t c = __builtin_amdgcn_readlane(v, 32);
  const uint32_t d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = row16_min_u32(v);
  const uint32_t a = __builtin_amdgcn_readlane(v, 0);
  const uint32_t b = __builtin_amdgcn_readlane(v, 16);
  const uint32_t c = __builtin_amdgcn_readlane(v, 32);
  const uint32_t d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

// Inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16
// (row_shr 1,2,4,8 with zero fill), then row_bcast15 / row_bcast31 carry the
// row totals across (6 DPP adds, no LDS crossbar traffic).
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
#define S4G_DPP0(CTRL, RM) \
  (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, 0xF, true)
  v += S4G_DPP0(0x111, 0xF);
  v += S4G_DPP0(0x112, 0xF);
  v += S4G_DPP0(0x114, 0xF);
  v += S4G_DPP0(0x118, 0xF);
  v += S4G_DPP0(0x142, 0xA);
  v += S4G_DPP0(0x143, 0xC);
#undef S4G_DPP0
  return v;
}

__device__ __forceinline__ int lane_id() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
}

// Number of set bits of `mask` strictly below this lane.
__device__ __forceinline__ int mask_rank(uint64_t mask) {
  return (int)__builtin_amdgcn_mbcnt_hi(
      (uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

}  // namespace s4g
