// 3-nearest-neighbour search + inverse-distance weights for gfx950.
//
// Replaces PointSearchKernel (reference pointnet2_utils/csrc/
// interpolate_kernel.cu:32-81, host :92-132) and the three elementwise torch
// launches of FeatureInterpolator.forward (modules.py:118-120).
// Semantics: ascending triple by (d, key index), strict '<'
// insertion, SQUARED distances returned.  The reference's odd initialiser
// ({1e40} -> {+inf,0,0}, {-1} -> {-1,0,0}) is kept literally so even the
// out-of-contract cases agree with the oracle.
//
// Lanes are queries; the key index is wave-uniform, so key coordinates arrive
// through the scalar cache (s_load) and feed the VALU as SGPR operands -- no
// LDS staging and no per-lane key traffic at all.  The insertion network only
// runs when some lane of the wave needs it (wave-uniform branch).
#include "s4g_common.h"

namespace s4g {

constexpr int NN_THREADS = 256;

// WEIGHTS: write the inverse-distance weights of modules.py:118-120 instead of
// the squared distances (same arithmetic as interp_weights_kernel below).
template <bool FMAD, bool WEIGHTS, typename IdxT>
__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(
    const float* __restrict__ q, const float* __restrict__ key, int N1, int N2,
    float eps, IdxT* __restrict__ idx, float* __restrict__ d2out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * NN_THREADS + threadIdx.x;
  const float* __restrict__ qx = q + (size_t)b * 3 * N1;
  const float* __restrict__ kx = key + (size_t)b * 3 * N2;
  const float* __restrict__ ky = kx + N2;
  const float* __restrict__ kz = ky + N2;
  const int ii = i < N1 ? i : N1 - 1;
  const float x1 = qx[ii], y1 = qx[N1 + ii], z1 = qx[2 * N1 + ii];

  float b0 = __builtin_inff(), b1 = 0.0f, b2 = 0.0f;  // interpolate_kernel.cu:53
  int i0 = -1, i1 = 0, i2 = 0;                         // :54
  for (int j = 0; j < N2; ++j) {
    // (q - key) as in :60; the square makes the sign irrelevant bit-for-bit.
    const float d = dist2<FMAD>(kx[j], ky[j], kz[j], x1, y1, z1);
    const bool c0 = d < b0, c1 = d < b1, c2 = d < b2;
    if (__any(c0 | c1 | c2)) {
      // :63-73 -- first slot k with d < best[k], shift right, insert.
      const bool c01 = c0 | c1;
      b2 = c01 ? b1 : (c2 ? d : b2);
      i2 = c01 ? i1 : (c2 ? j : i2);
      b1 = c0 ? b0 : (c1 ? d : b1);
      i1 = c0 ? i0 : (c1 ? j : i1);
      b0 = c0 ? d : b0;
      i0 = c0 ? j : i0;
    }
  }
  if (i < N1) {
    const size_t o = ((size_t)b * N1 + i) * 3;
    idx[o + 0] = (IdxT)i0;
    idx[o + 1] = (IdxT)i1;
    idx[o + 2] = (IdxT)i2;
    if constexpr (WEIGHTS) {
      const float ia = __fdiv_rn(1.0f, b0 < eps ? eps : b0);
      const float ib = __fdiv_rn(1.0f, b1 < eps ? eps : b1);
      const float ic = __fdiv_rn(1.0f, b2 < eps ? eps : b2);
      const float s = __fadd_rn(__fadd_rn(ia, ib), ic);
      d2out[o + 0] = __fdiv_rn(ia, s);
      d2out[o + 1] = __fdiv_rn(ib, s);
      d2out[o + 2] = __fdiv_rn(ic, s);
    } else {
      d2out[o + 0] = b0;
      d2out[o + 1] = b1;
      d2out[o + 2] = b2;
    }
  }
}

// w = (1/max(d2,eps)) / sum_k(1/max(d2,eps)), summed left to right like
// torch.sum over a length-3 last dim (modules.py:118-120).
__global__ __launch_bounds__(NN_THREADS) void interp_weights_kernel(
    const float* __restrict__ d2, int64_t n, float eps, float* __restrict__ w) {
  const int64_t t = (int64_t)blockIdx.x * NN_THREADS + threadIdx.x;
  if (t >= n) return;
  const float a = d2[t * 3 + 0], b = d2[t * 3 + 1], c = d2[t * 3 + 2];
  const float ia = __fdiv_rn(1.0f, a < eps ? eps : a);
  const float ib = __fdiv_rn(1.0f, b < eps ? eps : b);
  const float ic = __fdiv_rn(1.0f, c < eps ? eps : c);
  const float s = __fadd_rn(__fadd_rn(ia, ib), ic);
  w[t * 3 + 0] = __fdiv_rn(ia, s);
  w[t * 3 + 1] = __fdiv_rn(ib, s);
  w[t * 3 + 2] = __fdiv_rn(ic, s);
}

}  // namespace s4g

extern "C" int s4g_three_nn_f32(const float* q_b3n1, const float* k_b3n2,
                                int64_t B, int64_t N1, int64_t N2,
                                int64_t* idx_bn3, float* d2_bn3, void* ws,
                                size_t ws_bytes, int flags,
                                s4g_stream_t stream) {
  (void)ws;
  (void)ws_bytes;
  if (B < 0 || N1 < 0 || N2 < 3 || B > 65535 || N2 >= (1ll << 31) ||
      N1 >= (1ll << 31))
    return S4G_EINVAL;  // N2 >= 3: interpolate_kernel.cu:106
  if (B == 0 || N1 == 0) return S4G_OK;
  if (!q_b3n1 || !k_b3n2 || !idx_bn3 || !d2_bn3) return S4G_EINVAL;
  const dim3 grid((unsigned)((N1 + s4g::NN_THREADS - 1) / s4g::NN_THREADS),
                  (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL((s4g::three_nn_kernel<true, false, int64_t>), grid,
                       dim3(s4g::NN_THREADS), 0, st, q_b3n1, k_b3n2, (int)N1,
                       (int)N2, 0.f, idx_bn3, d2_bn3);
  else
    hipLaunchKernelGGL((s4g::three_nn_kernel<false, false, int64_t>), grid,
                       dim3(s4g::NN_THREADS), 0, st, q_b3n1, k_b3n2, (int)N1,
                       (int)N2, 0.f, idx_bn3, d2_bn3);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_three_nn_weights_i32(const float* q_b3n1, const float* k_b3n2,
                                        int64_t B, int64_t N1, int64_t N2,
                                        float eps, int32_t* idx_bn3,
                                        float* w_bn3, void* ws, size_t ws_bytes,
                                        int flags, s4g_stream_t stream) {
  (void)ws;
  (void)ws_bytes;
  if (B < 0 || N1 < 0 || N2 < 3 || B > 65535 || N2 >= (1ll << 31) ||
      N1 >= (1ll << 31))
    return S4G_EINVAL;
  if (B == 0 || N1 == 0) return S4G_OK;
  if (!q_b3n1 || !k_b3n2 || !idx_bn3 || !w_bn3) return S4G_EINVAL;
  const dim3 grid((unsigned)((N1 + s4g::NN_THREADS - 1) / s4g::NN_THREADS),
                  (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL((s4g::three_nn_kernel<true, true, int32_t>), grid,
                       dim3(s4g::NN_THREADS), 0, st, q_b3n1, k_b3n2, (int)N1,
                       (int)N2, eps, idx_bn3, w_bn3);
  else
    hipLaunchKernelGGL((s4g::three_nn_kernel<false, true, int32_t>), grid,
                       dim3(s4g::NN_THREADS), 0, st, q_b3n1, k_b3n2, (int)N1,
                       (int)N2, eps, idx_bn3, w_bn3);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_interp_weights_f32(const float* d2_bn3, int64_t B,
                                      int64_t N1, float eps, float* w_bn3,
                                      s4g_stream_t stream) {
  if (B < 0 || N1 < 0) return S4G_EINVAL;
  const int64_t n = B * N1;
  if (n == 0) return S4G_OK;
  if (!d2_bn3 || !w_bn3) return S4G_EINVAL;
  const dim3 grid((unsigned)((n + s4g::NN_THREADS - 1) / s4g::NN_THREADS));
  hipLaunchKernelGGL(s4g::interp_weights_kernel, grid, dim3(s4g::NN_THREADS), 0,
                     (hipStream_t)stream, d2_bn3, n, eps, w_bn3);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}
