// group_points / gather_points (+ group_points backward) for gfx950.
//
// Replaces GroupPointsForward (reference pointnet2_utils/csrc/
// grouping_kernel.cu:32-54: an ATen gather over an expanded view, which reads
// one 8-byte index per 4-byte element per channel), GroupPointsBackwardKernel
// (:57-96) and the torch.gather of functions.py:10-25.
//
// Forward: each lane owns FOUR consecutive (m,k) slots: one 32-byte index
// read, then for every channel of its channel group four L2-resident gathers
// and ONE 16-byte coalesced store.  The index is read once per channel group
// (all channels for C <= 8), not once per channel.
#include "s4g_common.h"

namespace s4g {

constexpr int GP_THREADS = 256;
constexpr int GP_CG = 8;  // channels per block row

// Vector path: MK % 4 == 0.
__global__ __launch_bounds__(GP_THREADS) void group_points_vec4_kernel(
    const float* __restrict__ in, const int64_t* __restrict__ idx, int C, int N,
    int64_t MK, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * GP_CG;
  const int64_t t4 = ((int64_t)blockIdx.x * GP_THREADS + threadIdx.x) * 4;
  if (t4 >= MK) return;
  const longlong2* ip =
      reinterpret_cast<const longlong2*>(idx + (size_t)b * MK + t4);
  const longlong2 a = ip[0], c = ip[1];
  const int j0 = (int)a.x, j1 = (int)a.y, j2 = (int)c.x, j3 = (int)c.y;
  const int cend = min(c0 + GP_CG, C);
  for (int ch = c0; ch < cend; ++ch) {
    const float* __restrict__ src = in + ((size_t)b * C + ch) * N;
    float4 v;
    v.x = src[j0];
    v.y = src[j1];
    v.z = src[j2];
    v.w = src[j3];
    *reinterpret_cast<float4*>(out + ((size_t)b * C + ch) * MK + t4) = v;
  }
}

__global__ __launch_bounds__(GP_THREADS) void group_points_scalar_kernel(
    const float* __restrict__ in, const int64_t* __restrict__ idx, int C, int N,
    int64_t MK, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * GP_CG;
  const int64_t t = (int64_t)blockIdx.x * GP_THREADS + threadIdx.x;
  if (t >= MK) return;
  const int j = (int)idx[(size_t)b * MK + t];
  const int cend = min(c0 + GP_CG, C);
  for (int ch = c0; ch < cend; ++ch)
    out[((size_t)b * C + ch) * MK + t] = in[((size_t)b * C + ch) * N + j];
}

__global__ __launch_bounds__(GP_THREADS) void group_points_backward_kernel(
    const float* __restrict__ gout, const int64_t* __restrict__ idx, int C,
    int N, int64_t MK, float* __restrict__ gin) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * GP_CG;
  const int64_t t = (int64_t)blockIdx.x * GP_THREADS + threadIdx.x;
  if (t >= MK) return;
  const int j = (int)idx[(size_t)b * MK + t];
  const int cend = min(c0 + GP_CG, C);
  for (int ch = c0; ch < cend; ++ch)
    atomicAdd(gin + ((size_t)b * C + ch) * N + j,
              gout[((size_t)b * C + ch) * MK + t]);
}

}  // namespace s4g

extern "C" int s4g_group_points_f32(const float* in_bcn,
                                    const int64_t* idx_bmk, int64_t B,
                                    int64_t C, int64_t N, int64_t M, int64_t K,
                                    float* out_bcmk, s4g_stream_t stream) {
  if (B < 0 || C < 0 || N <= 0 || M < 0 || K < 0 || B > 65535 ||
      N >= (1ll << 31))
    return S4G_EINVAL;
  const int64_t MK = M * K;
  if (B == 0 || C == 0 || MK == 0) return S4G_OK;
  if (!in_bcn || !idx_bmk || !out_bcmk) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned gy = (unsigned)((C + s4g::GP_CG - 1) / s4g::GP_CG);
  if (gy > 65535) return S4G_EINVAL;
  const bool aligned = ((uintptr_t)idx_bmk % 16 == 0) && ((uintptr_t)out_bcmk % 16 == 0);
  if (MK % 4 == 0 && aligned) {
    const dim3 grid((unsigned)((MK / 4 + s4g::GP_THREADS - 1) / s4g::GP_THREADS),
                    gy, (unsigned)B);
    hipLaunchKernelGGL(s4g::group_points_vec4_kernel, grid,
                       dim3(s4g::GP_THREADS), 0, st, in_bcn, idx_bmk, (int)C,
                       (int)N, MK, out_bcmk);
  } else {
    const dim3 grid((unsigned)((MK + s4g::GP_THREADS - 1) / s4g::GP_THREADS),
                    gy, (unsigned)B);
    hipLaunchKernelGGL(s4g::group_points_scalar_kernel, grid,
                       dim3(s4g::GP_THREADS), 0, st, in_bcn, idx_bmk, (int)C,
                       (int)N, MK, out_bcmk);
  }
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_group_points_backward_f32(const float* gout_bcmk,
                                             const int64_t* idx_bmk, int64_t B,
                                             int64_t C, int64_t N, int64_t M,
                                             int64_t K, float* gin_bcn,
                                             s4g_stream_t stream) {
  if (B < 0 || C < 0 || N <= 0 || M < 0 || K < 0 || B > 65535 ||
      N >= (1ll << 31))
    return S4G_EINVAL;
  if (B == 0 || C == 0) return S4G_OK;
  if (!gin_bcn) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(gin_bcn, 0, sizeof(float) * (size_t)(B * C * N), st);
  if (e != hipSuccess) return (int)e;
  const int64_t MK = M * K;
  if (MK == 0) return S4G_OK;
  if (!gout_bcmk || !idx_bmk) return S4G_EINVAL;
  const unsigned gy = (unsigned)((C + s4g::GP_CG - 1) / s4g::GP_CG);
  if (gy > 65535) return S4G_EINVAL;
  const dim3 grid((unsigned)((MK + s4g::GP_THREADS - 1) / s4g::GP_THREADS), gy,
                  (unsigned)B);
  hipLaunchKernelGGL(s4g::group_points_backward_kernel, grid,
                     dim3(s4g::GP_THREADS), 0, st, gout_bcmk, idx_bmk, (int)C,
                     (int)N, MK, gin_bcn);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_gather_points_f32(const float* in_bcn, const int64_t* idx_bm,
                                     int64_t B, int64_t C, int64_t N, int64_t M,
                                     float* out_bcm, s4g_stream_t stream) {
  // gather_points is group_points with K == 1.
  return s4g_group_points_f32(in_bcn, idx_bm, B, C, N, M, 1, out_bcm, stream);
}
