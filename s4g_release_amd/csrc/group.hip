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

// xyz grouping (C == 3, the QueryGrouper call of modules.py:42): the gather rate of
// the texture addresser, not HBM, bounds the generic kernel (12 four-byte gathers per
// lane).  With an index-ordered (x, y, z, 0) copy of the cloud a neighbour is ONE
// 16-byte gather: 4 instead of 12 gather instructions per lane.
__global__ __launch_bounds__(GP_THREADS) void xyz_to_aos_kernel(const float* __restrict__ in, int N,
                                                                float4* __restrict__ aos) {
  const int j = blockIdx.x * GP_THREADS + threadIdx.x;
  const int b = blockIdx.y;
  if (j >= N) return;
  const float* p = in + (size_t)b * 3 * N;
  aos[(size_t)b * N + j] = make_float4(p[j], p[N + j], p[2 * (size_t)N + j], 0.f);
}

__global__ __launch_bounds__(GP_THREADS) void group_xyz_aos_kernel(const float4* __restrict__ aos,
                                                                   const int64_t* __restrict__ idx,
                                                                   int N, int64_t MK,
                                                                   float* __restrict__ out) {
  const int b = blockIdx.y;
  const int64_t t4 = ((int64_t)blockIdx.x * GP_THREADS + threadIdx.x) * 4;
  if (t4 >= MK) return;
  const longlong2* ip = reinterpret_cast<const longlong2*>(idx + (size_t)b * MK + t4);
  const longlong2 a = ip[0], c = ip[1];
  const float4* __restrict__ src = aos + (size_t)b * N;
  const float4 p0 = src[(int)a.x], p1 = src[(int)a.y], p2 = src[(int)c.x], p3 = src[(int)c.y];
  float* o = out + (size_t)b * 3 * MK + t4;
  *reinterpret_cast<float4*>(o) = make_float4(p0.x, p1.x, p2.x, p3.x);
  *reinterpret_cast<float4*>(o + MK) = make_float4(p0.y, p1.y, p2.y, p3.y);
  *reinterpret_cast<float4*>(o + 2 * MK) = make_float4(p0.z, p1.z, p2.z, p3.z);
}

// (x_j - c_m, 0) per grouped row: the first SA layer's input as one 16-byte record per row
__global__ __launch_bounds__(GP_THREADS) void group_rel_xyz_kernel(const float* __restrict__ xyz,
                                                                   const float* __restrict__ ctr,
                                                                   const int* __restrict__ idx, int N,
                                                                   int M, int K,
                                                                   float4* __restrict__ out) {
  const int b = blockIdx.y;
  const int64_t MK = (int64_t)M * K;
  const int64_t t = (int64_t)blockIdx.x * GP_THREADS + threadIdx.x;
  if (t >= MK) return;
  const int m = (int)(t / K);
  const int j = idx[(size_t)b * MK + t];
  const float* __restrict__ x = xyz + (size_t)b * 3 * N;
  const float* __restrict__ c = ctr + (size_t)b * 3 * M;
  out[(size_t)b * MK + t] = make_float4(__fsub_rn(x[j], c[m]), __fsub_rn(x[N + j], c[M + m]),
                                        __fsub_rn(x[2 * (size_t)N + j], c[2 * (size_t)M + m]), 0.f);
}

// ---- the same records WITHOUT the duplicates (ABI 8).  ball_query pads a ball with fewer than K hits by
// repeating its first hit (ball_query_kernel.cu:64-67), so slots count .. K-1 of a centroid are copies of
// slot 0 and the max over the K neighbours (modules.py:243) does not see them: the first SA level only
// has to push a centroid's DISTINCT rows through its shared MLP.  Scene b's rows start at b M K as before;
// centroid m contributes c4(m) = its count (at least 1: an empty ball is K copies of point 0) rounded up to
// a multiple of 4 (slots past the count are copies, so the padding rows are too), centroids back to back.
//   row_start[b M + m]   first row of centroid m, relative to the scene's base (exclusive scan of c4)
//   seg4[row / 4]        b M + m for every group of 4 rows; -1 behind the scene's last centroid
//   rows[b]              the scene's rows rounded up to GU_TILE (the contraction's tile height); the rows
//                        between the last centroid and that edge are zero records of segment -1.
//                        rows[b] == M K: the scene keeps the PLAIN layout (see the scan kernel)
constexpr int GU_THREADS = 1024, GU_TILE = 256;   // (the tallest contraction tile: 256 rows in the single-plane form)
__global__ __launch_bounds__(GU_THREADS) void group_unique_scan_kernel(const int* __restrict__ cnt, int M, int K,
                                                                       int* __restrict__ row_start,
                                                                       int* __restrict__ rows,
                                                                       float4* __restrict__ out,
                                                                       int* __restrict__ seg4) {
  __shared__ int wsum[GU_THREADS / 64];
  __shared__ int total_s;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (M + GU_THREADS - 1) / GU_THREADS;
  const int m0 = t * per;
  int local = 0;
  for (int i = 0; i < per; ++i) {
    const int m = m0 + i;
    if (m < M) local += (max(cnt[(size_t)b * M + m], 1) + 3) & ~3;
  }
  int incl = local;   // inclusive scan over the wave, then over the waves
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int w = 0; w < GU_THREADS / 64; ++w) {
      const int v = wsum[w];
      wsum[w] = run;
      run += v;
    }
    total_s = run;
  }
  __syncthreads();
  int start = wsum[wave] + incl - local;
  for (int i = 0; i < per; ++i) {
    const int m = m0 + i;
    if (m < M) {
      row_start[(size_t)b * M + m] = start;
      start += (max(cnt[(size_t)b * M + m], 1) + 3) & ~3;
    }
  }
  // A scene whose balls are mostly full gains nothing (the segmented epilogue costs ~9 % of the launch): from
  // 7/8 of the capacity on it keeps the plain layout -- centroid m at row m K, all K slots -- and says so with
  // rows[b] == M K; the contraction then takes its 64-row epilogue for that scene.  Decided per scene, from
  // the scene alone: results never depend on the rest of the batch.
  // ... and a compact layout whose tile-padded row count reaches M K would carry the plain layout's MARKER
  // with compact row starts (possible below 2 048 rows, where 7/8 M K rounded up to a tile can be M K): plain too.
  if (total_s > (int)((int64_t)M * K / 8 * 7) || (total_s + GU_TILE - 1) / GU_TILE * GU_TILE >= M * K) {
    for (int i = 0; i < per; ++i) {
      const int m = m0 + i;
      if (m < M) row_start[(size_t)b * M + m] = m * K;
    }
    if (t == 0) rows[b] = M * K;
    return;
  }
  const int total = total_s;                                   // a multiple of 4
  const int padded = (total + GU_TILE - 1) / GU_TILE * GU_TILE;   // < M K (checked above)
  if (t == 0) rows[b] = padded;
  const size_t base = (size_t)b * M * K;
  for (int r = total + t; r < padded; r += GU_THREADS) {
    out[base + r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((r & 3) == 0) seg4[(base + r) >> 2] = -1;
  }
}

__global__ __launch_bounds__(GP_THREADS) void group_rel_unique_kernel(const float* __restrict__ xyz,
                                                                      const float* __restrict__ ctr,
                                                                      const int* __restrict__ idx,
                                                                      const int* __restrict__ cnt,
                                                                      const int* __restrict__ row_start,
                                                                      const int* __restrict__ rows, int N,
                                                                      int M, int K, float4* __restrict__ out,
                                                                      int* __restrict__ seg4) {
  const int b = blockIdx.y;
  const int64_t MK = (int64_t)M * K;
  const int64_t t = (int64_t)blockIdx.x * GP_THREADS + threadIdx.x;
  if (t >= MK) return;
  const int m = (int)(t / K), k = (int)(t - (int64_t)m * K);
  const int c4 = rows[b] == M * K ? K : (max(cnt[(size_t)b * M + m], 1) + 3) & ~3;   // (plain layout: every slot)
  if (k >= c4) return;
  const int j = idx[(size_t)b * MK + t];          // slots past the count repeat slot 0 (or hold 0: empty ball)
  const float* __restrict__ x = xyz + (size_t)b * 3 * N;
  const float* __restrict__ c = ctr + (size_t)b * 3 * M;
  const size_t row = (size_t)b * MK + row_start[(size_t)b * M + m] + k;
  out[row] = make_float4(__fsub_rn(x[j], c[m]), __fsub_rn(x[N + j], c[M + m]),
                         __fsub_rn(x[2 * (size_t)N + j], c[2 * (size_t)M + m]), 0.f);
  if ((k & 3) == 0) seg4[row >> 2] = b * M + m;
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

extern "C" int s4g_group_points_xyz_f32(const float* xyz_b3n, const int64_t* idx_bmk, int64_t B,
                                        int64_t N, int64_t M, int64_t K, float* out_b3mk,
                                        void* ws, size_t ws_bytes, s4g_stream_t stream) {
  const int64_t MK = M * K;
  const bool aligned = ((uintptr_t)idx_bmk % 16 == 0) && ((uintptr_t)out_b3mk % 16 == 0) &&
                       ((uintptr_t)ws % 16 == 0);
  // anything the fast form does not cover goes through the generic kernel: same results
  if (!ws || ws_bytes < (size_t)B * (size_t)N * sizeof(float4) || MK % 4 != 0 || !aligned ||
      B <= 0 || B > 65535 || N <= 0 || N >= (1ll << 31) || MK == 0)
    return s4g_group_points_f32(xyz_b3n, idx_bmk, B, 3, N, M, K, out_b3mk, stream);
  if (!xyz_b3n || !idx_bmk || !out_b3mk) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float4* aos = (float4*)ws;
  hipLaunchKernelGGL(s4g::xyz_to_aos_kernel, dim3((unsigned)((N + s4g::GP_THREADS - 1) / s4g::GP_THREADS), (unsigned)B),
                     dim3(s4g::GP_THREADS), 0, st, xyz_b3n, (int)N, aos);
  S4G_LAUNCH_CHECK();
  hipLaunchKernelGGL(s4g::group_xyz_aos_kernel,
                     dim3((unsigned)((MK / 4 + s4g::GP_THREADS - 1) / s4g::GP_THREADS), (unsigned)B),
                     dim3(s4g::GP_THREADS), 0, st, aos, idx_bmk, (int)N, MK, out_b3mk);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_group_rel_xyz_i32(const float* xyz_b3n, const float* ctr_b3m, const int32_t* idx_bmk,
                                     int64_t B, int64_t N, int64_t M, int64_t K, float* rel_pk4,
                                     s4g_stream_t stream) {
  if (B < 0 || N <= 0 || M < 0 || K < 0 || B > 65535 || N >= (1ll << 31) || M * K >= (1ll << 31))
    return S4G_EINVAL;
  const int64_t MK = M * K;
  if (B == 0 || MK == 0) return S4G_OK;
  if (!xyz_b3n || !ctr_b3m || !idx_bmk || !rel_pk4 || ((uintptr_t)rel_pk4 & 15)) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(s4g::group_rel_xyz_kernel,
                     dim3((unsigned)((MK + s4g::GP_THREADS - 1) / s4g::GP_THREADS), (unsigned)B),
                     dim3(s4g::GP_THREADS), 0, st, xyz_b3n, ctr_b3m, idx_bmk, (int)N, (int)M, (int)K,
                     reinterpret_cast<float4*>(rel_pk4));
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_group_rel_xyz_unique_i32(const float* xyz_b3n, const float* ctr_b3m, const int32_t* idx_bmk,
                                            const int32_t* cnt_bm, int64_t B, int64_t N, int64_t M, int64_t K,
                                            float* rel_pk4, int32_t* seg4, int32_t* row_start_bm,
                                            int32_t* rows_b, s4g_stream_t stream) {
  if (B < 0 || N <= 0 || M < 0 || K < 0 || B > 65535 || N >= (1ll << 31) || B * M * K >= (1ll << 31))
    return S4G_EINVAL;
  const int64_t MK = M * K;
  if (B == 0 || MK == 0) return S4G_OK;
  if ((K & 3) || (MK % s4g::GU_TILE) || M > s4g::GU_THREADS * 64) return S4G_EUNSUPPORTED;
  if (!xyz_b3n || !ctr_b3m || !idx_bmk || !cnt_bm || !rel_pk4 || !seg4 || !row_start_bm || !rows_b ||
      ((uintptr_t)rel_pk4 & 15))
    return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(s4g::group_unique_scan_kernel, dim3((unsigned)B), dim3(s4g::GU_THREADS), 0, st, cnt_bm, (int)M,
                     (int)K, row_start_bm, rows_b, reinterpret_cast<float4*>(rel_pk4), seg4);
  S4G_LAUNCH_CHECK();
  hipLaunchKernelGGL(s4g::group_rel_unique_kernel,
                     dim3((unsigned)((MK + s4g::GP_THREADS - 1) / s4g::GP_THREADS), (unsigned)B),
                     dim3(s4g::GP_THREADS), 0, st, xyz_b3n, ctr_b3m, idx_bmk, cnt_bm, row_start_bm, rows_b, (int)N, (int)M,
                     (int)K, reinterpret_cast<float4*>(rel_pk4), seg4);
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
