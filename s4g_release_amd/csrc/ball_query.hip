// Ball query for gfx950.
//
// Replaces BallQueryKernel (reference pointnet2_utils/csrc/
// ball_query_kernel.cu:33-76, host :89-133).  Semantics:
// r2 = radius*radius in fp32; per centroid scan points in INDEX ORDER, keep the
// first K with d < r2 (strict); slots cnt..K-1 hold the first hit; rows
// without a hit stay zero; count = min(hits, K).
//
// Kernel in this file: index-order scan, one WAVE per centroid with the 64
// lanes on 64 consecutive points.  A ballot gives the hit mask, mbcnt the rank
// of each hit inside the mask, so hits land in their output slot already in
// index order and the wave leaves the scan as soon as K are found (the
// reference's thread-per-centroid loop cannot exit a wave early and reads xyz
// as AoS).  xyz is consumed as the (B,3,N) SoA planes it arrives in: each
// plane load is one fully coalesced 256-byte request per wave.
#include "s4g_common.h"

namespace s4g {

constexpr int BQ_WAVES_PER_BLOCK = 4;
constexpr int BQ_UNROLL = 4;

template <bool FMAD, typename IdxT>
__global__ __launch_bounds__(64 * BQ_WAVES_PER_BLOCK) void ball_query_scan_kernel(
    const float* __restrict__ xyz, const float* __restrict__ ctr, int N, int M,
    float r2, int K, IdxT* __restrict__ idx, IdxT* __restrict__ cnt_out) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * BQ_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* __restrict__ px = xyz + (size_t)b * 3 * N;
  const float* __restrict__ py = px + N;
  const float* __restrict__ pz = py + N;
  const float* __restrict__ c = ctr + (size_t)b * 3 * M;
  const float cx = c[m], cy = c[M + m], cz = c[2 * M + m];
  IdxT* __restrict__ row = idx + ((size_t)b * M + m) * K;

  int cnt = 0;    // wave-uniform
  int first = 0;  // index of the first hit
  for (int j0 = 0; j0 < N && cnt < K; j0 += 64 * BQ_UNROLL) {
    float x[BQ_UNROLL], y[BQ_UNROLL], z[BQ_UNROLL];
#pragma unroll
    for (int u = 0; u < BQ_UNROLL; ++u) {
      const int j = j0 + u * 64 + lane;
      const int jj = j < N ? j : N - 1;
      x[u] = px[jj];
      y[u] = py[jj];
      z[u] = pz[jj];
    }
#pragma unroll
    for (int u = 0; u < BQ_UNROLL; ++u) {
      const int j = j0 + u * 64 + lane;
      const float d = dist2<FMAD>(cx, cy, cz, x[u], y[u], z[u]);
      const bool hit = (j < N) && (d < r2);
      const uint64_t mask = __ballot(hit);
      if (mask != 0) {
        const int pos = cnt + mask_rank(mask);
        if (hit && pos < K) row[pos] = (IdxT)j;
        if (cnt == 0) first = j0 + u * 64 + (__ffsll((unsigned long long)mask) - 1);
        cnt += __popcll(mask);
      }
    }
  }
  if (cnt > K) cnt = K;
  const IdxT fill = (IdxT)first;  // 0 when there was no hit
  for (int k = cnt + lane; k < K; k += 64) row[k] = fill;
  if (lane == 0) cnt_out[(size_t)b * M + m] = (IdxT)cnt;
}

}  // namespace s4g

namespace s4g {
template <typename IdxT>
static int ball_query_dispatch(const float* xyz, const float* ctr, int64_t B,
                               int64_t N, int64_t M, float radius, int64_t K,
                               IdxT* idx, IdxT* cnt, int flags,
                               hipStream_t st) {
  if (B < 0 || N <= 0 || M < 0 || K <= 0 || N >= (1ll << 31) || B > 65535)
    return S4G_EINVAL;
  if (B == 0 || M == 0) return S4G_OK;
  if (!xyz || !ctr || !idx || !cnt) return S4G_EINVAL;
  // r2 exactly as ball_query_kernel.cu:49 computes it: fp32 product.
  const float r2 = radius * radius;
  const dim3 block(64 * BQ_WAVES_PER_BLOCK);
  const dim3 grid((unsigned)((M + BQ_WAVES_PER_BLOCK - 1) / BQ_WAVES_PER_BLOCK),
                  (unsigned)B);
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL((ball_query_scan_kernel<true, IdxT>), grid, block, 0, st,
                       xyz, ctr, (int)N, (int)M, r2, (int)K, idx, cnt);
  else
    hipLaunchKernelGGL((ball_query_scan_kernel<false, IdxT>), grid, block, 0, st,
                       xyz, ctr, (int)N, (int)M, r2, (int)K, idx, cnt);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}
}  // namespace s4g

extern "C" int s4g_ball_query_f32(const float* xyz_b3n, const float* ctr_b3m,
                                  int64_t B, int64_t N, int64_t M, float radius,
                                  int64_t K, int64_t* idx_bmk, int64_t* cnt_bm,
                                  void* ws, size_t ws_bytes, int flags,
                                  s4g_stream_t stream) {
  (void)ws;
  (void)ws_bytes;
  return s4g::ball_query_dispatch<int64_t>(xyz_b3n, ctr_b3m, B, N, M, radius, K,
                                           idx_bmk, cnt_bm, flags,
                                           (hipStream_t)stream);
}

extern "C" int s4g_ball_query_i32(const float* xyz_b3n, const float* ctr_b3m,
                                  int64_t B, int64_t N, int64_t M, float radius,
                                  int64_t K, int32_t* idx_bmk, int32_t* cnt_bm,
                                  void* ws, size_t ws_bytes, int flags,
                                  s4g_stream_t stream) {
  (void)ws;
  (void)ws_bytes;
  return s4g::ball_query_dispatch<int32_t>(xyz_b3n, ctr_b3m, B, N, M, radius, K,
                                           idx_bmk, cnt_bm, flags,
                                           (hipStream_t)stream);
}

namespace s4g {
size_t ball_query_workspace_bytes(int64_t, int64_t, int64_t, int64_t) { return 0; }
}  // namespace s4g
