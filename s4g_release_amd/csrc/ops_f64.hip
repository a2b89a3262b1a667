// The five operators in DOUBLE precision (ABI 8).
//
// The reference dispatches every `pn2_ext` kernel over float AND double (`AT_DISPATCH_FLOATING_TYPES`:
// sampling_kernel.cu:148-167, ball_query_kernel.cu:116-128, grouping_kernel.cu:48-51 / 136-150,
// interpolate_kernel.cu:114-126 / 212-232 / 317-338).  S4G itself never leaves fp32 -- the float kernels
// of fps.hip, ball_query.hip, three_nn.hip, group.hip, interpolate.hip are the product path and the ones
// that are tuned -- but a drop-in for the operator API has to accept the other dtype too.  These are
// straightforward kernels with exactly the float kernels' semantics (tie rules, padding, strict '<'):
//   FPS            one workgroup per scene, running min-distances in the caller's workspace, the
//                  reference's tie rule through the same composite key
//   ball query     wave per centroid, index-order scan with ballot ranks, first-hit padding, early exit
//   3-NN           lane per query, strict '<' insertion in key order
//   group / gather / interpolate (+ the two backward scatters)   one output element per thread
// Distance contract as in include/s4g_ops.h, in double: strict (every operation rounded) or S4G_FLAG_FMAD.
#include "s4g_common.h"

namespace s4g {

template <bool FMAD>
__device__ __forceinline__ double dist2_f64(double x1, double y1, double z1, double x2, double y2, double z2) {
  const double dx = __dsub_rn(x2, x1), dy = __dsub_rn(y2, y1), dz = __dsub_rn(z2, z1);
  if constexpr (FMAD) {
    double t = __dmul_rn(dx, dx);
    t = __fma_rn(dy, dy, t);
    t = __fma_rn(dz, dz, t);
    return t;
  } else {
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
  }
}

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  const uint64_t u = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)u, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(u >> 32), m);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// ---------------------------------------------------------------- FPS (sampling_kernel.cu:49-119)
constexpr int F64_FPS_THREADS = 1024;

template <bool FMAD>
__global__ __launch_bounds__(F64_FPS_THREADS) void fps_f64_kernel(const double* __restrict__ xyz, int N, int M,
                                                                  int64_t* __restrict__ idx,
                                                                  double* __restrict__ temp, int lg_bs) {
  __shared__ double sd[F64_FPS_THREADS / 64];
  __shared__ uint32_t sk[F64_FPS_THREADS / 64];
  __shared__ int scur;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double* __restrict__ px = xyz + (size_t)b * 3 * N;
  const double* __restrict__ py = px + N;
  const double* __restrict__ pz = py + N;
  double* __restrict__ md = temp + (size_t)b * N;
  int64_t* __restrict__ out = idx + (size_t)b * M;
  const uint32_t bs_mask = (1u << lg_bs) - 1u;
  for (int j = t; j < N; j += F64_FPS_THREADS) md[j] = -1.0;   // :144, -1 = +infinity (:85-90)
  int cur = 0;
  if (t == 0) out[0] = 0;                                       // :62,67
  for (int i = 1; i < M; ++i) {
    const double cx = px[cur], cy = py[cur], cz = pz[cur];
    double best = 0.0;                  // (0, cur): :70-71 -- only a STRICTLY larger distance replaces it
    uint32_t bkey = 0xFFFFFFFFu;
    for (int j = t; j < N; j += F64_FPS_THREADS) {
      double d = dist2_f64<FMAD>(cx, cy, cz, px[j], py[j], pz[j]);
      const double o = md[j];
      if (o > d || o < 0.0) md[j] = d; else d = o;              // :85-90
      const uint32_t key = ((__brev((uint32_t)j & bs_mask) >> (32 - lg_bs)) << 23) | (uint32_t)j;
      if (d > best || (d == best && d > 0.0 && key < bkey)) {
        best = d;
        bkey = key;
      }
    }
    // maximise the distance, then minimise the key (the halving tree of :101-113 in closed form)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double od = shfl_xor_f64(best, m);
      const uint32_t ok = (uint32_t)__shfl_xor((int)bkey, m);
      if (od > best || (od == best && ok < bkey)) {
        best = od;
        bkey = ok;
      }
    }
    if (lane == 0) {
      sd[wave] = best;
      sk[wave] = bkey;
    }
    __syncthreads();
    if (t == 0) {
      double g = sd[0];
      uint32_t k = sk[0];
      for (int w = 1; w < F64_FPS_THREADS / 64; ++w)
        if (sd[w] > g || (sd[w] == g && sk[w] < k)) {
          g = sd[w];
          k = sk[w];
        }
      scur = g > 0.0 ? (int)(k & 0x7FFFFFu) : cur;              // nothing farther than 0: the index repeats
      out[i] = scur;
    }
    __syncthreads();
    cur = scur;
  }
}

// ---------------------------------------------------------------- ball query (ball_query_kernel.cu:33-76)
template <bool FMAD>
__global__ __launch_bounds__(256) void ball_query_f64_kernel(const double* __restrict__ xyz,
                                                             const double* __restrict__ ctr, int N, int M, double r2,
                                                             int K, int64_t* __restrict__ idx,
                                                             int64_t* __restrict__ cnt_out) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const double* __restrict__ px = xyz + (size_t)b * 3 * N;
  const double* __restrict__ c = ctr + (size_t)b * 3 * M;
  const double cx = c[m], cy = c[M + m], cz = c[2 * (size_t)M + m];
  int64_t* __restrict__ o = idx + ((size_t)b * M + m) * K;
  int cnt = 0, first = 0;
  for (int j0 = 0; j0 < N && cnt < K; j0 += 64) {                 // :57: stop at K
    const int j = j0 + lane;
    const bool hit = j < N && dist2_f64<FMAD>(cx, cy, cz, px[j], px[N + j], px[2 * (size_t)N + j]) < r2;   // strict, :63
    const uint64_t mask = __ballot(hit);
    if (mask) {
      if (cnt == 0) first = j0 + (__ffsll((unsigned long long)mask) - 1);
      const int slot = cnt + mask_rank(mask);
      if (hit && slot < K) o[slot] = j;                          // :69 (slot 0: the first hit itself)
      cnt += __popcll(mask);
    }
  }
  if (cnt > K) cnt = K;
  for (int s = cnt + lane; s < K; s += 64) o[s] = cnt ? first : 0;   // :64-67 first-hit padding; no hit: zeros (:109)
  if (lane == 0) cnt_out[(size_t)b * M + m] = cnt;                 // :74
}

// ---------------------------------------------------------------- 3-NN (interpolate_kernel.cu:32-81)
template <bool FMAD>
__global__ __launch_bounds__(256) void three_nn_f64_kernel(const double* __restrict__ q, const double* __restrict__ k,
                                                           int N1, int N2, int64_t* __restrict__ idx,
                                                           double* __restrict__ d2) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N1) return;
  const double* __restrict__ qx = q + (size_t)b * 3 * N1;
  const double* __restrict__ kx = k + (size_t)b * 3 * N2;
  const double x = qx[i], y = qx[N1 + i], z = qx[2 * (size_t)N1 + i];
  double b0 = 1e40, b1 = 0.0, b2 = 0.0;      // {1e40} / {-1}: :53-54, the other slots zero-initialised
  int i0 = -1, i1 = 0, i2 = 0;
  for (int j = 0; j < N2; ++j) {
    const double d = dist2_f64<FMAD>(kx[j], kx[N2 + j], kx[2 * (size_t)N2 + j], x, y, z);   // (x1 - x2): :60
    if (d < b0) {                            // strict: the earlier key stays in front (:64-71)
      b2 = b1; i2 = i1; b1 = b0; i1 = i0; b0 = d; i0 = j;
    } else if (d < b1) {
      b2 = b1; i2 = i1; b1 = d; i1 = j;
    } else if (d < b2) {
      b2 = d; i2 = j;
    }
  }
  const size_t o = ((size_t)b * N1 + i) * 3;
  // (an unfilled slot -- non-finite query -- as index 0, never the initialiser -1: see three_nn.hip nn_safe_index)
  idx[o] = i0 < 0 ? 0 : i0; idx[o + 1] = i1; idx[o + 2] = i2;
  d2[o] = b0; d2[o + 1] = b1; d2[o + 2] = b2;
}

// ---------------------------------------------------------------- group / interpolate (+ backward)
__global__ __launch_bounds__(256) void group_f64_kernel(const double* __restrict__ in, const int64_t* __restrict__ idx,
                                                        int C, int N, int64_t MK, double* __restrict__ out) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= MK) return;
  out[((size_t)b * C + ch) * MK + t] = in[((size_t)b * C + ch) * N + idx[(size_t)b * MK + t]];   // grouping_kernel.cu:48-51
}
__global__ __launch_bounds__(256) void group_bwd_f64_kernel(const double* __restrict__ g, const int64_t* __restrict__ idx,
                                                            int C, int N, int64_t MK, double* __restrict__ gin) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= MK) return;
  atomicAdd(gin + ((size_t)b * C + ch) * N + idx[(size_t)b * MK + t], g[((size_t)b * C + ch) * MK + t]);   // :72-96
}
template <bool FMAD>
__global__ __launch_bounds__(256) void interp_f64_kernel(const double* __restrict__ f, const int64_t* __restrict__ idx,
                                                         const double* __restrict__ w, int C, int N2, int N1,
                                                         double* __restrict__ out) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N1) return;
  const double* __restrict__ row = f + ((size_t)b * C + ch) * N2;
  const size_t o = ((size_t)b * N1 + i) * 3;
  double acc;                                                     // acc = 0; acc += f_k w_k (:160-174)
  if constexpr (FMAD) {
    acc = __fma_rn(row[idx[o + 2]], w[o + 2], __fma_rn(row[idx[o + 1]], w[o + 1], __dmul_rn(row[idx[o]], w[o])));
  } else {
    acc = __dadd_rn(__dadd_rn(__dmul_rn(row[idx[o]], w[o]), __dmul_rn(row[idx[o + 1]], w[o + 1])),
                    __dmul_rn(row[idx[o + 2]], w[o + 2]));
  }
  out[((size_t)b * C + ch) * N1 + i] = acc;
}
__global__ __launch_bounds__(256) void interp_bwd_f64_kernel(const double* __restrict__ g, const int64_t* __restrict__ idx,
                                                             const double* __restrict__ w, int C, int N2, int N1,
                                                             double* __restrict__ gin) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N1) return;
  const double gv = g[((size_t)b * C + ch) * N1 + i];
  double* __restrict__ row = gin + ((size_t)b * C + ch) * N2;
  const size_t o = ((size_t)b * N1 + i) * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) atomicAdd(row + idx[o + k], gv * w[o + k]);   // interpolate_kernel.cu:272-286
}

static int ref_block_lg_f64(int64_t n) {   // get_block() of sampling_kernel.cu:34-42 with the switch's 16-thread floor
  int cnt = 0;
  for (int64_t x = n - 1; x > 0; x >>= 1) ++cnt;
  return cnt > 9 ? 9 : (cnt < 4 ? 4 : cnt);
}

}  // namespace s4g

using namespace s4g;

extern "C" int s4g_fps_f64(const double* xyz_b3n, int64_t B, int64_t N, int64_t M, int64_t* idx_bm, void* ws,
                           size_t ws_bytes, int flags, s4g_stream_t stream) {
  if (B < 0 || N <= 0 || M <= 0 || M > N || N >= (1ll << 23) || B > 65535) return S4G_EINVAL;
  if (B == 0) return S4G_OK;
  if (!xyz_b3n || !idx_bm) return S4G_EINVAL;
  if (!ws || ws_bytes < sizeof(double) * (size_t)B * (size_t)N) return S4G_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int lg = ref_block_lg_f64(N);
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL(fps_f64_kernel<true>, dim3((unsigned)B), dim3(F64_FPS_THREADS), 0, st, xyz_b3n, (int)N, (int)M,
                       idx_bm, (double*)ws, lg);
  else
    hipLaunchKernelGGL(fps_f64_kernel<false>, dim3((unsigned)B), dim3(F64_FPS_THREADS), 0, st, xyz_b3n, (int)N, (int)M,
                       idx_bm, (double*)ws, lg);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_ball_query_f64(const double* xyz_b3n, const double* ctr_b3m, int64_t B, int64_t N, int64_t M,
                                  float radius, int64_t K, int64_t* idx_bmk, int64_t* cnt_bm, int flags,
                                  s4g_stream_t stream) {
  if (B < 0 || N <= 0 || M < 0 || K <= 0 || N >= (1ll << 31) || B > 65535) return S4G_EINVAL;
  if (B == 0 || M == 0) return S4G_OK;
  if (!xyz_b3n || !ctr_b3m || !idx_bmk || !cnt_bm) return S4G_EINVAL;
  const double r = (double)radius;      // the extension receives a C float and casts it: ball_query_kernel.cu:126
  const double r2 = r * r;              // :49
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((M + 3) / 4), (unsigned)B);
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL(ball_query_f64_kernel<true>, grid, dim3(256), 0, st, xyz_b3n, ctr_b3m, (int)N, (int)M, r2, (int)K,
                       idx_bmk, cnt_bm);
  else
    hipLaunchKernelGGL(ball_query_f64_kernel<false>, grid, dim3(256), 0, st, xyz_b3n, ctr_b3m, (int)N, (int)M, r2, (int)K,
                       idx_bmk, cnt_bm);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_three_nn_f64(const double* query_b3n1, const double* key_b3n2, int64_t B, int64_t N1, int64_t N2,
                                int64_t* idx_bn3, double* d2_bn3, int flags, s4g_stream_t stream) {
  if (B < 0 || N1 < 0 || N2 < 3 || N1 >= (1ll << 31) || N2 >= (1ll << 31) || B > 65535) return S4G_EINVAL;   // :106
  if (B == 0 || N1 == 0) return S4G_OK;
  if (!query_b3n1 || !key_b3n2 || !idx_bn3 || !d2_bn3) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((N1 + 255) / 256), (unsigned)B);
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL(three_nn_f64_kernel<true>, grid, dim3(256), 0, st, query_b3n1, key_b3n2, (int)N1, (int)N2, idx_bn3,
                       d2_bn3);
  else
    hipLaunchKernelGGL(three_nn_f64_kernel<false>, grid, dim3(256), 0, st, query_b3n1, key_b3n2, (int)N1, (int)N2, idx_bn3,
                       d2_bn3);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_group_points_f64(const double* in_bcn, const int64_t* idx_bmk, int64_t B, int64_t C, int64_t N,
                                    int64_t M, int64_t K, double* out_bcmk, s4g_stream_t stream) {
  if (B < 0 || C < 0 || N <= 0 || M < 0 || K < 0 || B > 65535 || C > 65535 || N >= (1ll << 31)) return S4G_EINVAL;
  const int64_t MK = M * K;
  if (B == 0 || C == 0 || MK == 0) return S4G_OK;
  if (!in_bcn || !idx_bmk || !out_bcmk) return S4G_EINVAL;
  hipLaunchKernelGGL(group_f64_kernel, dim3((unsigned)((MK + 255) / 256), (unsigned)C, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, in_bcn, idx_bmk, (int)C, (int)N, MK, out_bcmk);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_group_points_backward_f64(const double* gout_bcmk, const int64_t* idx_bmk, int64_t B, int64_t C,
                                             int64_t N, int64_t M, int64_t K, double* gin_bcn, s4g_stream_t stream) {
  if (B < 0 || C < 0 || N <= 0 || M < 0 || K < 0 || B > 65535 || C > 65535 || N >= (1ll << 31)) return S4G_EINVAL;
  if (B == 0 || C == 0) return S4G_OK;
  if (!gin_bcn) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(gin_bcn, 0, sizeof(double) * (size_t)(B * C * N), st);
  if (e != hipSuccess) return (int)e;
  const int64_t MK = M * K;
  if (MK == 0) return S4G_OK;
  if (!gout_bcmk || !idx_bmk) return S4G_EINVAL;
  hipLaunchKernelGGL(group_bwd_f64_kernel, dim3((unsigned)((MK + 255) / 256), (unsigned)C, (unsigned)B), dim3(256), 0, st,
                     gout_bcmk, idx_bmk, (int)C, (int)N, MK, gin_bcn);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_three_interpolate_f64(const double* feat_bcn2, const int64_t* idx_bn3, const double* w_bn3, int64_t B,
                                         int64_t C, int64_t N2, int64_t N1, double* out_bcn1, int flags,
                                         s4g_stream_t stream) {
  if (B < 0 || C < 0 || N2 <= 0 || N1 < 0 || B > 65535 || C > 65535 || N1 >= (1ll << 31)) return S4G_EINVAL;
  if (B == 0 || C == 0 || N1 == 0) return S4G_OK;
  if (!feat_bcn2 || !idx_bn3 || !w_bn3 || !out_bcn1) return S4G_EINVAL;
  const dim3 grid((unsigned)((N1 + 255) / 256), (unsigned)C, (unsigned)B);
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL(interp_f64_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, feat_bcn2, idx_bn3, w_bn3, (int)C,
                       (int)N2, (int)N1, out_bcn1);
  else
    hipLaunchKernelGGL(interp_f64_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, feat_bcn2, idx_bn3, w_bn3, (int)C,
                       (int)N2, (int)N1, out_bcn1);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_three_interpolate_backward_f64(const double* gout_bcn1, const int64_t* idx_bn3, const double* w_bn3,
                                                  int64_t B, int64_t C, int64_t N2, int64_t N1, double* gin_bcn2,
                                                  s4g_stream_t stream) {
  if (B < 0 || C < 0 || N2 <= 0 || N1 < 0 || B > 65535 || C > 65535 || N1 >= (1ll << 31)) return S4G_EINVAL;
  if (B == 0 || C == 0) return S4G_OK;
  if (!gin_bcn2) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(gin_bcn2, 0, sizeof(double) * (size_t)(B * C * N2), st);
  if (e != hipSuccess) return (int)e;
  if (N1 == 0) return S4G_OK;
  if (!gout_bcn1 || !idx_bn3 || !w_bn3) return S4G_EINVAL;
  hipLaunchKernelGGL(interp_bwd_f64_kernel, dim3((unsigned)((N1 + 255) / 256), (unsigned)C, (unsigned)B), dim3(256), 0, st,
                     gout_bcn1, idx_bn3, w_bn3, (int)C, (int)N2, (int)N1, gin_bcn2);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}
