// three_interpolate forward / backward for gfx950.
//
// Replaces InterpolateForwardKernel / InterpolateBackwardKernel (reference
// pointnet2_utils/csrc/interpolate_kernel.cu:138-181, :243-286; hosts
// :191-236, :296-341).  Forward semantics:
//   acc = 0; for k in 0,1,2: acc += feat[b,c,idx[b,n,k]] * w[b,n,k]
// in fp32, each product and each sum rounded (canonical mode) or as an fma
// chain (S4G_FLAG_FMAD, what nvcc's -fmad would emit for `acc += a*b`).
//
// Lanes are consecutive dense points n (coalesced 4-byte stores per channel
// row); a lane reads its 3 indices + 3 weights ONCE (the reference re-derives
// generic-stride offsets and re-reads them for every channel) and walks a
// group of channels, gathering from one L2/L1-resident (N2 x 4 B) row at a
// time.
#include "s4g_common.h"

#include <stdlib.h>

namespace s4g {

constexpr int IP_THREADS = 256;
constexpr int IP_CG = 16;  // channels per block row

template <bool FMAD>
__global__ __launch_bounds__(IP_THREADS) void three_interpolate_kernel(
    const float* __restrict__ feat, const int64_t* __restrict__ idx,
    const float* __restrict__ w, int C, int N2, int N1,
    float* __restrict__ out) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * IP_CG;
  const int n = blockIdx.x * IP_THREADS + threadIdx.x;
  if (n >= N1) return;
  const size_t o = ((size_t)b * N1 + n) * 3;
  const int j0 = (int)idx[o], j1 = (int)idx[o + 1], j2 = (int)idx[o + 2];
  const float w0 = w[o], w1 = w[o + 1], w2 = w[o + 2];
  const int cend = min(c0 + IP_CG, C);
  for (int ch = c0; ch < cend; ++ch) {
    const float* __restrict__ src = feat + ((size_t)b * C + ch) * N2;
    float acc = 0.0f;
    if constexpr (FMAD) {
      acc = __fmaf_rn(src[j0], w0, acc);
      acc = __fmaf_rn(src[j1], w1, acc);
      acc = __fmaf_rn(src[j2], w2, acc);
    } else {
      acc = __fadd_rn(acc, __fmul_rn(src[j0], w0));
      acc = __fadd_rn(acc, __fmul_rn(src[j1], w1));
      acc = __fadd_rn(acc, __fmul_rn(src[j2], w2));
    }
    out[((size_t)b * C + ch) * N1 + n] = acc;
  }
}

// Channels-last variant (C % 4 == 0, with a (B, N2, C) workspace): the kernel above is
// bound by the texture addresser's gather rate -- three 4-byte gathers per output
// element (~2 gathers per clock and CU measured).  Transposing the sparse features once
// makes the four channels of a quad ONE 16-byte gather: a quarter of the gathers for the
// same arithmetic (each channel keeps its own rounded products and sums, so the
// results are bit-identical).
__global__ __launch_bounds__(256) void feat_to_channels_last_kernel(const float* __restrict__ in, int C,
                                                                    int N2, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* src = in + (size_t)b * C * N2;
  float* dst = out + (size_t)b * N2 * C;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int c = c0 + ty + r, n = n0 + tx;
    tile[ty + r][tx] = (c < C && n < N2) ? src[(size_t)c * N2 + n] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int n = n0 + ty + r, c = c0 + tx;
    if (c < C && n < N2) dst[(size_t)n * C + c] = tile[tx][ty + r];
  }
}

constexpr int IPQ = 32;  // channel quads (128 channels) per block row

template <bool FMAD>
__global__ __launch_bounds__(IP_THREADS) void three_interpolate_cl_kernel(
    const float* __restrict__ featT, const int64_t* __restrict__ idx, const float* __restrict__ w,
    int C, int N2, int N1, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int q0 = blockIdx.y * IPQ;
  const int n = blockIdx.x * IP_THREADS + threadIdx.x;
  if (n >= N1) return;
  const size_t o = ((size_t)b * N1 + n) * 3;
  const int j0 = (int)idx[o], j1 = (int)idx[o + 1], j2 = (int)idx[o + 2];
  const float w0 = w[o], w1 = w[o + 1], w2 = w[o + 2];
  const float* __restrict__ base = featT + (size_t)b * N2 * C;
  const float4* __restrict__ r0 = reinterpret_cast<const float4*>(base + (size_t)j0 * C);
  const float4* __restrict__ r1 = reinterpret_cast<const float4*>(base + (size_t)j1 * C);
  const float4* __restrict__ r2 = reinterpret_cast<const float4*>(base + (size_t)j2 * C);
  const int qend = min(q0 + IPQ, C / 4);
  for (int q = q0; q < qend; ++q) {
    const float4 a = r0[q], bb = r1[q], c = r2[q];
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = 0.0f;
      if constexpr (FMAD) {
        acc = __fmaf_rn(av[e], w0, acc);
        acc = __fmaf_rn(bv[e], w1, acc);
        acc = __fmaf_rn(cv[e], w2, acc);
      } else {
        acc = __fadd_rn(acc, __fmul_rn(av[e], w0));
        acc = __fadd_rn(acc, __fmul_rn(bv[e], w1));
        acc = __fadd_rn(acc, __fmul_rn(cv[e], w2));
      }
      out[((size_t)b * C + 4 * q + e) * N1 + n] = acc;
    }
  }
}

// Tile form of the channels-last kernel: a workgroup produces 64 points x 64 channels.
// Reads: 16 lanes cover the 64 channels of one neighbour row (one coalesced 256-byte
// segment per neighbour, indices / weights broadcast); the three products are summed
// in the reference's order and parked in LDS as [point][channel].  Writes: a lane
// collects 4 consecutive points of one channel and stores them as 16 bytes (64 points =
// 256 contiguous bytes per channel row): 1 KB per store instruction instead of 256 B,
// and 4x fewer gather instructions than the lane-per-point form.
#ifndef S4G_IPT_CH
#define S4G_IPT_CH 64
#endif
#ifndef S4G_IPT_PTS
#define S4G_IPT_PTS 64
#endif
constexpr int IPT_PTS = S4G_IPT_PTS, IPT_CH = S4G_IPT_CH, IPT_STRIDE = IPT_CH + 1;
constexpr int IPT_NQ = IPT_PTS / 4, IPT_CPP = 256 / IPT_NQ;  // point quads, channels per output pass
constexpr int IPT_LPR = IPT_CH / 4, IPT_RPP = 256 / IPT_LPR;  // lanes per row, rows per pass

template <bool FMAD>
__global__ __launch_bounds__(256) void three_interpolate_tile_kernel(
    const float* __restrict__ featT, const int64_t* __restrict__ idx, const float* __restrict__ w,
    int C, int N2, int N1, float* __restrict__ out, int tiles_x, int tiles_y, int nslices) {
  __shared__ float tile[IPT_PTS * IPT_STRIDE];
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs, so XCD x walks the
  // (scene, channel slice) pairs x, x + 8, ... one after the other and its L2 holds the
  // slice's rows (N2 x 64 channels) while every point tile of that slice gathers from them.
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int slice = (kk / tiles_x) * 8 + xcd;
  if (slice >= nslices) return;
  const int b = slice / tiles_y;
  const int n0 = (kk % tiles_x) * IPT_PTS;
  const int c0 = (slice % tiles_y) * IPT_CH;
  const int t = threadIdx.x;
  const int c4 = t % IPT_LPR, pr = t / IPT_LPR;
  const float* __restrict__ base = featT + (size_t)b * N2 * C;
  const bool cok = c0 + 4 * c4 < C;
#pragma unroll
  for (int i = 0; i < IPT_PTS / IPT_RPP; ++i) {
    const int r = pr + IPT_RPP * i;
    const int n = n0 + r;
    if (n < N1 && cok) {
      const size_t o = ((size_t)b * N1 + n) * 3;
      const int j0 = (int)idx[o], j1 = (int)idx[o + 1], j2 = (int)idx[o + 2];
      const float w0 = w[o], w1 = w[o + 1], w2 = w[o + 2];
      const float4 a = *reinterpret_cast<const float4*>(base + (size_t)j0 * C + c0 + 4 * c4);
      const float4 bb = *reinterpret_cast<const float4*>(base + (size_t)j1 * C + c0 + 4 * c4);
      const float4 c = *reinterpret_cast<const float4*>(base + (size_t)j2 * C + c0 + 4 * c4);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w},
                  cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = 0.0f;
        if constexpr (FMAD) {
          acc = __fmaf_rn(av[e], w0, acc);
          acc = __fmaf_rn(bv[e], w1, acc);
          acc = __fmaf_rn(cv[e], w2, acc);
        } else {
          acc = __fadd_rn(acc, __fmul_rn(av[e], w0));
          acc = __fadd_rn(acc, __fmul_rn(bv[e], w1));
          acc = __fadd_rn(acc, __fmul_rn(cv[e], w2));
        }
        tile[r * IPT_STRIDE + 4 * c4 + e] = acc;
      }
    }
  }
  __syncthreads();
  const int nq = t % IPT_NQ, cc = t / IPT_NQ;
  const int n = n0 + 4 * nq;
  const bool vec = (N1 & 3) == 0 && n + 3 < N1;
#pragma unroll
  for (int k = 0; k < IPT_CH / IPT_CPP; ++k) {
    const int ch = cc + IPT_CPP * k;
    if (c0 + ch >= C || n >= N1) continue;
    float* __restrict__ dst = out + ((size_t)b * C + c0 + ch) * N1 + n;
    const float v0 = tile[(4 * nq + 0) * IPT_STRIDE + ch], v1 = tile[(4 * nq + 1) * IPT_STRIDE + ch],
                v2 = tile[(4 * nq + 2) * IPT_STRIDE + ch], v3 = tile[(4 * nq + 3) * IPT_STRIDE + ch];
    if (vec) {
      *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
    } else {
      dst[0] = v0;
      if (n + 1 < N1) dst[1] = v1;
      if (n + 2 < N1) dst[2] = v2;
      if (n + 3 < N1) dst[3] = v3;
    }
  }
}

// group_points as the same tile: out[b][c][j] = feat[b][c][idx[b][j]], j over M*K.  The
// lane-per-output kernel in group.hip issues 64 four-byte gathers per wave instruction (64
// cache lines: the L1 tag rate bounds it); here 16 lanes read one neighbour's 64 channels
// as 256 contiguous bytes and the channel-first rows leave as 16-byte stores.
__global__ __launch_bounds__(256) void group_points_tile_kernel(
    const float* __restrict__ featT, const int64_t* __restrict__ idx, int C, int N, int64_t MK,
    float* __restrict__ out, int tiles_x, int tiles_y, int nslices) {
  __shared__ float tile[IPT_PTS * IPT_STRIDE];
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int slice = (kk / tiles_x) * 8 + xcd;
  if (slice >= nslices) return;
  const int b = slice / tiles_y;
  const int64_t n0 = (int64_t)(kk % tiles_x) * IPT_PTS;
  const int c0 = (slice % tiles_y) * IPT_CH;
  const int t = threadIdx.x;
  const int c4 = t % IPT_LPR, pr = t / IPT_LPR;
  const float* __restrict__ base = featT + (size_t)b * N * C;
  const bool cok = c0 + 4 * c4 < C;
#pragma unroll
  for (int i = 0; i < IPT_PTS / IPT_RPP; ++i) {
    const int r = pr + IPT_RPP * i;
    const int64_t n = n0 + r;
    if (n < MK && cok) {
      const int j = (int)idx[(size_t)b * MK + n];
      const float4 a = *reinterpret_cast<const float4*>(base + (size_t)j * C + c0 + 4 * c4);
      tile[r * IPT_STRIDE + 4 * c4 + 0] = a.x;
      tile[r * IPT_STRIDE + 4 * c4 + 1] = a.y;
      tile[r * IPT_STRIDE + 4 * c4 + 2] = a.z;
      tile[r * IPT_STRIDE + 4 * c4 + 3] = a.w;
    }
  }
  __syncthreads();
  const int nq = t % IPT_NQ, cc = t / IPT_NQ;
  const int64_t n = n0 + 4 * nq;
  const bool vec = (MK & 3) == 0 && n + 3 < MK;
#pragma unroll
  for (int k = 0; k < IPT_CH / IPT_CPP; ++k) {
    const int ch = cc + IPT_CPP * k;
    if (c0 + ch >= C || n >= MK) continue;
    float* __restrict__ dst = out + ((size_t)b * C + c0 + ch) * MK + n;
    const float v0 = tile[(4 * nq + 0) * IPT_STRIDE + ch], v1 = tile[(4 * nq + 1) * IPT_STRIDE + ch],
                v2 = tile[(4 * nq + 2) * IPT_STRIDE + ch], v3 = tile[(4 * nq + 3) * IPT_STRIDE + ch];
    if (vec) {
      *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
    } else {
      dst[0] = v0;
      if (n + 1 < MK) dst[1] = v1;
      if (n + 2 < MK) dst[2] = v2;
      if (n + 3 < MK) dst[3] = v3;
    }
  }
}

// Channels-last interpolate + add + bias + ReLU for the fast path's FP levels, where the
// first shared-MLP layer is applied BEFORE the interpolation (it is linear, the weights
// sum the same three rows): out[p][c] = act(y[p][c] + bias[c] + sum_k w[p][k] * s[b*N2 + idx[p][k]][c]).
// y (the skip features' share of the layer) may be NULL.  A lane owns 4 channels of a row
// (one 16-byte gather per neighbour); the wave maximum of |out| goes to a 64-slot row
// (the f16x2 contraction that consumes `out` derives its scale from it).
constexpr int IA_ROWS = 8;   // rows per lane group per workgroup pass

__global__ __launch_bounds__(256) void interp_add_cl_kernel(
    const float* __restrict__ y, const float* __restrict__ sp, const int* __restrict__ nidx,
    const float* __restrict__ nw, const float* __restrict__ bias, int64_t P, int N1, int N2, int C,
    int relu, float* __restrict__ out, uint32_t* __restrict__ out_amax) {
  const int lpr = C >> 2;                       // lanes per row
  const int rpb = 256 / lpr;                    // rows per block pass (C <= 1024)
  const int c4 = (threadIdx.x % lpr) * 4;
  const int rl = threadIdx.x / lpr;
  const bool lane_ok = rl < rpb;
  const float4 bb = *reinterpret_cast<const float4*>(bias + c4);
  float tmax = 0.f;
  for (int it = 0; it < IA_ROWS; ++it) {
    const int64_t p = ((int64_t)blockIdx.x * IA_ROWS + it) * rpb + rl;
    if (!lane_ok || p >= P) continue;
    const int b = (int)(p / N1);
    const int i0 = nidx[p * 3], i1 = nidx[p * 3 + 1], i2 = nidx[p * 3 + 2];
    const float w0 = nw[p * 3], w1 = nw[p * 3 + 1], w2 = nw[p * 3 + 2];
    const float* __restrict__ base = sp + (size_t)b * N2 * C + c4;
    const float4 a = *reinterpret_cast<const float4*>(base + (size_t)i0 * C);
    const float4 bq = *reinterpret_cast<const float4*>(base + (size_t)i1 * C);
    const float4 c = *reinterpret_cast<const float4*>(base + (size_t)i2 * C);
    float4 v = y ? *reinterpret_cast<const float4*>(y + (size_t)p * C + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w}, cv[4] = {c.x, c.y, c.z, c.w};
    const float bi[4] = {bb.x, bb.y, bb.z, bb.w};
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = __fmul_rn(av[e], w0);
      acc = __fadd_rn(acc, __fmul_rn(bv[e], w1));
      acc = __fadd_rn(acc, __fmul_rn(cv[e], w2));
      float x = __fadd_rn(__fadd_rn(o[e], acc), bi[e]);
      if (relu) x = fmaxf(x, 0.f);
      o[e] = x;
      tmax = fmaxf(tmax, fabsf(x));
    }
    *reinterpret_cast<float4*>(out + (size_t)p * C + c4) = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (out_amax) {
    // one 64-slot row per scene: the block's maximum goes to every scene its rows touch (one,
    // whenever N1 is a multiple of the block's row count)
    const uint32_t wm = wave_max_u32(__float_as_uint(tmax));
    if ((threadIdx.x & 63) == 0) {
      const int64_t r0 = (int64_t)blockIdx.x * IA_ROWS * rpb;
      const int64_t r1 = (r0 + (int64_t)IA_ROWS * rpb < P ? r0 + (int64_t)IA_ROWS * rpb : P) - 1;
      for (int64_t sc = r0 / N1; sc <= r1 / N1; ++sc)
        atomicMax(out_amax + sc * 64 + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63), wm);
    }
  }
}

__global__ __launch_bounds__(IP_THREADS) void three_interpolate_backward_kernel(
    const float* __restrict__ gout, const int64_t* __restrict__ idx,
    const float* __restrict__ w, int C, int N2, int N1,
    float* __restrict__ gin) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * IP_CG;
  const int n = blockIdx.x * IP_THREADS + threadIdx.x;
  if (n >= N1) return;
  const size_t o = ((size_t)b * N1 + n) * 3;
  const int j0 = (int)idx[o], j1 = (int)idx[o + 1], j2 = (int)idx[o + 2];
  const float w0 = w[o], w1 = w[o + 1], w2 = w[o + 2];
  const int cend = min(c0 + IP_CG, C);
  for (int ch = c0; ch < cend; ++ch) {
    float* __restrict__ dst = gin + ((size_t)b * C + ch) * N2;
    const float g = gout[((size_t)b * C + ch) * N1 + n];
    atomicAdd(dst + j0, __fmul_rn(g, w0));  // interpolate_kernel.cu:283
    atomicAdd(dst + j1, __fmul_rn(g, w1));
    atomicAdd(dst + j2, __fmul_rn(g, w2));
  }
}

}  // namespace s4g

extern "C" int s4g_three_interpolate_f32(const float* feat_bcn2,
                                         const int64_t* idx_bn3,
                                         const float* w_bn3, int64_t B,
                                         int64_t C, int64_t N2, int64_t N1,
                                         float* out_bcn1, int flags,
                                         s4g_stream_t stream) {
  if (B < 0 || C < 0 || N2 <= 0 || N1 < 0 || B > 65535 || N2 >= (1ll << 31) ||
      N1 >= (1ll << 31))
    return S4G_EINVAL;
  if (B == 0 || C == 0 || N1 == 0) return S4G_OK;
  if (!feat_bcn2 || !idx_bn3 || !w_bn3 || !out_bcn1) return S4G_EINVAL;
  const unsigned gy = (unsigned)((C + s4g::IP_CG - 1) / s4g::IP_CG);
  if (gy > 65535) return S4G_EINVAL;
  const dim3 grid((unsigned)((N1 + s4g::IP_THREADS - 1) / s4g::IP_THREADS), gy,
                  (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL((s4g::three_interpolate_kernel<true>), grid,
                       dim3(s4g::IP_THREADS), 0, st, feat_bcn2, idx_bn3, w_bn3,
                       (int)C, (int)N2, (int)N1, out_bcn1);
  else
    hipLaunchKernelGGL((s4g::three_interpolate_kernel<false>), grid,
                       dim3(s4g::IP_THREADS), 0, st, feat_bcn2, idx_bn3, w_bn3,
                       (int)C, (int)N2, (int)N1, out_bcn1);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_three_interpolate_ws_f32(const float* feat_bcn2, const int64_t* idx_bn3,
                                            const float* w_bn3, int64_t B, int64_t C, int64_t N2,
                                            int64_t N1, float* out_bcn1, void* ws, size_t ws_bytes,
                                            int flags, s4g_stream_t stream) {
  // anything the channels-last form does not cover is the plain call: same results
  if (!ws || ws_bytes < (size_t)B * (size_t)N2 * (size_t)C * sizeof(float) || (C & 3) != 0 ||
      ((uintptr_t)ws & 15) != 0 || B <= 0 || B > 65535 || C <= 0 || N1 <= 0 || N2 <= 0 ||
      N2 >= (1ll << 31) || N1 >= (1ll << 31) || (C + 31) / 32 > 65535)
    return s4g_three_interpolate_f32(feat_bcn2, idx_bn3, w_bn3, B, C, N2, N1, out_bcn1, flags, stream);
  if (!feat_bcn2 || !idx_bn3 || !w_bn3 || !out_bcn1) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* featT = (float*)ws;
  hipLaunchKernelGGL(s4g::feat_to_channels_last_kernel,
                     dim3((unsigned)((N2 + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B), dim3(256), 0,
                     st, feat_bcn2, (int)C, (int)N2, featT);
  S4G_LAUNCH_CHECK();
  const bool tiled = s4g::knob("S4G_INTERP_MODE") == nullptr || s4g::knob("S4G_INTERP_MODE")[0] != 'l';
  if (tiled && ((uintptr_t)out_bcn1 & 15) == 0) {
    const int tiles_x = (int)((N1 + s4g::IPT_PTS - 1) / s4g::IPT_PTS);
    const int tiles_y = (int)((C + s4g::IPT_CH - 1) / s4g::IPT_CH);
    const int nslices = tiles_y * (int)B;
    const dim3 tgrid((unsigned)((nslices + 7) / 8 * 8) * (unsigned)tiles_x);
    if (flags & S4G_FLAG_FMAD)
      hipLaunchKernelGGL((s4g::three_interpolate_tile_kernel<true>), tgrid, dim3(256), 0, st, featT,
                         idx_bn3, w_bn3, (int)C, (int)N2, (int)N1, out_bcn1, tiles_x, tiles_y, nslices);
    else
      hipLaunchKernelGGL((s4g::three_interpolate_tile_kernel<false>), tgrid, dim3(256), 0, st, featT,
                         idx_bn3, w_bn3, (int)C, (int)N2, (int)N1, out_bcn1, tiles_x, tiles_y, nslices);
    S4G_LAUNCH_CHECK();
    return S4G_OK;
  }
  const dim3 grid((unsigned)((N1 + s4g::IP_THREADS - 1) / s4g::IP_THREADS),
                  (unsigned)((C / 4 + s4g::IPQ - 1) / s4g::IPQ), (unsigned)B);
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL((s4g::three_interpolate_cl_kernel<true>), grid, dim3(s4g::IP_THREADS), 0, st,
                       featT, idx_bn3, w_bn3, (int)C, (int)N2, (int)N1, out_bcn1);
  else
    hipLaunchKernelGGL((s4g::three_interpolate_cl_kernel<false>), grid, dim3(s4g::IP_THREADS), 0, st,
                       featT, idx_bn3, w_bn3, (int)C, (int)N2, (int)N1, out_bcn1);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_three_interpolate_backward_f32(
    const float* gout_bcn1, const int64_t* idx_bn3, const float* w_bn3,
    int64_t B, int64_t C, int64_t N2, int64_t N1, float* gin_bcn2,
    s4g_stream_t stream) {
  if (B < 0 || C < 0 || N2 <= 0 || N1 < 0 || B > 65535 || N2 >= (1ll << 31) ||
      N1 >= (1ll << 31))
    return S4G_EINVAL;
  if (B == 0 || C == 0) return S4G_OK;
  if (!gin_bcn2) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e =
      hipMemsetAsync(gin_bcn2, 0, sizeof(float) * (size_t)(B * C * N2), st);
  if (e != hipSuccess) return (int)e;
  if (N1 == 0) return S4G_OK;
  if (!gout_bcn1 || !idx_bn3 || !w_bn3) return S4G_EINVAL;
  const unsigned gy = (unsigned)((C + s4g::IP_CG - 1) / s4g::IP_CG);
  if (gy > 65535) return S4G_EINVAL;
  const dim3 grid((unsigned)((N1 + s4g::IP_THREADS - 1) / s4g::IP_THREADS), gy,
                  (unsigned)B);
  hipLaunchKernelGGL(s4g::three_interpolate_backward_kernel, grid,
                     dim3(s4g::IP_THREADS), 0, st, gout_bcn1, idx_bn3, w_bn3,
                     (int)C, (int)N2, (int)N1, gin_bcn2);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_group_points_ws_f32(const float* feat_bcn, const int64_t* idx_bmk, int64_t B,
                                       int64_t C, int64_t N, int64_t M, int64_t K, float* out_bcmk,
                                       void* ws, size_t ws_bytes, s4g_stream_t stream) {
  const int64_t MK = M * K;
  const int64_t tiles_x = (MK + s4g::IPT_PTS - 1) / s4g::IPT_PTS;
  const int64_t tiles_y = (C + s4g::IPT_CH - 1) / s4g::IPT_CH;
  // anything the channels-last form does not cover is the plain call: same results
  if (!ws || ws_bytes < (size_t)B * (size_t)N * (size_t)C * sizeof(float) || (C & 3) != 0 ||
      ((uintptr_t)ws & 15) != 0 || ((uintptr_t)out_bcmk & 15) != 0 || B <= 0 || B > 65535 || C <= 0 ||
      N <= 0 || N >= (1ll << 31) || MK <= 0 || (C + 31) / 32 > 65535 ||
      (tiles_y * B + 7) / 8 * 8 * tiles_x >= (1ll << 31))
    return s4g_group_points_f32(feat_bcn, idx_bmk, B, C, N, M, K, out_bcmk, stream);
  if (!feat_bcn || !idx_bmk || !out_bcmk) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* featT = (float*)ws;
  hipLaunchKernelGGL(s4g::feat_to_channels_last_kernel,
                     dim3((unsigned)((N + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B), dim3(256), 0,
                     st, feat_bcn, (int)C, (int)N, featT);
  S4G_LAUNCH_CHECK();
  const int nslices = (int)(tiles_y * B);
  const dim3 tgrid((unsigned)((nslices + 7) / 8 * 8) * (unsigned)tiles_x);
  hipLaunchKernelGGL(s4g::group_points_tile_kernel, tgrid, dim3(256), 0, st, featT, idx_bmk, (int)C,
                     (int)N, MK, out_bcmk, (int)tiles_x, (int)tiles_y, nslices);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_interp_add_cl_f32(const float* y_pc, const float* sparse_rc, const int32_t* nidx_p3,
                                     const float* nw_p3, const float* bias_c, int64_t B, int64_t N1,
                                     int64_t N2, int64_t C, int relu, float* out_pc, float* out_amax64,
                                     s4g_stream_t stream) {
  if (B < 0 || N1 < 0 || N2 <= 0 || C <= 0 || (C & 3) || C > 1024 ||
      N2 >= (1ll << 31) || N1 >= (1ll << 31))
    return S4G_EINVAL;
  if (B == 0 || N1 == 0) return S4G_OK;
  if (!sparse_rc || !nidx_p3 || !nw_p3 || !bias_c || !out_pc) return S4G_EINVAL;
  if (((uintptr_t)sparse_rc | (uintptr_t)out_pc | (uintptr_t)bias_c | (uintptr_t)y_pc) & 15) return S4G_EINVAL;
  const int64_t P = B * N1;
  const int rpb = 256 / (int)(C / 4);
  const int64_t per_block = (int64_t)rpb * s4g::IA_ROWS;
  const int64_t blocks = (P + per_block - 1) / per_block;
  if (blocks >= (1ll << 31)) return S4G_EINVAL;
  hipLaunchKernelGGL(s4g::interp_add_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     y_pc, sparse_rc, nidx_p3, nw_p3, bias_c, P, (int)N1, (int)N2, (int)C, relu, out_pc,
                     reinterpret_cast<uint32_t*>(out_amax64));
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}
