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
#include "grid.h"

#include <stdlib.h>

namespace s4g {

constexpr int NN_THREADS = 256;

// A slot that no key ever entered (a query with a NaN / inf coordinate, or coordinates so large that every squared
// distance overflows: `d < best` is never true) keeps the reference's distance, but its index is written as 0 where the
// reference leaves its initialiser -1 (interpolate_kernel.cu:54) and the grid search its internal sentinel: the
// consumers -- three_interpolate, its backward scatter, the fused loaders -- index feature rows with it, and an
// out-of-range row is a GPU memory fault that takes the whole process down (found by
// tests/test_batch_invariance_gpu.py::test_nonfinite_scene_is_contained_and_refused_on_request).  In-contract inputs
// (finite coordinates, N2 >= 3) fill all three slots: nothing changes for them, bit for bit.
__device__ __forceinline__ int nn_safe_index(int j, int N2) { return (unsigned)j < (unsigned)N2 ? j : 0; }

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
    idx[o + 0] = (IdxT)nn_safe_index(i0, N2);
    idx[o + 1] = (IdxT)nn_safe_index(i1, N2);
    idx[o + 2] = (IdxT)nn_safe_index(i2, N2);
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

// ---------------------------------------------------------------------------
// GRID path (fast path of the FP layers): the keys are binned into the same
// toroidal cell grid the ball query uses (grid.h) and every query looks only at
// the 27 cells around it -- ~30-100 candidates instead of all N2 keys.
// The reference result is "the 3 smallest by (d, key index)" (strict '<' while
// scanning keys in index order), so candidates may be visited in
// any order as long as ties are broken on the index.  A query is DONE when it
// has three candidates and its third distance is below (cell * (1 - 1e-3))^2:
// every key outside the 27-cell block is at least one cell edge away, hence
// strictly farther.  Anything else (isolated points, out-of-range scenes) goes
// to a list that a second kernel answers with the literal index-order scan, one
// wave per query -- so the result is exact for every input.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool nn_less(float d, int j, float bd, int bj) {
  return d < bd || (d == bd && j < bj);
}

template <bool WEIGHTS, typename IdxT>
__device__ __forceinline__ void nn_write(IdxT* __restrict__ idx, float* __restrict__ out, size_t o,
                                         int i0, int i1, int i2, float b0, float b1, float b2,
                                         float eps, int N2) {
  idx[o + 0] = (IdxT)nn_safe_index(i0, N2);
  idx[o + 1] = (IdxT)nn_safe_index(i1, N2);
  idx[o + 2] = (IdxT)nn_safe_index(i2, N2);
  if constexpr (WEIGHTS) {
    const float ia = __fdiv_rn(1.0f, b0 < eps ? eps : b0);
    const float ib = __fdiv_rn(1.0f, b1 < eps ? eps : b1);
    const float ic = __fdiv_rn(1.0f, b2 < eps ? eps : b2);
    const float s = __fadd_rn(__fadd_rn(ia, ib), ic);
    out[o + 0] = __fdiv_rn(ia, s);
    out[o + 1] = __fdiv_rn(ib, s);
    out[o + 2] = __fdiv_rn(ic, s);
  } else {
    out[o + 0] = b0;
    out[o + 1] = b1;
    out[o + 2] = b2;
  }
}

// ---------------------------------------------------------------------------
// SPLIT scan for small key sets (24 <= N2 <= 2048: the two coarse FP levels).  The
// lane-per-query scan above is a serial chain of N2 steps on ~1 wave per SIMD (FP2 size,
// 5 120 x 1 024: 0.23 ms at 18 GB/s).  Here S lanes share a query: the keys sit in LDS as
// 16-byte records, sub-lane s scans keys s, s + S, s + 2 S, ... (ascending inside a lane, so
// the strict '<' insertion keeps the earlier key among equals exactly like the reference,
// interpolate_kernel.cu:63-73), and the S partial triples are merged by a xor butterfly with
// the (distance, key index) order -- the same three keys in the same order as one index-order
// scan.  N2 / S steps instead of N2, S times as many waves.
// ---------------------------------------------------------------------------
template <bool FMAD, bool WEIGHTS, typename IdxT, int S>
__global__ __launch_bounds__(NN_THREADS) void three_nn_split_kernel(
    const float* __restrict__ q, const float* __restrict__ key, int N1, int N2, float eps,
    IdxT* __restrict__ idx, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float4 keys[];
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const float* __restrict__ kx = key + (size_t)b * 3 * N2;
  for (int j = t; j < N2; j += NN_THREADS) keys[j] = make_float4(kx[j], kx[N2 + j], kx[2 * N2 + j], 0.f);
  __syncthreads();
  constexpr int QPB = NN_THREADS / S;
  const int i = blockIdx.x * QPB + t / S;
  const int sub = t % S;
  const float* __restrict__ qx = q + (size_t)b * 3 * N1;
  const int ii = i < N1 ? i : N1 - 1;
  const float x1 = qx[ii], y1 = qx[N1 + ii], z1 = qx[2 * N1 + ii];
  float b0 = __builtin_inff(), b1 = b0, b2 = b0;
  int i0 = -1, i1 = -1, i2 = -1;
  for (int j = sub; j < N2; j += S) {
    const float4 k = keys[j];
    const float d = dist2<FMAD>(k.x, k.y, k.z, x1, y1, z1);
    const bool c0 = d < b0, c1 = d < b1, c2 = d < b2;
    const bool c01 = c0 | c1;
    b2 = c01 ? b1 : (c2 ? d : b2);
    i2 = c01 ? i1 : (c2 ? j : i2);
    b1 = c0 ? b0 : (c1 ? d : b1);
    i1 = c0 ? i0 : (c1 ? j : i1);
    b0 = c0 ? d : b0;
    i0 = c0 ? j : i0;
  }
#pragma unroll
  for (int off = 1; off < S; off <<= 1) {
    const float pb[3] = {__shfl_xor(b0, off), __shfl_xor(b1, off), __shfl_xor(b2, off)};
    const int pi[3] = {__shfl_xor(i0, off), __shfl_xor(i1, off), __shfl_xor(i2, off)};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const float d = pb[e];
      const int j = pi[e];
      const bool c0 = nn_less(d, j, b0, i0), c1 = nn_less(d, j, b1, i1), c2 = nn_less(d, j, b2, i2);
      const bool c01 = c0 | c1;
      b2 = c01 ? b1 : (c2 ? d : b2);
      i2 = c01 ? i1 : (c2 ? j : i2);
      b1 = c0 ? b0 : (c1 ? d : b1);
      i1 = c0 ? i0 : (c1 ? j : i1);
      b0 = c0 ? d : b0;
      i0 = c0 ? j : i0;
    }
  }
  if (i < N1 && sub == 0)
    nn_write<WEIGHTS, IdxT>(idx, out, ((size_t)b * N1 + i) * 3, i0, i1, i2, b0, b1, b2, eps, N2);
}

// lane-per-query scan or, for the small key sets, the split scan (S4G_NN_SPLIT=0: never)
template <bool FMAD, bool WEIGHTS, typename IdxT>
static int launch_three_nn_scan(const float* q, const float* k, int64_t B, int64_t N1, int64_t N2,
                                float eps, IdxT* idx, float* out, hipStream_t st) {
  const char* e = s4g::knob("S4G_NN_SPLIT");
  const bool split = !(e && e[0] == '0') && N2 >= 24 && N2 <= 2048;
  if (split) {
    const size_t lds = sizeof(float4) * (size_t)N2;
    if (N2 >= 256) {
      const dim3 grid((unsigned)((N1 + NN_THREADS / 8 - 1) / (NN_THREADS / 8)), (unsigned)B);
      hipLaunchKernelGGL((three_nn_split_kernel<FMAD, WEIGHTS, IdxT, 8>), grid, dim3(NN_THREADS), lds, st, q, k,
                         (int)N1, (int)N2, eps, idx, out);
    } else {
      const dim3 grid((unsigned)((N1 + NN_THREADS / 4 - 1) / (NN_THREADS / 4)), (unsigned)B);
      hipLaunchKernelGGL((three_nn_split_kernel<FMAD, WEIGHTS, IdxT, 4>), grid, dim3(NN_THREADS), lds, st, q, k,
                         (int)N1, (int)N2, eps, idx, out);
    }
  } else {
    const dim3 grid((unsigned)((N1 + NN_THREADS - 1) / NN_THREADS), (unsigned)B);
    hipLaunchKernelGGL((three_nn_kernel<FMAD, WEIGHTS, IdxT>), grid, dim3(NN_THREADS), 0, st, q, k, (int)N1,
                       (int)N2, eps, idx, out);
  }
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

// Cell edge for the operator-API grid search, chosen on the device (no host read):
// `factor` x the median distance from a key to its third-nearest other key, measured on 64
// sample keys spread over the batch (one wave per sample scans its scene's keys, lanes
// keep their four smallest squared distances, four wave-min rounds merge them).  The
// search cost grows with (cell / key spacing)^2..3 and every query whose third neighbour
// is farther than one cell pays an O(N2) scan, so the spacing has to be measured: a
// bounding-box estimate is off by 3x on surface-like clouds.  ANY value gives the same
// results; out = (1 / cell, (cell (1 - 1e-3))^2), or (1, -1) = "accept nothing" when no
// positive finite spacing comes out (all keys coincide, NaN / inf coordinates).
constexpr int NN_SAMPLES = 64;

// hdr: the fail list's header -- word 0 the fail count, 4 / 5 the result (1 / cell, bound), 8 the arrival
// counter, 16 .. 16 + NN_SAMPLES the samples' third-neighbour distances.  One wave per sample (round 4: the
// first version ran all 64 scans in ONE workgroup, 0.09 ms of latency per call); the last workgroup to
// arrive sums the samples in index order, so the edge is the same run to run.
constexpr int NN_HDR_WORDS = 16 + NN_SAMPLES;

__global__ __launch_bounds__(64) void nn_auto_cell_kernel(const float* __restrict__ key, int B, int N2,
                                                          int* __restrict__ hdr, float factor) {
  const int lane = threadIdx.x, s = blockIdx.x;
  float* __restrict__ d3s = reinterpret_cast<float*>(hdr + 16);
  float* __restrict__ out = reinterpret_cast<float*>(hdr + 4);
  {
    const int b = s % B;
    const int j = (int)(((unsigned long long)s * 2654435761ull + 12345ull) % (unsigned)N2);
    const float* __restrict__ kx = key + (size_t)b * 3 * N2;
    const float x = kx[j], y = kx[N2 + j], z = kx[2 * N2 + j];
    float a0 = __builtin_inff(), a1 = a0, a2 = a0, a3 = a0;   // ascending
    // eight keys per lane in flight: the scan is a chain of L2 round trips otherwise (80 of them for 5 120 keys)
    for (int i0 = lane; i0 < N2; i0 += 64 * 8) {
      float kxv[8], kyv[8], kzv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u;
        const int ic = i < N2 ? i : j;          // past the end: the sample itself (distance 0 is discarded below)
        kxv[u] = kx[ic];
        kyv[u] = kx[N2 + ic];
        kzv[u] = kx[2 * N2 + ic];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float dx = kxv[u] - x, dy = kyv[u] - y, dz = kzv[u] - z;
        const float d = i0 + 64 * u < N2 ? dx * dx + dy * dy + dz * dz : __builtin_inff();
        if (d < a3) {
          a3 = d;
          if (a3 < a2) { const float q = a2; a2 = a3; a3 = q; }
          if (a2 < a1) { const float q = a1; a1 = a2; a2 = q; }
          if (a1 < a0) { const float q = a0; a0 = a1; a1 = q; }
        }
      }
    }
    float kth = 0.f;   // 4th smallest over the wave = third-nearest OTHER key (self is 0)
    for (int r = 0; r < 4; ++r) {
      const float m = __uint_as_float(wave_min_u32(__float_as_uint(a0)));   // d >= 0: bit order = value order
      kth = m;
      const uint64_t own = __ballot(a0 == m);
      if (lane == (int)(__ffsll((unsigned long long)own) - 1)) {
        a0 = a1;
        a1 = a2;
        a2 = a3;
        a3 = __builtin_inff();
      }
    }
    if (lane == 0) {
      __hip_atomic_store(d3s + s, kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
    }
  }
  int last = 0;
  if (lane == 0) last = atomicAdd(hdr + 8, 1) == NN_SAMPLES - 1;
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  __threadfence();
  {
    // the MEDIAN of the valid samples (lane q holds sample q; its rank by counting): a few keys that are far
    // outliers -- FPS picks those first, so they ARE keys -- would drag a mean to an edge many times too large
    const float d = __hip_atomic_load(d3s + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = d > 0.f && d < __builtin_inff();
    const int n = __popcll(__ballot(ok));
    int rank = 0;
    for (int q = 0; q < NN_SAMPLES; ++q) {
      const float dq = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), q));
      const bool okq = dq > 0.f && dq < __builtin_inff();
      rank += (okq && (dq < d || (dq == d && q < lane))) ? 1 : 0;
    }
    const uint64_t mid = __ballot(ok && rank == n / 2);
    float cell = 0.f;
    if (mid) cell = factor * sqrtf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), (int)(__ffsll((unsigned long long)mid) - 1))));
    if (lane == 0) {
      if (!(cell > 0.f) || !(cell < 1e18f)) {
        out[0] = 1.0f;
        out[1] = -1.0f;
      } else {
        const float edge = cell * (1.0f - 1e-3f);
        out[0] = 1.0f / cell;
        out[1] = edge * edge;
      }
    }
  }
}

// Lane per query, queries taken in CELL order (the build launch bins them into the keys' grid,
// `cw`): the 64 queries of a wave sit in a few x-adjacent cells, so their 9 key rows are the
// same lines (coalesced / broadcast loads, equal trip counts) instead of 64 unrelated walks.
// Per query: the 9 rows' record ranges up front (18 independent loads, the rare x-wrap pieces
// apart), then the rows in groups of three with four range-checked buffer loads in flight per
// row; the insertion is branch-free (a (d, index)-ordered triple, as the scan kernels keep it).
template <bool FMAD, bool WEIGHTS, typename IdxT>
__global__ __launch_bounds__(NN_THREADS) void three_nn_grid_kernel(
    const float* __restrict__ key, int N1, int N2, float inv_h, float d2_done, GridWs ws,
    CellWs cw, float eps, IdxT* __restrict__ idx, float* __restrict__ out,
    int* __restrict__ fail_list, int* __restrict__ fail_count,
    const float* __restrict__ cell_dev = nullptr) {
  if (cell_dev) {  // (1 / cell, acceptance bound) chosen by nn_auto_cell_kernel
    inv_h = cell_dev[0];
    d2_done = cell_dev[1];
  }
  const int b = blockIdx.y;
  int r = blockIdx.x * NN_THREADS + threadIdx.x;
  if (r >= N1) return;
  // record r of the scene's cell-ordered queries: 8 stripes, each with its own count
  const int* __restrict__ cnt = cw.ncell + ((size_t)gridDim.y + b) * GR_RANGES;
  int g = 0;
#pragma unroll
  for (int k = 0; k < GR_RANGES - 1; ++k) {
    const int c = cnt[k];
    if (g == k && r >= c) {
      r -= c;
      g = k + 1;
    }
  }
  const float4 qr = cw.sorted[((size_t)b * GR_RANGES + g) * N1 + r];
  const int i = __float_as_int(qr.w);
  const float x1 = qr.x, y1 = qr.y, z1 = qr.z;
  const float* __restrict__ kx = key + (size_t)b * 3 * N2;
  const float ox = kx[0], oy = kx[N2], oz = kx[2 * N2];
  const bool exact = ws.flags[b] == 0 && grid_coord_ok(x1, ox, inv_h) &&
                     grid_coord_ok(y1, oy, inv_h) && grid_coord_ok(z1, oz, inv_h);
  float b0 = __builtin_inff(), b1 = __builtin_inff(), b2 = __builtin_inff();
  int i0 = 0x7FFFFFFF, i1 = 0x7FFFFFFF, i2 = 0x7FFFFFFF;
  if (exact) {
    const int icx = grid_coord(x1, ox, inv_h), icy = grid_coord(y1, oy, inv_h),
              icz = grid_coord(z1, oz, inv_h);
    const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(ws.sorted + (size_t)b * GR_RANGES * N2), 0, GR_RANGES * N2 * 16, 0x00020000);
    const int* __restrict__ starts_b = ws.starts + (size_t)b * GR_RANGES * GR_START_STRIDE;
    const int x0 = (icx - 1) & 31;
    const bool wrap = x0 > GR_DIM - 3;
    const int xe = wrap ? GR_DIM : x0 + 3;
    int beg[9], n0[9], beg1[9], n1[9];
#pragma unroll
    for (int rr = 0; rr < 9; ++rr) {
      const int zz = (icz + rr / 3 - 1) & 31, yy = (icy + rr % 3 - 1) & 31;
      const int* __restrict__ st = starts_b + grid_range(yy, zz) * GR_START_STRIDE +
                                   grid_local_row(yy, zz);
      beg[rr] = st[x0];
      n0[rr] = st[xe] - beg[rr];
      beg1[rr] = 0;
      n1[rr] = 0;
    }
    if (__ballot(wrap) != 0ull) {  // the window crosses the torus seam: a second piece
#pragma unroll
      for (int rr = 0; rr < 9; ++rr) {
        const int zz = (icz + rr / 3 - 1) & 31, yy = (icy + rr % 3 - 1) & 31;
        const int* __restrict__ st = starts_b + grid_range(yy, zz) * GR_START_STRIDE +
                                     grid_local_row(yy, zz);
        beg1[rr] = st[0];
        n1[rr] = wrap ? st[(x0 + 3) & 31] - beg1[rr] : 0;
      }
    }
    auto visit = [&](const float4 p, bool valid) {
      const int j = valid ? __float_as_int(p.w) : 0x7FFFFFFF;
      float d = dist2<FMAD>(p.x, p.y, p.z, x1, y1, z1);
      d = valid ? d : __builtin_inff();
      const bool c0 = nn_less(d, j, b0, i0), c1 = nn_less(d, j, b1, i1), c2 = nn_less(d, j, b2, i2);
      const bool c01 = c0 | c1;
      b2 = c01 ? b1 : (c2 ? d : b2);
      i2 = c01 ? i1 : (c2 ? j : i2);
      b1 = c0 ? b0 : (c1 ? d : b1);
      i1 = c0 ? i0 : (c1 ? j : i1);
      b0 = c0 ? d : b0;
      i0 = c0 ? j : i0;
    };
    // element t of a row: piece 0 then piece 1; past the end -> an out-of-range offset (reads 0)
    auto rec_off = [&](int rr, int t) -> int {
      const int j = t < n0[rr] ? beg[rr] + t : beg1[rr] + (t - n0[rr]);
      return t < n0[rr] + n1[rr] ? j * 16 : 0x7FFFFFF0;
    };
#pragma unroll
    for (int grp = 0; grp < 3; ++grp) {
      float4 p[12];
#pragma unroll
      for (int u = 0; u < 12; ++u)
        p[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                              rrec, rec_off(grp * 3 + u / 4, u % 4), 0, 0));
#pragma unroll
      for (int u = 0; u < 12; ++u)  // pinned: no sinking of a load into its use
        asm volatile("" : "+v"(p[u].x), "+v"(p[u].y), "+v"(p[u].z), "+v"(p[u].w));
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int rr = grp * 3 + u / 4;
        visit(p[u], (u % 4) < n0[rr] + n1[rr]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int rr = grp * 3 + k;
        const int n = n0[rr] + n1[rr];
        for (int t = 4; t < n; t += 4) {
          float4 q4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            q4[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rrec, rec_off(rr, t + u), 0, 0));
#pragma unroll
          for (int u = 0; u < 4; ++u)
            asm volatile("" : "+v"(q4[u].x), "+v"(q4[u].y), "+v"(q4[u].z), "+v"(q4[u].w));
#pragma unroll
          for (int u = 0; u < 4; ++u) visit(q4[u], t + u < n);
        }
      }
    }
  }
  const bool done = exact && i2 != 0x7FFFFFFF && b2 < d2_done;
  if (done) {
    nn_write<WEIGHTS, IdxT>(idx, out, ((size_t)b * N1 + i) * 3, i0, i1, i2, b0, b1, b2, eps, N2);
  } else {
    fail_list[atomicAdd(fail_count, 1)] = b * N1 + i;
  }
}

// One workgroup per unanswered query (a handful per batch, so the walk over all keys is pure
// latency: four waves x eight strides of loads in flight per round trip).  Lanes keep their own
// (d, index)-ordered triple, three wave-argmin rounds merge a wave's, wave 0 merges the four.
template <bool FMAD, bool WEIGHTS, typename IdxT>
__global__ __launch_bounds__(NN_THREADS) void three_nn_fallback_kernel(
    const float* __restrict__ q, const float* __restrict__ key, int N1, int N2, float eps,
    IdxT* __restrict__ idx, float* __restrict__ out, const int* __restrict__ fail_list,
    const int* __restrict__ fail_count) {
  constexpr int NW = NN_THREADS / 64;
  __shared__ float sd[NW * 3];
  __shared__ int sj[NW * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int count = *fail_count;
  auto insert = [](float d, int j, float& b0, float& b1, float& b2, int& i0, int& i1, int& i2) {
    const bool c0 = nn_less(d, j, b0, i0), c1 = nn_less(d, j, b1, i1), c2 = nn_less(d, j, b2, i2);
    const bool c01 = c0 | c1;
    b2 = c01 ? b1 : (c2 ? d : b2);
    i2 = c01 ? i1 : (c2 ? j : i2);
    b1 = c0 ? b0 : (c1 ? d : b1);
    i1 = c0 ? i0 : (c1 ? j : i1);
    b0 = c0 ? d : b0;
    i0 = c0 ? j : i0;
  };
  // pops the wave's three smallest heads by (d, index) into rd / rj (wave-uniform)
  auto merge3 = [&](float& b0, float& b1, float& b2, int& i0, int& i1, int& i2, float* rd, int* rj) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // d >= 0 so its bits order like u32
      const uint32_t dm = wave_min_u32(__float_as_uint(b0));
      const uint32_t jm = wave_min_u32(__float_as_uint(b0) == dm ? (uint32_t)i0 : 0xFFFFFFFFu);
      rd[k] = __uint_as_float(dm);
      rj[k] = (int)jm;
      if (__float_as_uint(b0) == dm && (uint32_t)i0 == jm) {  // pop this lane's head
        b0 = b1; i0 = i1; b1 = b2; i1 = i2; b2 = __builtin_inff(); i2 = 0x7FFFFFFF;
      }
    }
  };
  for (int f = blockIdx.x; f < count; f += gridDim.x) {
    const int qi = fail_list[f];
    const int b = qi / N1, i = qi - b * N1;
    const float* __restrict__ qx = q + (size_t)b * 3 * N1;
    const float* __restrict__ kx = key + (size_t)b * 3 * N2;
    const float x1 = qx[i], y1 = qx[N1 + i], z1 = qx[2 * N1 + i];
    float b0 = __builtin_inff(), b1 = __builtin_inff(), b2 = __builtin_inff();
    int i0 = 0x7FFFFFFF, i1 = 0x7FFFFFFF, i2 = 0x7FFFFFFF;
    constexpr int FU = 8;
    for (int j0 = threadIdx.x; j0 < N2; j0 += NN_THREADS * FU) {
      float px[FU], py[FU], pz[FU];
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const int j = j0 + NN_THREADS * u < N2 ? j0 + NN_THREADS * u : 0;
        px[u] = kx[j];
        py[u] = kx[N2 + j];
        pz[u] = kx[2 * N2 + j];
      }
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const bool valid = j0 + NN_THREADS * u < N2;
        const float d = dist2<FMAD>(px[u], py[u], pz[u], x1, y1, z1);
        insert(valid ? d : __builtin_inff(), valid ? j0 + NN_THREADS * u : 0x7FFFFFFF, b0, b1, b2,
               i0, i1, i2);
      }
    }
    float rd[3];
    int rj[3];
    merge3(b0, b1, b2, i0, i1, i2, rd, rj);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sd[wave * 3 + k] = rd[k];
        sj[wave * 3 + k] = rj[k];
      }
    }
    __syncthreads();
    if (wave == 0) {
      b0 = lane < NW * 3 ? sd[lane] : __builtin_inff();
      i0 = lane < NW * 3 ? sj[lane] : 0x7FFFFFFF;
      b1 = b2 = __builtin_inff();
      i1 = i2 = 0x7FFFFFFF;
      merge3(b0, b1, b2, i0, i1, i2, rd, rj);
      if (lane == 0)
        nn_write<WEIGHTS, IdxT>(idx, out, ((size_t)b * N1 + i) * 3, rj[0], rj[1], rj[2], rd[0],
                                rd[1], rd[2], eps, N2);
    }
    __syncthreads();
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
  hipStream_t st = (hipStream_t)stream;
  if (flags & S4G_FLAG_FMAD)
    return s4g::launch_three_nn_scan<true, false, int64_t>(q_b3n1, k_b3n2, B, N1, N2, 0.f, idx_bn3, d2_bn3, st);
  return s4g::launch_three_nn_scan<false, false, int64_t>(q_b3n1, k_b3n2, B, N1, N2, 0.f, idx_bn3, d2_bn3, st);
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
  hipStream_t st = (hipStream_t)stream;
  if (flags & S4G_FLAG_FMAD)
    return s4g::launch_three_nn_scan<true, true, int32_t>(q_b3n1, k_b3n2, B, N1, N2, eps, idx_bn3, w_bn3, st);
  return s4g::launch_three_nn_scan<false, true, int32_t>(q_b3n1, k_b3n2, B, N1, N2, eps, idx_bn3, w_bn3, st);
}

// workspace: [keys' grid][fail header (16 ints) + fail list][queries binned into the keys' grid]
static size_t nn_fail_bytes(int64_t B, int64_t N1) {
  return (sizeof(int) * ((size_t)B * N1 + s4g::NN_HDR_WORDS) + 63) & ~(size_t)63;
}

extern "C" size_t s4g_three_nn_grid_workspace_bytes(int64_t B, int64_t N1, int64_t N2) {
  return s4g::grid_ws_bytes(B, N2) + nn_fail_bytes(B, N1) + s4g::cell_ws_bytes(B, N1);
}

// Diagnostic: byte offset of the fail list's header inside that workspace (int32 words: 0 = queries the 27 cells
// could not answer in the last call, 4 / 5 = the device-chosen 1 / edge and acceptance bound as floats).
extern "C" size_t s4g_three_nn_grid_header_offset(int64_t B, int64_t N2) { return s4g::grid_ws_bytes(B, N2); }

namespace s4g {

// build (keys + queries, one launch) -> cell-ordered lane-per-query search -> scan of the
// queries it could not prove; cell < 0: the edge is chosen on the device first.
template <bool WEIGHTS, typename IdxT>
static int launch_three_nn_grid(const float* q, const float* k, int64_t B, int64_t N1, int64_t N2,
                                float eps, float cell, IdxT* idx, float* out, void* ws,
                                size_t ws_bytes, int flags, hipStream_t st) {
  const bool auto_cell = cell < 0.f;
  if (B < 0 || N1 < 0 || N2 < 3 || B > 65535 || N2 > GR_MAX_POINTS || N1 >= (1ll << 31) ||
      B * N1 >= (1ll << 31) || (!auto_cell && (!(cell > 0.f) || !(cell < 1e18f))))
    return S4G_EINVAL;
  if (B == 0 || N1 == 0) return S4G_OK;
  if (!q || !k || !idx || !out || !ws) return S4G_EINVAL;
  if (ws_bytes < s4g_three_nn_grid_workspace_bytes(B, N1, N2)) return S4G_EWORKSPACE;
  const GridWs g = grid_ws_carve(ws, B, N2);
  int* fail_count = reinterpret_cast<int*>((char*)ws + grid_ws_bytes(B, N2));
  int* fail_list = fail_count + NN_HDR_WORDS;
  const CellWs cw = cell_ws_carve((char*)fail_count + nn_fail_bytes(B, N1), B, N1);
  hipError_t e = hipMemsetAsync(fail_count, 0, sizeof(int) * 16, st);   // fail count + the edge search's arrival counter
  if (e != hipSuccess) return (int)e;
  float* cell_dev = nullptr;   // header words 4, 5 of the fail list
  if (auto_cell) {
    cell_dev = reinterpret_cast<float*>(fail_count + 4);
    static const float factor = [] { const char* e = s4g::knob("S4G_NN_CELL_FACTOR"); return e ? (float)atof(e) : 1.75f; }();
    hipLaunchKernelGGL(nn_auto_cell_kernel, dim3(NN_SAMPLES), dim3(64), 0, st, k, (int)B, (int)N2, fail_count, factor);
    S4G_LAUNCH_CHECK();
    cell = 1.0f;
  }
  const float inv_h = 1.0f / cell;
  if (int rc = launch_grid_build_queries(k, q, B, N2, N1, inv_h, g, cw, st, false, cell_dev)) return rc;
  const float edge = cell * (1.0f - 1e-3f);
  const float d2_done = edge * edge;
  const dim3 grid((unsigned)((N1 + NN_THREADS - 1) / NN_THREADS), (unsigned)B);
  const bool fmad = (flags & S4G_FLAG_FMAD) != 0;
  if (fmad)
    hipLaunchKernelGGL((three_nn_grid_kernel<true, WEIGHTS, IdxT>), grid, dim3(NN_THREADS), 0, st,
                       k, (int)N1, (int)N2, inv_h, d2_done, g, cw, eps, idx, out, fail_list,
                       fail_count, (const float*)cell_dev);
  else
    hipLaunchKernelGGL((three_nn_grid_kernel<false, WEIGHTS, IdxT>), grid, dim3(NN_THREADS), 0, st,
                       k, (int)N1, (int)N2, inv_h, d2_done, g, cw, eps, idx, out, fail_list,
                       fail_count, (const float*)cell_dev);
  S4G_LAUNCH_CHECK();
  if (fmad)
    hipLaunchKernelGGL((three_nn_fallback_kernel<true, WEIGHTS, IdxT>), dim3(512), dim3(NN_THREADS),
                       0, st, q, k, (int)N1, (int)N2, eps, idx, out, fail_list, fail_count);
  else
    hipLaunchKernelGGL((three_nn_fallback_kernel<false, WEIGHTS, IdxT>), dim3(512), dim3(NN_THREADS),
                       0, st, q, k, (int)N1, (int)N2, eps, idx, out, fail_list, fail_count);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

}  // namespace s4g

extern "C" int s4g_three_nn_weights_grid_i32(const float* q_b3n1, const float* k_b3n2,
                                             int64_t B, int64_t N1, int64_t N2, float eps,
                                             float cell, int32_t* idx_bn3, float* w_bn3,
                                             void* ws, size_t ws_bytes, int flags,
                                             s4g_stream_t stream) {
  // cell < 0: the edge is chosen on the device from the keys' measured spacing (nn_auto_cell_kernel), as in
  // s4g_three_nn_grid_f32; cell == 0 / NaN is refused
  return s4g::launch_three_nn_grid<true, int32_t>(q_b3n1, k_b3n2, B, N1, N2, eps, cell, idx_bn3,
                                                  w_bn3, ws, ws_bytes, flags, (hipStream_t)stream);
}

// Operator-API form of the grid search: int64 indices + squared distances, i.e. the
// outputs of s4g_three_nn_f32, for callers that can name a cell edge (cell < 0: chosen on
// the device from the keys' spacing).  Identical results for ANY cell: queries whose
// third neighbour is not proven inside the 27 cells are answered by the index-order
// scan kernel.
extern "C" int s4g_three_nn_grid_f32(const float* q_b3n1, const float* k_b3n2, int64_t B,
                                     int64_t N1, int64_t N2, float cell, int64_t* idx_bn3,
                                     float* d2_bn3, void* ws, size_t ws_bytes, int flags,
                                     s4g_stream_t stream) {
  return s4g::launch_three_nn_grid<false, int64_t>(q_b3n1, k_b3n2, B, N1, N2, 0.f, cell, idx_bn3,
                                                   d2_bn3, ws, ws_bytes, flags, (hipStream_t)stream);
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
