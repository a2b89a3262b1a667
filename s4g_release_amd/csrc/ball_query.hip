// Ball query for gfx950.
//
// Replaces BallQueryKernel (reference pointnet2_utils/csrc/
// ball_query_kernel.cu:33-76, host :89-133).  Semantics:
// r2 = radius*radius in fp32; per centroid scan points in INDEX ORDER, keep the
// first K with d < r2 (strict); slots cnt..K-1 hold the first hit; rows
// without a hit stay zero; count = min(hits, K).
//
// Two exact implementations, chosen per call:
//
//  SCAN  index-order scan, one WAVE per centroid with the 64 lanes on 64
//        consecutive points.  A ballot gives the hit mask, mbcnt the rank of
//        each hit, so hits land in their slot already in index order and the
//        wave leaves as soon as K are found.  O(M*N) pair tests: right for the
//        small, dense levels (SA2/SA3: K hits appear after ~N*K/hits points).
//
//  GRID  for big sparse levels (SA1: 25 600 points, 76 % of the balls never
//        reach K = 64, so a scan reads everything).  Per call:
//        1. build: a 32x32x32 *toroidal* cell grid with cell edge h = r*(1+2^-8)
//           (cell coordinates mod 32; clouds wider than 32 cells alias, which
//           only adds candidates).  One launch, 8 workgroups per scene, each
//           owning the 4096 slots of one y-stripe class (y mod 8): LDS
//           histogram -> scan -> scatter of (x,y,z,index) records.  No global
//           atomics, no cross-workgroup exchange, no bounding-box pass (cell
//           coordinates are relative to the scene's first point).
//        2. query: a wave walks 8 centroids; for each it visits the 9 rows of 3
//           x-adjacent cells around it (each row is ONE contiguous record
//           range, all rows in flight at once), tests the ~100-350 candidates,
//           and marks hits in an LDS *bitmap over the point index*.  Reading the bitmap back in order yields exactly the
//           reference's "first K in index order" -- candidates may arrive in any
//           order, be duplicated by aliasing, or exceed K, without affecting
//           the result.  Output rows are staged in LDS and stored coalesced.
//        Exactness: a point with fp32 d < r2 has |dx| <= r(1+1e-6) in every
//        axis, so its cell coordinate differs by at most 1 from the
//        centroid's as long as the fp32 error of (p-o)/h stays below the 2^-8
//        margin; scenes with |cell coordinate| >= 4096 raise a flag and their
//        centroids take the SCAN path inside the same kernel.
//        The query kernel can also emit the grouped xyz (group_points of the
//        same indices, grouping_kernel.cu:32-54) so the operator pair
//        QueryGrouper uses (modules.py:39-42) costs one pass over the output.
#include <stdlib.h>

#include "grid.h"

namespace s4g {

constexpr int BQ_WAVES_PER_BLOCK = 4;
constexpr int BQ_UNROLL = 4;

// ---------------------------------------------------------------- SCAN path
template <bool FMAD, typename IdxT>
__device__ __forceinline__ void bq_scan_centroid(
    const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, int N, float cx, float cy, float cz, float r2,
    int K, int lane, IdxT* __restrict__ row, IdxT* __restrict__ cnt_slot) {
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
  if (lane == 0) *cnt_slot = (IdxT)cnt;
}

template <bool FMAD, typename IdxT>
__global__ __launch_bounds__(64 * BQ_WAVES_PER_BLOCK) void ball_query_scan_kernel(
    const float* __restrict__ xyz, const float* __restrict__ ctr, int N, int M,
    float r2, int K, IdxT* __restrict__ idx, IdxT* __restrict__ cnt_out) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * BQ_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* __restrict__ px = xyz + (size_t)b * 3 * N;
  const float* __restrict__ c = ctr + (size_t)b * 3 * M;
  bq_scan_centroid<FMAD, IdxT>(px, px + N, px + 2 * N, N, c[m], c[M + m], c[2 * M + m], r2, K,
                               lane, idx + ((size_t)b * M + m) * K, cnt_out + (size_t)b * M + m);
}

// ---------------------------------------------------------------- GRID path
// (grid layout, slot mapping and workspace carving: grid.h)
// PPT > 0: the cloud fits PPT points per thread -- coordinates and slots stay in registers
// between the counting and the scatter pass (one round trip to memory instead of
// 2 * ceil(points per thread / GR_BUILD_U)); PPT == 0: any size, two streaming passes.
template <int PPT>
__global__ __launch_bounds__(GR_BUILD_THREADS) void bq_grid_build_kernel(
    const float* __restrict__ xyz, int N, float inv_h, GridWs ws, int write_aos,
    const float* __restrict__ ctr, int M, CellWs cw, const float* __restrict__ inv_h_dev) {
  if (inv_h_dev) inv_h = *inv_h_dev;  // cell edge chosen on the device (3-NN operator API)
  __shared__ uint32_t hist[GR_RANGE_SLOTS];
  __shared__ uint32_t wsum[GR_BUILD_THREADS / 64];
  __shared__ uint32_t wsum2[GR_BUILD_THREADS / 64];
  const int b = blockIdx.y;
  const int g = blockIdx.x & (GR_RANGES - 1);
  const bool queries = blockIdx.x >= GR_RANGES;  // workgroups 8..15 bin the queries
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const float* __restrict__ p0 = xyz + (size_t)b * 3 * N;
  const float ox = p0[0], oy = p0[N], oz = p0[2 * (size_t)N];  // origin: the scene's first point
  const int n = queries ? M : N;
  const float* __restrict__ px = queries ? ctr + (size_t)b * 3 * M : p0;
  const float* __restrict__ py = px + n;
  const float* __restrict__ pz = py + n;

  for (int s = t; s < GR_RANGE_SLOTS; s += GR_BUILD_THREADS) hist[s] = 0;
  __syncthreads();

  // The point loops issue GR_BUILD_U independent plane loads per lane before
  // touching them: the slab workgroups are latency-, not bandwidth-bound.
  int bad = 0;
  constexpr int RP = PPT > 0 ? PPT : 1;
  float rx[RP], ry[RP], rz[RP];
  if constexpr (PPT > 0) {
    // buffer loads: one lane offset for all of them, the slot offset is an immediate / scalar
    // and the range check returns 0 past the end (no per-load address registers)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)px, 0, n * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)py, 0, n * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void*)pz, 0, n * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int soff = u * GR_BUILD_THREADS * 4;
      rx[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, t * 4, soff, 0));
      ry[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, t * 4, soff, 0));
      rz[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(zr, t * 4, soff, 0));
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      if (t + u * GR_BUILD_THREADS < n) {
        if (!(grid_coord_ok(rx[u], ox, inv_h) && grid_coord_ok(ry[u], oy, inv_h) &&
              grid_coord_ok(rz[u], oz, inv_h)))
          bad = 1;
        const int slot = grid_slot(grid_coord(rx[u], ox, inv_h), grid_coord(ry[u], oy, inv_h),
                                   grid_coord(rz[u], oz, inv_h));
        if ((slot >> 12) == g) atomicAdd(&hist[slot & (GR_RANGE_SLOTS - 1)], 1u);
      }
    }
  } else {
  for (int j0 = t; j0 < n; j0 += GR_BUILD_THREADS * GR_BUILD_U) {
    float x[GR_BUILD_U], y[GR_BUILD_U], z[GR_BUILD_U];
#pragma unroll
    for (int u = 0; u < GR_BUILD_U; ++u) {
      const int j = j0 + u * GR_BUILD_THREADS;
      const int jj = j < n ? j : 0;
      x[u] = px[jj];
      y[u] = py[jj];
      z[u] = pz[jj];
    }
#pragma unroll
    for (int u = 0; u < GR_BUILD_U; ++u) {
      if (j0 + u * GR_BUILD_THREADS < n) {
        if (!(grid_coord_ok(x[u], ox, inv_h) && grid_coord_ok(y[u], oy, inv_h) &&
              grid_coord_ok(z[u], oz, inv_h)))
          bad = 1;
        const int slot = grid_slot(grid_coord(x[u], ox, inv_h), grid_coord(y[u], oy, inv_h),
                                   grid_coord(z[u], oz, inv_h));
        if ((slot >> 12) == g) atomicAdd(&hist[slot & (GR_RANGE_SLOTS - 1)], 1u);
      }
    }
  }
  }
  bad = __syncthreads_or(bad);
  if (!queries && g == 0 && t == 0) ws.flags[b] = bad ? 1 : 0;

  // exclusive scan of the 4096 counts: 4 consecutive entries per thread
  const uint32_t v0 = hist[4 * t], v1 = hist[4 * t + 1], v2 = hist[4 * t + 2], v3 = hist[4 * t + 3];
  const uint32_t s = v0 + v1 + v2 + v3;
  const uint32_t nzc = s != 0u ? 1u : 0u;  // non-empty x-quads
  // counts and non-empty-cell counts scanned together (both < 2^16 per workgroup here?
  // no: n may reach 2^16 points, so keep two scans)
  uint32_t incl = s, incl2 = nzc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off);
    const uint32_t o2 = __shfl_up(incl2, off);
    if (lane >= off) {
      incl += o;
      incl2 += o2;
    }
  }
  if (lane == 63) {
    wsum[wave] = incl;
    wsum2[wave] = incl2;
  }
  __syncthreads();
  uint32_t wbase = 0, wbase2 = 0;
  for (int w = 0; w < wave; ++w) {
    wbase += wsum[w];
    wbase2 += wsum2[w];
  }
  const uint32_t base = (uint32_t)g * (uint32_t)n + wbase + (incl - s);
  hist[4 * t] = base;  // each thread rewrites only the four entries it read
  hist[4 * t + 1] = base + v0;
  hist[4 * t + 2] = base + v0 + v1;
  hist[4 * t + 3] = base + v0 + v1 + v2;
  if (!queries) {
    int* __restrict__ st = ws.starts + ((size_t)b * GR_RANGES + g) * GR_START_STRIDE;
    *reinterpret_cast<int4*>(st + 4 * t) =
        make_int4((int)base, (int)(base + v0), (int)(base + v0 + v1), (int)(base + v0 + v1 + v2));
    if (t == GR_BUILD_THREADS - 1) st[GR_RANGE_SLOTS] = (int)(base + s);
  } else {
    // compact list of the non-empty x-quads (4 x-adjacent cells = this thread's 4 slots,
    // whose records are contiguous): (first slot, first record, count)
    int4* __restrict__ ce = cw.cells + ((size_t)b * GR_RANGES + g) * GR_RANGE_SLOTS;
    const uint32_t pos = wbase2 + (incl2 - nzc);
    if (s) ce[pos] = make_int4((g << 12) | (4 * t), (int)base, (int)s, 0);
    if (t == GR_BUILD_THREADS - 1) {
      cw.ncell[b * GR_RANGES + g] = (int)(pos + (s ? 1u : 0u));
      cw.ncell[(gridDim.y + b) * GR_RANGES + g] = (int)(wbase + incl);  // records of this stripe
    }
  }
  __syncthreads();

  float4* __restrict__ rec = queries ? cw.sorted + (size_t)b * GR_RANGES * M
                                     : ws.sorted + (size_t)b * GR_RANGES * N;
  if constexpr (PPT > 0) {
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int j = t + u * GR_BUILD_THREADS;
      if (!queries && g == 0 && write_aos && j < n)
        ws.xyz4[(size_t)b * N + j] = make_float4(rx[u], ry[u], rz[u], 0.f);
      // (the slot is recomputed: keeping it would cost a register per point)
      const int slot = grid_slot(grid_coord(rx[u], ox, inv_h), grid_coord(ry[u], oy, inv_h),
                                 grid_coord(rz[u], oz, inv_h));
      if (j < n && (slot >> 12) == g) {
        const uint32_t pos = atomicAdd(&hist[slot & (GR_RANGE_SLOTS - 1)], 1u);
        rec[pos] = make_float4(rx[u], ry[u], rz[u], __int_as_float(j));
      }
    }
    return;
  }
  for (int j0 = t; j0 < n; j0 += GR_BUILD_THREADS * GR_BUILD_U) {
    float x[GR_BUILD_U], y[GR_BUILD_U], z[GR_BUILD_U];
#pragma unroll
    for (int u = 0; u < GR_BUILD_U; ++u) {
      const int j = j0 + u * GR_BUILD_THREADS;
      const int jj = j < n ? j : 0;
      x[u] = px[jj];
      y[u] = py[jj];
      z[u] = pz[jj];
    }
#pragma unroll
    for (int u = 0; u < GR_BUILD_U; ++u) {
      const int j = j0 + u * GR_BUILD_THREADS;
      if (j < n) {
        if (!queries && g == 0 && write_aos)
          ws.xyz4[(size_t)b * N + j] = make_float4(x[u], y[u], z[u], 0.f);
        const int slot = grid_slot(grid_coord(x[u], ox, inv_h), grid_coord(y[u], oy, inv_h),
                                   grid_coord(z[u], oz, inv_h));
        if ((slot >> 12) == g) {
          const uint32_t pos = atomicAdd(&hist[slot & (GR_RANGE_SLOTS - 1)], 1u);
          rec[pos] = make_float4(x[u], y[u], z[u], __int_as_float(j));
        }
      }
    }
  }
}

// Reads a wave's hit bitmap back in index order: lane l owns words [l*W, (l+1)*W),
// the first K set bits land in row[0..), touched words are cleared.  Returns the
// number of set bits (may exceed K).
template <int WPL>
__device__ __forceinline__ int bq_bitmap_readback(uint32_t* __restrict__ bm, int lane,
                                                  int words_per_lane, int K,
                                                  int* __restrict__ row) {
  uint32_t* mine = bm + lane * words_per_lane;
  int local = 0;
  uint32_t nz = 0;  // which of this lane's words are non-zero
  if constexpr (WPL > 0) {
    uint32_t wv[WPL];
#pragma unroll
    for (int w = 0; w < WPL; ++w) wv[w] = mine[w];
#pragma unroll
    for (int w = 0; w < WPL; ++w) {
      local += __popc(wv[w]);
      nz |= (wv[w] != 0u ? 1u : 0u) << w;
    }
  } else {
    for (int w = 0; w < words_per_lane; ++w) local += __popc(mine[w]);
  }
  const int incl = (int)wave_inclusive_scan_u32((uint32_t)local);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  int pos = incl - local;
  if constexpr (WPL > 0) {
    while (nz) {  // only the touched words; clear them for the next centroid
      const int w = __ffs(nz) - 1;
      nz &= nz - 1;
      uint32_t bits = mine[w];
      mine[w] = 0u;
      while (bits && pos < K) {
        const int bit = __ffs(bits) - 1;
        row[pos++] = (lane * WPL + w) * 32 + bit;
        bits &= bits - 1;
      }
    }
  } else if (local > 0) {
    for (int w = 0; w < words_per_lane; ++w) {
      uint32_t bits = mine[w];
      if (bits) mine[w] = 0u;
      while (bits && pos < K) {
        const int bit = __ffs(bits) - 1;
        row[pos++] = (lane * words_per_lane + w) * 32 + bit;
        bits &= bits - 1;
      }
    }
  }
  return total;
}

// One wave per centroid.  Dynamic LDS per wave: bitmap of N bits + K-entry row.
// Each wave owns BQ_CPW (template) consecutive centroids and one LDS bitmap that it keeps
// clean (touched words are cleared while they are read back).  Lane l < BQ_CPW
// fetches centroid l up front, the 9 row ranges of centroid c+1 are fetched
// while centroid c is processed, and a lane keeps BQ_REC record loads in
// flight: a wave is latency-bound, so dependent round trips are what is
// minimised.  WPL = bitmap words per lane (compile time; 0 = runtime value).
constexpr int BQ_REC = 4;

template <bool FMAD, typename IdxT, bool GROUP, int WPL, int BQ_CPW>
__global__ __launch_bounds__(64 * BQ_WAVES_PER_BLOCK) void bq_grid_query_kernel(
    const float* __restrict__ xyz, const float* __restrict__ ctr, int N, int M,
    float r2, float inv_h, int K, GridWs ws, IdxT* __restrict__ idx,
    IdxT* __restrict__ cnt_out, float* __restrict__ grouped, int words_per_lane_rt) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int words_per_lane = WPL > 0 ? WPL : words_per_lane_rt;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m0 = (blockIdx.x * BQ_WAVES_PER_BLOCK + wave) * BQ_CPW;
  if (m0 >= M) return;
  const int nc = min(BQ_CPW, M - m0);
  const int bm_words = 64 * words_per_lane;          // multiple of 4
  const int wave_words = bm_words + ((K + 3) & ~3);
  uint32_t* __restrict__ bm = lds + wave * wave_words;
  int* __restrict__ row = (int*)(bm + bm_words);
  for (int w = lane * 4; w < bm_words; w += 256)
    *reinterpret_cast<uint4*>(bm + w) = make_uint4(0u, 0u, 0u, 0u);

  const float* __restrict__ px = xyz + (size_t)b * 3 * N;
  const float* __restrict__ py = px + N;
  const float* __restrict__ pz = py + N;
  const float* __restrict__ c = ctr + (size_t)b * 3 * M;
  const float ox = px[0], oy = py[0], oz = pz[0];
  const bool scene_ok = ws.flags[b] == 0;
  const float4* __restrict__ rec = ws.sorted + (size_t)b * GR_RANGES * N;
  const __amdgpu_buffer_rsrc_t rrec =
      __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, (int)((size_t)GR_RANGES * N * sizeof(float4)), 0x00020000);
  const int* __restrict__ starts_b = ws.starts + (size_t)b * GR_RANGES * GR_START_STRIDE;

  // lane l holds centroid m0 + l
  const int ml = m0 + (lane < nc ? lane : 0);
  const float lcx = c[ml], lcy = c[M + ml], lcz = c[2 * M + ml];
  const int lix = grid_coord(lcx, ox, inv_h), liy = grid_coord(lcy, oy, inv_h),
            liz = grid_coord(lcz, oz, inv_h);
  const bool lexact = scene_ok && grid_coord_ok(lcx, ox, inv_h) &&
                      grid_coord_ok(lcy, oy, inv_h) && grid_coord_ok(lcz, oz, inv_h);

  // record ranges of the 9 (dy, dz) rows of centroid `ci`, fetched by lanes 0..8
  auto fetch_rows = [&](int ci, int& beg0, int& end0, int& beg1, int& end1) {
    const int icx = __shfl(lix, ci), icy = __shfl(liy, ci), icz = __shfl(liz, ci);
    beg0 = end0 = beg1 = end1 = 0;
    if (lane < 9) {
      const int dz = lane / 3 - 1, dy = lane % 3 - 1;
      const int zz = (icz + dz) & 31, yy = (icy + dy) & 31;
      const int* __restrict__ st = starts_b + grid_range(yy, zz) * GR_START_STRIDE +
                                   grid_local_row(yy, zz);
      const int x0 = (icx - 1) & 31;
      if (x0 <= GR_DIM - 3) {
        beg0 = st[x0];
        end0 = st[x0 + 3];
      } else {  // the 3 cells wrap around the row
        beg0 = st[x0];
        end0 = st[GR_DIM];
        beg1 = st[0];
        end1 = st[(x0 + 3) & 31];
      }
    }
  };

  const int grp = lane / 7, sub = lane - grp * 7;  // lanes 7r..7r+6 walk row r; lane 63 idles
  const int src = grp < 9 ? grp : 0;
  int nb0, ne0, nb1, ne1;
  fetch_rows(0, nb0, ne0, nb1, ne1);

  for (int ci = 0; ci < nc; ++ci) {
    const int m = m0 + ci;
    const float cx = __shfl(lcx, ci), cy = __shfl(lcy, ci), cz = __shfl(lcz, ci);
    const bool exact = __shfl((int)lexact, ci) != 0;
    const int beg0 = nb0, end0 = ne0, beg1 = nb1, end1 = ne1;
    if (ci + 1 < nc) fetch_rows(ci + 1, nb0, ne0, nb1, ne1);  // in flight during this centroid
    IdxT* __restrict__ out_row = idx + ((size_t)b * M + m) * K;
    IdxT* __restrict__ out_cnt = cnt_out + (size_t)b * M + m;
    bool in_lds = false;
    // (round 4) A ball whose 27 cells hold a large share of the cloud -- a radius far above the point spacing --
    // is answered faster by the index-order scan, which stops at the K-th hit: with C candidates and h ~ 0.3 C
    // hits the scan reads ~ K N / h points, fewer than C once C^2 > ~200 N (16 x 25 600 points, 5 120 centres, r = 0.2:
    // 2.85 -> 0.08 ms; r = 0.05: 0.30 -> 0.21; the shipped r = 0.02: 0.090 -> 0.091).
    // Same result by construction ("first K in index order" either way).  The sum is only formed when a row
    // is long, so the usual ball (rows of ~30 records) pays one compare.
    bool dense = false;
    {
      const int len = (end0 - beg0) + (end1 - beg1);      // lanes 0..8: this row's records
      if (__any(len > 128)) {
        int tot = len;
        tot += __shfl_xor(tot, 1);
        tot += __shfl_xor(tot, 2);
        tot += __shfl_xor(tot, 4);
        tot += __shfl_xor(tot, 8);
        tot = __shfl(tot, 0);                              // lanes 0..15 hold the nine rows (the others are 0)
        dense = (float)tot * (float)tot > 200.0f * (float)N && tot > 8 * K;
      }
    }
    if (!exact || dense) {
      // out of the grid's exactness range (or a dense ball): index-order scan, straight to global
      bq_scan_centroid<FMAD, IdxT>(px, py, pz, N, cx, cy, cz, r2, K, lane, out_row, out_cnt);
      if constexpr (GROUP) __threadfence_block();  // the row is re-read below by other lanes
    } else {
      in_lds = true;
      const int npiece = __any(end1 > beg1) ? 2 : 1;  // the wrapped second piece is rare
      for (int piece = 0; piece < npiece; ++piece) {
        int j = __shfl(piece == 0 ? beg0 : beg1, src) + sub;
        const int je = grp < 9 ? __shfl(piece == 0 ? end0 : end1, src) : 0;
        if (grp >= 9) j = 0;
        while (__any(j < je)) {
          // ALL BQ_REC record loads go out before any is tested: buffer loads (descriptor + a 32-bit
          // lane offset: no per-load 64-bit address arithmetic; a lane past its row reads whatever
          // follows -- or zeros past the end of the array -- and ignores it), pinned by an empty asm:
          // left to itself the compiler sinks the first load INTO the guarded hit test and splits off
          // its index word, two dependent round trips per iteration
          float4 p[BQ_REC];
#pragma unroll
          for (int u = 0; u < BQ_REC; ++u)
            p[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrec, (j + 7 * u) * 16, 0, 0));
#pragma unroll
          for (int u = 0; u < BQ_REC; ++u) asm volatile("" : "+v"(p[u].x), "+v"(p[u].y), "+v"(p[u].z), "+v"(p[u].w));
#pragma unroll
          for (int u = 0; u < BQ_REC; ++u)
            if (j + 7 * u < je && dist2<FMAD>(cx, cy, cz, p[u].x, p[u].y, p[u].z) < r2) {
              const int pi = __float_as_int(p[u].w);
              atomicOr(&bm[pi >> 5], 1u << (pi & 31));
            }
          j += 7 * BQ_REC;
        }
      }
      const int total = bq_bitmap_readback<WPL>(bm, lane, words_per_lane, K, row);
      const int cnt = total < K ? total : K;
      const int first = cnt > 0 ? row[0] : 0;
      for (int k = lane; k < K; k += 64) {
        const int v = k < cnt ? row[k] : first;
        out_row[k] = (IdxT)v;
        if constexpr (GROUP) row[k] = v;
      }
      if (lane == 0) *out_cnt = (IdxT)cnt;
    }
    if constexpr (GROUP) {
      // group_points(xyz, index) for this centroid: out[b][c][m][k] = xyz[b][c][idx]
      const size_t MK = (size_t)M * K;
      float* __restrict__ gx = grouped + (size_t)b * 3 * MK + (size_t)m * K;
      const float4* __restrict__ p4 = ws.xyz4 + (size_t)b * N;
      for (int k = lane; k < K; k += 64) {
        const int v = in_lds ? row[k] : (int)out_row[k];
        const float4 p = p4[v];   // one 16-byte gather instead of three 4-byte ones
        gx[k] = p.x;
        gx[MK + k] = p.y;
        gx[2 * MK + k] = p.z;
      }
    }
  }
}

#ifdef S4G_VARIANTS
#include "variants/bq_cell.inc"
#endif

int launch_grid_build(const float* xyz, int64_t B, int64_t N, float inv_h, GridWs ws,
                      hipStream_t st, bool write_aos, const float* inv_h_dev) {
#define S4G_GB_LAUNCH(P)                                                                          \
  hipLaunchKernelGGL(bq_grid_build_kernel<P>, dim3(GR_RANGES, (unsigned)B), dim3(GR_BUILD_THREADS), \
                     0, st, xyz, (int)N, inv_h, ws, write_aos ? 1 : 0, (const float*)nullptr, 0, \
                     CellWs{nullptr, nullptr, nullptr}, inv_h_dev)
  // S4G_GRID_BUILD=loop: the streaming variant for every size.  The register-resident workgroups
  // (1024 threads x up to 128 registers) need a whole free CU: alone they are faster (36 -> 20 us at
  // 16 x 25 600 points), underneath a saturating contraction stream they wait longer for one
  // (ball query 0.19 -> 0.42 ms per batch; the step time does not move, geometry has slack there).
  static const bool loop = [] { const char* e = s4g::knob("S4G_GRID_BUILD"); return e && e[0] == 'l'; }();
  if (loop) S4G_GB_LAUNCH(0);
  else if (N <= 8 * GR_BUILD_THREADS) S4G_GB_LAUNCH(8);
  else if (N <= 25 * GR_BUILD_THREADS) S4G_GB_LAUNCH(25);
  else S4G_GB_LAUNCH(0);
#undef S4G_GB_LAUNCH
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

int launch_grid_build_queries(const float* xyz, const float* ctr, int64_t B, int64_t N,
                              int64_t M, float inv_h, GridWs ws, CellWs cw, hipStream_t st,
                              bool write_aos, const float* inv_h_dev) {
  const int64_t big = N > M ? N : M;
#define S4G_GB_LAUNCH(P)                                                                  \
  hipLaunchKernelGGL(bq_grid_build_kernel<P>, dim3(2 * GR_RANGES, (unsigned)B),          \
                     dim3(GR_BUILD_THREADS), 0, st, xyz, (int)N, inv_h, ws, write_aos ? 1 : 0, \
                     ctr, (int)M, cw, inv_h_dev)
  if (big <= 8 * GR_BUILD_THREADS) S4G_GB_LAUNCH(8);
  else if (big <= 25 * GR_BUILD_THREADS) S4G_GB_LAUNCH(25);
  else S4G_GB_LAUNCH(0);
#undef S4G_GB_LAUNCH
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

enum { BQ_AUTO = 0, BQ_SCAN = 1, BQ_GRID = 2, BQ_CELL = 3 };

static int bq_mode() {  // S4G_BQ_MODE=scan|grid|cell|auto (tuning / test knob, read per call)
  const char* e = s4g::knob("S4G_BQ_MODE");
  if (e && e[0] == 's') return BQ_SCAN;
  if (e && e[0] == 'g') return BQ_GRID;
  if (e && e[0] == 'c') return BQ_CELL;
  return BQ_AUTO;
}

static bool bq_use_grid(int64_t N, int64_t K) {
  if (N > GR_MAX_POINTS || K > 1024) return false;
  const int mode = bq_mode();
  if (mode == BQ_SCAN) return false;
  if (mode == BQ_GRID || mode == BQ_CELL) return true;
  return N >= 8192;
}

// centre-quad query instead of the per-centre one (needs the centres binned too).
// OPT-IN (S4G_BQ_MODE=cell): exact, but as built it is slower than the per-centre
// kernel (SA1, 16 scenes: 0.142 ms against 0.116 ms) -- see the CELL path's header.
static bool bq_use_cell(int64_t M) {
#ifdef S4G_VARIANTS
  return bq_mode() == BQ_CELL && M >= 1 && M < (1 << 24);
#else
  (void)M;
  return false;   // the cell-centric kernel is in measurement builds only: S4G_BQ_MODE=cell then means grid
#endif
}

size_t ball_query_workspace_bytes(int64_t B, int64_t N, int64_t M, int64_t K) {
  if (!bq_use_grid(N, K)) return 0;
  return grid_ws_bytes(B, N) + (bq_use_cell(M) ? cell_ws_bytes(B, M) : 0);
}

template <typename IdxT>
static int ball_query_dispatch(const float* xyz, const float* ctr, int64_t B,
                               int64_t N, int64_t M, float radius, int64_t K,
                               IdxT* idx, IdxT* cnt, float* grouped, void* ws,
                               size_t ws_bytes, int flags, hipStream_t st) {
  if (B < 0 || N <= 0 || M < 0 || K <= 0 || N >= (1ll << 31) || B > 65535)
    return S4G_EINVAL;
  if (B == 0 || M == 0) return S4G_OK;
  if (!xyz || !ctr || !idx || !cnt) return S4G_EINVAL;
  // r2 exactly as ball_query_kernel.cu:49 computes it: fp32 product.
  const float r2 = radius * radius;
  const bool fmad = (flags & S4G_FLAG_FMAD) != 0;
  const dim3 block(64 * BQ_WAVES_PER_BLOCK);
  const dim3 grid((unsigned)((M + BQ_WAVES_PER_BLOCK - 1) / BQ_WAVES_PER_BLOCK), (unsigned)B);
  // centroids per wave: 1 measured best on MI355X (0.115 ms vs 0.141 / 0.136 / 0.146 ms
  // for 2 / 4 / 8 at B = 16, SA1 size): more waves in flight beat amortised set-up.
  constexpr int cpw = 1;
  const int cpb = BQ_WAVES_PER_BLOCK * cpw;
  const dim3 qgrid((unsigned)((M + cpb - 1) / cpb), (unsigned)B);
  const bool use_grid = bq_use_grid(N, K) && ws && ws_bytes >= grid_ws_bytes(B, N) &&
                        radius > 0.f && radius < 1e18f;
  if (!use_grid) {
    if (fmad)
      hipLaunchKernelGGL((ball_query_scan_kernel<true, IdxT>), grid, block, 0, st, xyz, ctr,
                         (int)N, (int)M, r2, (int)K, idx, cnt);
    else
      hipLaunchKernelGGL((ball_query_scan_kernel<false, IdxT>), grid, block, 0, st, xyz, ctr,
                         (int)N, (int)M, r2, (int)K, idx, cnt);
    S4G_LAUNCH_CHECK();
    if (grouped) return S4G_EUNSUPPORTED;  // callers group separately on this path
    return S4G_OK;
  }
  const GridWs g = grid_ws_carve(ws, B, N);
  const float h = radius * (1.0f + 1.0f / 256.0f);
  const float inv_h = 1.0f / h;
  const int words = (int)((N + 31) / 32);
  int wpl = (words + 63) / 64;
  if ((wpl & 1) == 0) ++wpl;  // odd stride: conflict-free per-lane word runs
#ifdef S4G_VARIANTS
  if (bq_use_cell(M) && ws_bytes >= grid_ws_bytes(B, N) + cell_ws_bytes(B, M)) {
    const CellWs cw = cell_ws_carve((char*)ws + grid_ws_bytes(B, N), B, M);
    if (int rc = launch_grid_build_queries(xyz, ctr, B, N, M, inv_h, g, cw, st, false)) return rc;
    // workgroups per scene: a workgroup strides over the scene's non-empty centre quads
    int64_t wps = 4096 / B;
    if (const char* e = s4g::knob("S4G_BQ_CELL_WGS")) wps = atoi(e);
    if (wps < 32) wps = 32;
    const int64_t max_quads = (M < GR_RANGES * GR_RANGE_SLOTS / 4) ? M : GR_RANGES * GR_RANGE_SLOTS / 4;
    if (wps > max_quads) wps = max_quads;
    const int per = 4 * BQC_THREADS;  // bitmap words: multiple of 4 per thread
    const int bmw = (words + per - 1) / per * per;
    const size_t clds = sizeof(uint32_t) * (size_t)(bmw + bmw / 2 + 4 * BQC_WIN + 8 +
                                                    BQC_WAVES * (grouped ? 4 : 1) * ((K + 3) & ~3));
    const dim3 cgrid((unsigned)wps, (unsigned)B);
#define S4G_BQC_LAUNCH(F, G)                                                                 \
  hipLaunchKernelGGL((bq_cell_query_kernel<F, IdxT, G>), cgrid, dim3(BQC_THREADS), clds, st, \
                     xyz, (int)N, (int)M, r2, inv_h, (int)K, g, cw, idx, cnt, grouped, bmw,  \
                     (int)wps)
    if (grouped) {
      if (fmad) S4G_BQC_LAUNCH(true, true); else S4G_BQC_LAUNCH(false, true);
    } else {
      if (fmad) S4G_BQC_LAUNCH(true, false); else S4G_BQC_LAUNCH(false, false);
    }
#undef S4G_BQC_LAUNCH
    S4G_LAUNCH_CHECK();
    return S4G_OK;
  }
#endif
  if (int rc = launch_grid_build(xyz, B, N, inv_h, g, st, grouped != nullptr)) return rc;
  const size_t lds = sizeof(uint32_t) * BQ_WAVES_PER_BLOCK * (size_t)(64 * wpl + ((K + 3) & ~3));
#define S4G_BQ_LAUNCH4(F, G, W, C)                                                        \
  hipLaunchKernelGGL((bq_grid_query_kernel<F, IdxT, G, W, C>), qgrid, block, lds, st, xyz, \
                     ctr, (int)N, (int)M, r2, inv_h, (int)K, g, idx, cnt, grouped, wpl)
#define S4G_BQ_LAUNCH3(F, G, W) S4G_BQ_LAUNCH4(F, G, W, cpw)
#define S4G_BQ_LAUNCH(F, G)                      \
  do {                                           \
    if (wpl == 13) S4G_BQ_LAUNCH3(F, G, 13);     \
    else if (wpl == 25) S4G_BQ_LAUNCH3(F, G, 25);\
    else S4G_BQ_LAUNCH3(F, G, 0);                \
  } while (0)
  if (grouped) {
    if (fmad) S4G_BQ_LAUNCH(true, true); else S4G_BQ_LAUNCH(false, true);
  } else {
    if (fmad) S4G_BQ_LAUNCH(true, false); else S4G_BQ_LAUNCH(false, false);
  }
#undef S4G_BQ_LAUNCH
#undef S4G_BQ_LAUNCH3
#undef S4G_BQ_LAUNCH4
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

}  // namespace s4g

extern "C" int s4g_ball_query_f32(const float* xyz_b3n, const float* ctr_b3m,
                                  int64_t B, int64_t N, int64_t M, float radius,
                                  int64_t K, int64_t* idx_bmk, int64_t* cnt_bm,
                                  void* ws, size_t ws_bytes, int flags,
                                  s4g_stream_t stream) {
  return s4g::ball_query_dispatch<int64_t>(xyz_b3n, ctr_b3m, B, N, M, radius, K, idx_bmk, cnt_bm,
                                           nullptr, ws, ws_bytes, flags, (hipStream_t)stream);
}

extern "C" int s4g_ball_query_i32(const float* xyz_b3n, const float* ctr_b3m,
                                  int64_t B, int64_t N, int64_t M, float radius,
                                  int64_t K, int32_t* idx_bmk, int32_t* cnt_bm,
                                  void* ws, size_t ws_bytes, int flags,
                                  s4g_stream_t stream) {
  return s4g::ball_query_dispatch<int32_t>(xyz_b3n, ctr_b3m, B, N, M, radius, K, idx_bmk, cnt_bm,
                                           nullptr, ws, ws_bytes, flags, (hipStream_t)stream);
}

extern "C" int s4g_query_group_f32(const float* xyz_b3n, const float* ctr_b3m,
                                   int64_t B, int64_t N, int64_t M, float radius,
                                   int64_t K, int64_t* idx_bmk, int64_t* cnt_bm,
                                   float* grouped_b3mk, void* ws, size_t ws_bytes,
                                   int flags, s4g_stream_t stream) {
  if (!grouped_b3mk) return S4G_EINVAL;
  int rc = s4g::ball_query_dispatch<int64_t>(xyz_b3n, ctr_b3m, B, N, M, radius, K, idx_bmk,
                                             cnt_bm, grouped_b3mk, ws, ws_bytes, flags,
                                             (hipStream_t)stream);
  if (rc == S4G_EUNSUPPORTED)  // SCAN path: indices are done, group with the plain kernel
    rc = s4g_group_points_f32(xyz_b3n, idx_bmk, B, 3, N, M, K, grouped_b3mk, stream);
  return rc;
}
