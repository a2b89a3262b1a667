// Deterministic backward of group_points / three_interpolate for gfx950.
//
// The reference scatters with atomicAdd (grouping_kernel.cu:94, interpolate_kernel.cu:283): the order in
// which a point's contributions meet is whatever the hardware serves, so its gradients differ run to
// run in the last bits.  Here every contribution is keyed by its target (scene, point), the
// (key, position) pairs are ordered by a STABLE radix sort -- ascending target, ascending position
// inside a target -- and each target is summed by one thread in that order:
//     gin[b][c][j] = ((0 + g(t_0)) + g(t_1)) + ...      t_0 < t_1 < ... the positions with index j
// which is the sum a sequential loop over the positions forms (oracle/s4g_oracle.c: the linearId
// order) -- run-to-run bit-identical and equal to the oracle bit for bit.  Targets nobody points at
// get an exact 0.  The atomic kernels stay (csrc/group.hip, csrc/interpolate.hip): fewer passes,
// undefined order.
//
// Work: one radix sort of B T 8-byte pairs (csrc/radix_sort.hip: the repo's own stable LSD sort since round 6 --
// rocPRIM's device radix sort before), two binary searches per target, then for every (scene, channel) row one
// coalesced sweep over the targets whose reads of the row's T gradients land in L2 (a row is
// T x 4 bytes: 1.3 MB at the first SA level).
#include <hip/hip_runtime.h>
#include "radix_sort.h"
#include "s4g_common.h"

namespace s4g {

constexpr int SC_THREADS = 256;

// key = b * N + index (an index outside [0, N) gets the sentinel B * N: sorted behind every target,
// summed by nobody -- the reference leaves such an index to an out-of-bounds atomicAdd), value = b * T + t
__global__ __launch_bounds__(SC_THREADS) void scatter_keys_kernel(const int64_t* __restrict__ idx, int64_t BT,
                                                                  int64_t T, int64_t N, uint32_t sentinel,
                                                                  uint32_t* __restrict__ keys,
                                                                  uint32_t* __restrict__ vals) {
  const int64_t p = (int64_t)blockIdx.x * SC_THREADS + threadIdx.x;
  if (p >= BT) return;
  const int64_t b = p / T, j = idx[p];
  keys[p] = (j >= 0 && j < N) ? (uint32_t)(b * N + j) : sentinel;
  vals[p] = (uint32_t)p;
}

// start[g] = first sorted slot whose key is >= g, for g = 0 .. B N (lower bound; start[B N] ends the last target)
__global__ __launch_bounds__(SC_THREADS) void scatter_starts_kernel(const uint32_t* __restrict__ keys, int64_t BT,
                                                                    int64_t BN, uint32_t* __restrict__ start) {
  const int64_t g = (int64_t)blockIdx.x * SC_THREADS + threadIdx.x;
  if (g > BN) return;
  int64_t lo = 0, hi = BT;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < g) lo = mid + 1; else hi = mid;
  }
  start[g] = (uint32_t)lo;
}

// grid (ceil(N / SC_THREADS), C, B): thread = one target of one channel row.  WEIGHTED: three_interpolate --
// position p = (b N1 + n) * 3 + k reads gout[b][c][n] and is scaled by w[p] (product rounded, then added:
// interpolate_kernel.cu:283's grad * w); else group_points -- position p = b T + t reads gout[b][c][t].
template <bool WEIGHTED>
__global__ __launch_bounds__(SC_THREADS) void scatter_sum_kernel(const float* __restrict__ gout,
                                                                 const float* __restrict__ w,
                                                                 const uint32_t* __restrict__ pos,
                                                                 const uint32_t* __restrict__ start, int C,
                                                                 int64_t N, int64_t T, float* __restrict__ gin) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * SC_THREADS + threadIdx.x;
  if (j >= N) return;
  const int64_t g = (int64_t)b * N + j;
  const uint32_t s0 = start[g], s1 = start[g + 1];
  const int64_t src_row = WEIGHTED ? T / 3 : T;                 // elements of one (scene, channel) row of gout
  const float* __restrict__ row = gout + ((size_t)b * C + c) * src_row;
  const int64_t base = (int64_t)b * T;
  float acc = 0.f;
  for (uint32_t i = s0; i < s1; ++i) {
    const int64_t p = pos[i];
    if constexpr (WEIGHTED)
      acc = __fadd_rn(acc, __fmul_rn(row[(p - base) / 3], w[p]));
    else
      acc = __fadd_rn(acc, row[p - base]);
  }
  gin[((size_t)b * C + c) * N + j] = acc;
}

// ---- channels-last form for feature tensors (C >= 32): the sums above gather 4 bytes per (channel, position) from a
// channel-first row -- a 32-byte sector per useful float.  With the gradients copied once to (B, S, C) a target's wave
// reads a position's 64 channels as ONE 256-byte row, lanes = channels; same positions, same order, same arithmetic per
// (channel, target): bit-identical to the thread-per-target kernel.
constexpr int SC_TILE = 64;

// (B, C, S) -> (B, S, C) through a 64 x 64 LDS tile; grid (ceil(S / 64), ceil(C / 64), B), 256 threads
__global__ __launch_bounds__(SC_THREADS) void scatter_to_cl_kernel(const float* __restrict__ in, int C, int64_t S,
                                                                   float* __restrict__ out) {
  __shared__ float tile[SC_TILE][SC_TILE + 1];
  const int b = blockIdx.z, c0 = blockIdx.y * SC_TILE;
  const int64_t s0 = (int64_t)blockIdx.x * SC_TILE;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < SC_TILE; r += 4) {      // rows = channels, lanes = positions
    const int c = c0 + r;
    const int64_t sp = s0 + tx;
    tile[r][tx] = (c < C && sp < S) ? in[((size_t)b * C + c) * S + sp] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < SC_TILE; r += 4) {      // rows = positions, lanes = channels
    const int64_t sp = s0 + r;
    const int c = c0 + tx;
    if (sp < S && c < C) out[((size_t)b * S + sp) * C + c] = tile[tx][r];
  }
}

// grid (ceil(N / 64), ceil(C / 64), B), 256 threads: wave w sums targets j0 + 16 w .. + 15 one after the other, lane =
// channel c0 + lane; the 64 x 64 results leave through LDS as 256-byte rows of the channel-first output.
template <bool WEIGHTED>
__global__ __launch_bounds__(SC_THREADS) void scatter_sum_cl_kernel(const float* __restrict__ gcl,
                                                                    const float* __restrict__ w,
                                                                    const uint32_t* __restrict__ pos,
                                                                    const uint32_t* __restrict__ start, int C,
                                                                    int64_t N, int64_t T, float* __restrict__ gin) {
  __shared__ float tile[SC_TILE][SC_TILE + 1];   // [channel][target]
  const int b = blockIdx.z, c0 = blockIdx.y * SC_TILE;
  const int64_t j0 = (int64_t)blockIdx.x * SC_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = c0 + lane;
  const int64_t S = WEIGHTED ? T / 3 : T, base = (int64_t)b * T;
  const float* __restrict__ rows = gcl + (size_t)b * S * C + (c < C ? c : 0);
  for (int i = 0; i < 16; ++i) {
    const int64_t j = j0 + wave * 16 + i;
    float acc = 0.f;
    if (j < N) {
      const int64_t g = (int64_t)b * N + j;
      const uint32_t s0 = start[g], s1 = start[g + 1];
      for (uint32_t k = s0; k < s1; ++k) {
        const int64_t p = (int64_t)__builtin_amdgcn_readfirstlane((int)pos[k]);     // (wave-uniform: one target per wave)
        const int64_t src = WEIGHTED ? (p - base) / 3 : p - base;
        const float v = rows[(size_t)src * C];
        if constexpr (WEIGHTED)
          acc = __fadd_rn(acc, __fmul_rn(v, w[p]));
        else
          acc = __fadd_rn(acc, v);
      }
    }
    tile[lane][wave * 16 + i] = acc;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < SC_TILE; r += 4) {      // rows = channels, lanes = targets
    const int cc = c0 + r;
    const int64_t j = j0 + tx;
    if (cc < C && j < N) gin[((size_t)b * C + cc) * N + j] = tile[r][tx];
  }
}

struct ScatterWs {
  uint32_t *keys_in, *keys_out, *vals_in, *vals_out, *start;
  void* tmp;
  float* gcl;          // behind `total`: the channels-last copy of the gradients (total_cl includes it)
  size_t tmp_bytes, total, total_cl;
};

constexpr int SC_CL_MIN_C = 32;   // from here the channels-last form pays (a 256-byte row is at least half used)

static int key_bits(int64_t BN) {   // keys run 0 .. B N (the sentinel)
  int bits = 1;
  while (((int64_t)1 << bits) <= BN) ++bits;
  return bits;
}

static ScatterWs scatter_ws(void* base, int64_t B, int64_t N, int64_t T, int64_t C = 0, int64_t S = 0) {
  ScatterWs w;
  const size_t BT = (size_t)(B * T), BN = (size_t)(B * N);
  const size_t sort_bytes = radix_sort_ws_bytes(BT);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  char* p = (char*)base;
  size_t off = 0;
  w.keys_in = (uint32_t*)(p + off); off += up(BT * 4);
  w.keys_out = (uint32_t*)(p + off); off += up(BT * 4);
  w.vals_in = (uint32_t*)(p + off); off += up(BT * 4);
  w.vals_out = (uint32_t*)(p + off); off += up(BT * 4);
  w.start = (uint32_t*)(p + off); off += up((BN + 1) * 4);
  w.tmp = p + off; off += up(sort_bytes);
  w.tmp_bytes = sort_bytes;
  w.total = off;
  w.gcl = (float*)(p + off);
  w.total_cl = off + up((size_t)B * (size_t)S * (size_t)C * 4);
  return w;
}

template <bool WEIGHTED>
static int scatter_det(const float* gout, const int64_t* idx, const float* wgt, int64_t B, int64_t C, int64_t N,
                       int64_t T, float* gin, void* ws, size_t ws_bytes, hipStream_t st) {
  if (B < 0 || C < 0 || N <= 0 || T < 0 || B > 65535 || C > 65535) return S4G_EINVAL;
  if (B * T >= ((int64_t)1 << 31) || B * N >= ((int64_t)1 << 31) - 1) return S4G_EUNSUPPORTED;   // 32-bit keys / positions
  if (B == 0 || C == 0) return S4G_OK;
  if (!gin) return S4G_EINVAL;
  if (T == 0) {
    if (hipMemsetAsync(gin, 0, sizeof(float) * (size_t)(B * C * N), st) != hipSuccess) return S4G_EINVAL;
    return S4G_OK;
  }
  if (!gout || !idx || (WEIGHTED && !wgt) || !ws || ((uintptr_t)ws & 255)) return S4G_EINVAL;
  const int64_t S = WEIGHTED ? T / 3 : T;
  ScatterWs w = scatter_ws(ws, B, N, T, C, S);
  if (ws_bytes < w.total) return S4G_EWORKSPACE;
  const bool cl = C >= SC_CL_MIN_C && ws_bytes >= w.total_cl;   // (a workspace without room for the copy: the gather form)
  const int64_t BT = B * T, BN = B * N;
  hipLaunchKernelGGL(scatter_keys_kernel, dim3((unsigned)((BT + SC_THREADS - 1) / SC_THREADS)), dim3(SC_THREADS), 0, st,
                     idx, BT, T, N, (uint32_t)BN, w.keys_in, w.vals_in);
  S4G_LAUNCH_CHECK();
  if (radix_sort_pairs(w.tmp, w.tmp_bytes, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)BT,
                       (unsigned)key_bits(BN), st) != (int)hipSuccess)
    return S4G_EINVAL;
  hipLaunchKernelGGL(scatter_starts_kernel, dim3((unsigned)((BN + 1 + SC_THREADS - 1) / SC_THREADS)), dim3(SC_THREADS), 0,
                     st, w.keys_out, BT, BN, w.start);
  S4G_LAUNCH_CHECK();
  if (cl) {
    const dim3 tg((unsigned)((S + SC_TILE - 1) / SC_TILE), (unsigned)((C + SC_TILE - 1) / SC_TILE), (unsigned)B);
    hipLaunchKernelGGL(scatter_to_cl_kernel, tg, dim3(SC_THREADS), 0, st, gout, (int)C, S, w.gcl);
    S4G_LAUNCH_CHECK();
    const dim3 sg((unsigned)((N + SC_TILE - 1) / SC_TILE), (unsigned)((C + SC_TILE - 1) / SC_TILE), (unsigned)B);
    hipLaunchKernelGGL((scatter_sum_cl_kernel<WEIGHTED>), sg, dim3(SC_THREADS), 0, st, w.gcl, wgt, w.vals_out, w.start, (int)C,
                       N, T, gin);
    S4G_LAUNCH_CHECK();
    return S4G_OK;
  }
  const dim3 grid((unsigned)((N + SC_THREADS - 1) / SC_THREADS), (unsigned)C, (unsigned)B);
  hipLaunchKernelGGL((scatter_sum_kernel<WEIGHTED>), grid, dim3(SC_THREADS), 0, st, gout, wgt, w.vals_out, w.start, (int)C,
                     N, T, gin);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

}  // namespace s4g

extern "C" size_t s4g_scatter_det_workspace_bytes(int64_t B, int64_t N, int64_t T) {
  if (B <= 0 || N <= 0 || T <= 0 || B * T >= ((int64_t)1 << 31) || B * N >= ((int64_t)1 << 31) - 1) return 0;
  return s4g::scatter_ws(nullptr, B, N, T).total;
}

// ... with room for the channels-last copy of the gradients (C >= 32: the sums then read 256-byte rows instead of 4-byte
// gathers, ~3 x faster; same results bit for bit).  weighted = 0: group_points (T = M K), 1: three_interpolate (T = 3 N1).
extern "C" size_t s4g_scatter_det_workspace_bytes_c(int64_t B, int64_t C, int64_t N, int64_t T, int weighted) {
  if (B <= 0 || N <= 0 || T <= 0 || C < 0 || B * T >= ((int64_t)1 << 31) || B * N >= ((int64_t)1 << 31) - 1) return 0;
  const s4g::ScatterWs w = s4g::scatter_ws(nullptr, B, N, T, C, weighted ? T / 3 : T);
  return C >= s4g::SC_CL_MIN_C ? w.total_cl : w.total;
}

extern "C" int s4g_group_points_backward_det_f32(const float* gout_bcmk, const int64_t* idx_bmk, int64_t B, int64_t C,
                                                 int64_t N, int64_t M, int64_t K, float* gin_bcn, void* ws,
                                                 size_t ws_bytes, s4g_stream_t stream) {
  if (M < 0 || K < 0) return S4G_EINVAL;
  return s4g::scatter_det<false>(gout_bcmk, idx_bmk, nullptr, B, C, N, M * K, gin_bcn, ws, ws_bytes,
                                 (hipStream_t)stream);
}

extern "C" int s4g_three_interpolate_backward_det_f32(const float* gout_bcn1, const int64_t* idx_bn3,
                                                      const float* w_bn3, int64_t B, int64_t C, int64_t N2, int64_t N1,
                                                      float* gin_bcn2, void* ws, size_t ws_bytes, s4g_stream_t stream) {
  if (N1 < 0) return S4G_EINVAL;
  return s4g::scatter_det<true>(gout_bcn1, idx_bn3, w_bn3, B, C, N2, N1 * 3, gin_bcn2, ws, ws_bytes,
                                (hipStream_t)stream);
}
