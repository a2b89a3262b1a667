// Farthest point sampling for gfx950.
//
// Replaces FarthestPointSampleKernel (reference
// pointnet2_utils/csrc/sampling_kernel.cu:49-119, host :128-172).
//
// Design (MI355X-first, not the reference's shape):
//   * one 1024-thread workgroup (16 waves, one CU) per scene; the scene's xyz
//     AND the running min-distance live in VGPRs for the whole kernel
//     (N <= 25600: 25 points x 4 floats per thread), so a step touches no
//     memory except one 8-byte index store;
//   * a step = 25 distance updates per lane, a DPP wave-argmax, ONE
//     workgroup barrier with a double-buffered 16-slot LDS exchange that also
//     carries the winner's coordinates (no dependent global load per step);
//   * the reference's tie rule is reproduced exactly through a composite key:
//     maximise d, then minimise (bitrev_{log2 bs}(j mod bs) << 23 | j), where
//     bs = clamp(pow2ceil(N),16,512) is the REFERENCE's block size
//     (sampling_kernel.cu:34-42,148-167) -- see SURVEY.md Appendix A.1.
//     Thread t owns points j = t + 1024 p, so all its points share j mod bs
//     and an ascending-p strict '>' scan resolves ties inside a thread.
//   * larger clouds fall back to a streaming kernel (xyz from L2, min-distance
//     in a global workspace).
#include <stdlib.h>

#include "s4g_common.h"

namespace s4g {

constexpr int FPS_THREADS = 1024;  // streaming fallback
constexpr int FPS_MAX_WAVES = 16;
constexpr uint32_t FPS_JMASK = 0x7FFFFFu;  // 23 bits of point index

struct FpsSlot {
  uint32_t d;
  uint32_t tie;
  float x, y, z;
  uint32_t pad[3];
};

// Block-wide argmax exchange.  Input: this wave's (wmax, wtie) and the
// coordinates of its candidate (wave-uniform values).  Output: block winner.
template <int WAVES>
__device__ __forceinline__ void fps_block_exchange(FpsSlot* slots, int wave,
                                                   int lane, uint32_t wmax,
                                                   uint32_t wtie, float sx,
                                                   float sy, float sz,
                                                   int& cur, float& cx,
                                                   float& cy, float& cz) {
  if (lane == 0) {
    FpsSlot s;
    s.d = wmax;
    s.tie = wtie;
    s.x = sx;
    s.y = sy;
    s.z = sz;
    s.pad[0] = s.pad[1] = s.pad[2] = 0;
    slots[wave] = s;
  }
  __syncthreads();
  const FpsSlot s = slots[lane & (WAVES - 1)];
  const uint32_t bmax = row16_max_u32(s.d);
  // lanes 0..WAVES-1 hold the distinct slots; a unique maximum (the usual
  // case) needs no tie reduction
  uint64_t win = __ballot(s.d == bmax) & ((1ull << WAVES) - 1ull);
  uint32_t btie;
  if (__popcll(win) > 1) {
    const uint32_t cand = (s.d == bmax) ? s.tie : 0xFFFFFFFFu;
    btie = __builtin_amdgcn_readlane(row16_min_u32(cand), 0);
    win = __ballot(s.d == bmax && s.tie == btie);
  } else {
    btie = __builtin_amdgcn_readlane(s.tie, __ffsll((unsigned long long)win) - 1);
  }
  const int wl = __ffsll((unsigned long long)win) - 1;  // lane < 16, uniform
  cur = (int)(btie & FPS_JMASK);
  cx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.x), wl));
  cy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.y), wl));
  cz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.z), wl));
}

// Read (x[pw], y[pw], z[pw]) of lane `wl` into wave-uniform values.  `pw` is
// wave-uniform, so this is a scalar branch tree over STATIC register indices
// (a runtime-indexed register array would be demoted to scratch).
template <int PPT, int LO, int HI>
__device__ __forceinline__ void fps_pick(const float (&x)[PPT],
                                         const float (&y)[PPT],
                                         const float (&z)[PPT], int pw, int wl,
                                         float& sx, float& sy, float& sz) {
  if constexpr (HI - LO == 1) {
    sx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[LO]), wl));
    sy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y[LO]), wl));
    sz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(z[LO]), wl));
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (pw < MID)
      fps_pick<PPT, LO, MID>(x, y, z, pw, wl, sx, sy, sz);
    else
      fps_pick<PPT, MID, HI>(x, y, z, pw, wl, sx, sy, sz);
  }
}

// THREADS must be a multiple of the reference block size bs (<= 512) so that a
// thread's points share j mod bs; launch_fps only picks such combinations.
template <int THREADS, int PPT, bool FMAD, typename IdxT>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(
    const float* __restrict__ xyz, int N, int M, IdxT* __restrict__ idx,
    float* __restrict__ ctr, int lg_bs) {
  constexpr int WAVES = THREADS / 64;
  __shared__ FpsSlot slots[2][FPS_MAX_WAVES];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const float* __restrict__ px = xyz + (size_t)b * 3 * N;
  const float* __restrict__ py = px + N;
  const float* __restrict__ pz = py + N;
  IdxT* __restrict__ out = idx + (size_t)b * M;
  float* __restrict__ cout = ctr ? ctr + (size_t)b * 3 * M : nullptr;

  float x[PPT], y[PPT], z[PPT], md[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int j = t + THREADS * p;
    const bool ok = j < N;
    const int jj = ok ? j : 0;
    x[p] = px[jj];
    y[p] = py[jj];
    z[p] = pz[jj];
    // -1 on padding lanes: min(-1, d) stays negative and never beats best=0.
    md[p] = ok ? __builtin_inff() : -1.0f;
  }
  const uint32_t bs_mask = (1u << lg_bs) - 1u;
  const uint32_t rkey = (__brev((uint32_t)t & bs_mask) >> (32 - lg_bs)) << 23;

  int cur = 0;
  float cx = px[0], cy = py[0], cz = pz[0];
  if (t == 0) {
    out[0] = 0;
    if (cout) {
      cout[0] = cx;
      cout[M] = cy;
      cout[2 * M] = cz;
    }
  }

  for (int i = 1; i < M; ++i) {
    float best = 0.0f;
    int bestp = -1;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float d = dist2<FMAD>(cx, cy, cz, x[p], y[p], z[p]);
      float m;  // one v_min_f32 (== (d < md) ? d : md for the non-NaN contract; fminf adds a canonicalising v_max)
      asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(d), "v"(md[p]));
      md[p] = m;
      if (m > best) {
        best = m;
        bestp = p;
      }
    }
    const uint32_t jbest =
        (bestp < 0) ? (uint32_t)cur : (uint32_t)(t + THREADS * bestp);
    const uint32_t tie = rkey | jbest;
    const uint32_t dbits = __float_as_uint(best);
    const uint32_t wmax = wave_max_u32(dbits);
    uint64_t win = __ballot(dbits == wmax);
    uint32_t wtie;
    if (__popcll(win) > 1) {  // exact tie inside the wave: reference tie rule
      wtie = wave_min_u32((dbits == wmax) ? tie : 0xFFFFFFFFu);
      win = __ballot(dbits == wmax && tie == wtie);
    } else {
      wtie = __builtin_amdgcn_readlane(tie, __ffsll((unsigned long long)win) - 1);
    }
    const int wl = __ffsll((unsigned long long)win) - 1;
    const int pw = __builtin_amdgcn_readlane(bestp, wl);
    // coordinates of this wave's candidate: static register index per case.
    float sx = cx, sy = cy, sz = cz;
    if (pw >= 0) fps_pick<PPT, 0, PPT>(x, y, z, pw, wl, sx, sy, sz);
    fps_block_exchange<WAVES>(slots[i & 1], wave, lane, wmax, wtie, sx, sy, sz, cur,
                       cx, cy, cz);
    if (t == 0) {
      out[i] = (IdxT)cur;
      if (cout) {  // centroid gather fused in: the winner's xyz is already here
        cout[i] = cx;
        cout[M + i] = cy;
        cout[2 * M + i] = cz;
      }
    }
  }
}

// Streaming fallback: any N < 2^23.  min-distance in `temp` (B,N) fp32.
template <bool FMAD, typename IdxT>
__global__ __launch_bounds__(FPS_THREADS) void fps_stream_kernel(
    const float* __restrict__ xyz, int N, int M, IdxT* __restrict__ idx,
    float* __restrict__ ctr, float* __restrict__ temp, int lg_bs) {
  __shared__ FpsSlot slots[2][FPS_MAX_WAVES];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const float* __restrict__ px = xyz + (size_t)b * 3 * N;
  const float* __restrict__ py = px + N;
  const float* __restrict__ pz = py + N;
  float* __restrict__ md = temp + (size_t)b * N;
  IdxT* __restrict__ out = idx + (size_t)b * M;
  float* __restrict__ cout = ctr ? ctr + (size_t)b * 3 * M : nullptr;

  for (int j = t; j < N; j += FPS_THREADS) md[j] = __builtin_inff();
  const uint32_t bs_mask = (1u << lg_bs) - 1u;
  const uint32_t rkey = (__brev((uint32_t)t & bs_mask) >> (32 - lg_bs)) << 23;

  int cur = 0;
  float cx = px[0], cy = py[0], cz = pz[0];
  if (t == 0) {
    out[0] = 0;
    if (cout) {
      cout[0] = cx;
      cout[M] = cy;
      cout[2 * M] = cz;
    }
  }

  for (int i = 1; i < M; ++i) {
    float best = 0.0f;
    int bestj = -1;
    for (int j = t; j < N; j += FPS_THREADS) {
      const float d = dist2<FMAD>(cx, cy, cz, px[j], py[j], pz[j]);
      const float o = md[j];
      const float m = (d < o) ? d : o;
      if (d < o) md[j] = m;
      if (m > best) {
        best = m;
        bestj = j;
      }
    }
    const uint32_t jbest = (bestj < 0) ? (uint32_t)cur : (uint32_t)bestj;
    const uint32_t tie = rkey | jbest;
    const uint32_t dbits = __float_as_uint(best);
    const uint32_t wmax = wave_max_u32(dbits);
    const uint32_t wtie = wave_min_u32((dbits == wmax) ? tie : 0xFFFFFFFFu);
    const int jw = (int)(wtie & FPS_JMASK);  // uniform
    const float sx = px[jw], sy = py[jw], sz = pz[jw];
    fps_block_exchange<FPS_THREADS / 64>(slots[i & 1], wave, lane, wmax, wtie, sx, sy, sz, cur,
                       cx, cy, cz);
    if (t == 0) {
      out[i] = (IdxT)cur;
      if (cout) {  // centroid gather fused in: the winner's xyz is already here
        cout[i] = cx;
        cout[M + i] = cy;
        cout[2 * M + i] = cz;
      }
    }
  }
}

static int ref_block_lg(int64_t n) {
  // get_block() of sampling_kernel.cu:34-42 with the switch's 16-thread floor.
  int cnt = 0;
  int64_t x = n - 1;
  while (x > 0) {
    x >>= 1;
    ++cnt;
  }
  if (cnt > 9) cnt = 9;
  if (cnt < 4) cnt = 4;
  return cnt;
}

template <bool FMAD, typename IdxT>
static int launch_fps(const float* xyz, int64_t B, int64_t N, int64_t M,
                      IdxT* idx, float* ctr, void* ws, size_t ws_bytes,
                      hipStream_t stream) {
  const int lg = ref_block_lg(N);
  const dim3 grid((unsigned)B);
  int variant = 0;  // S4G_FPS_VARIANT=1024x25 | 512x50 | 256x100 (tuning knob)
  if (const char* e = getenv("S4G_FPS_THREADS")) variant = atoi(e);
#define S4G_FPS_CASE(T, P)                                                   \
  if (N <= (int64_t)T * P) {                                                 \
    hipLaunchKernelGGL((fps_reg_kernel<T, P, FMAD, IdxT>), grid, dim3(T), 0, \
                       stream, xyz, (int)N, (int)M, idx, ctr, lg);           \
    S4G_LAUNCH_CHECK();                                                      \
    return S4G_OK;                                                           \
  }
  if (variant == 1024) {
    S4G_FPS_CASE(1024, 25)
  }
  // small levels: fewer waves = cheaper exchange (measured: 512x2 beats 1024x1 at N=1024)
  S4G_FPS_CASE(256, 1)
  S4G_FPS_CASE(512, 1)
  S4G_FPS_CASE(512, 2)
  S4G_FPS_CASE(512, 5)
  S4G_FPS_CASE(512, 10)
  S4G_FPS_CASE(512, 20)
  S4G_FPS_CASE(512, 32)
  S4G_FPS_CASE(512, 50)
#undef S4G_FPS_CASE
  if (ws_bytes < (size_t)B * (size_t)N * sizeof(float) || ws == nullptr)
    return S4G_EWORKSPACE;
  hipLaunchKernelGGL((fps_stream_kernel<FMAD, IdxT>), grid, dim3(FPS_THREADS), 0, stream,
                     xyz, (int)N, (int)M, idx, ctr, (float*)ws, lg);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

size_t fps_workspace_bytes(int64_t B, int64_t N) {
  if (N <= (int64_t)512 * 50) return 0;
  return (size_t)B * (size_t)N * sizeof(float);
}

}  // namespace s4g

extern "C" int s4g_fps_f32(const float* xyz_b3n, int64_t B, int64_t N,
                           int64_t M, int64_t* idx_bm, void* ws,
                           size_t ws_bytes, int flags, s4g_stream_t stream) {
  if (B < 0 || M <= 0 || N < M || N >= (1 << 23)) return S4G_EINVAL;
  if (B == 0) return S4G_OK;
  if (!xyz_b3n || !idx_bm) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (flags & S4G_FLAG_FMAD)
    return s4g::launch_fps<true, int64_t>(xyz_b3n, B, N, M, idx_bm, nullptr, ws, ws_bytes, st);
  return s4g::launch_fps<false, int64_t>(xyz_b3n, B, N, M, idx_bm, nullptr, ws, ws_bytes, st);
}

extern "C" int s4g_fps_gather_i32(const float* xyz_b3n, int64_t B, int64_t N,
                                  int64_t M, int32_t* idx_bm, float* ctr_b3m,
                                  void* ws, size_t ws_bytes, int flags,
                                  s4g_stream_t stream) {
  if (B < 0 || M <= 0 || N < M || N >= (1 << 23)) return S4G_EINVAL;
  if (B == 0) return S4G_OK;
  if (!xyz_b3n || !idx_bm || !ctr_b3m) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (flags & S4G_FLAG_FMAD)
    return s4g::launch_fps<true, int32_t>(xyz_b3n, B, N, M, idx_bm, ctr_b3m, ws, ws_bytes, st);
  return s4g::launch_fps<false, int32_t>(xyz_b3n, B, N, M, idx_bm, ctr_b3m, ws, ws_bytes, st);
}
