// Farthest point sampling for gfx950.
//
// Replaces FarthestPointSampleKernel (reference
// pointnet2_utils/csrc/sampling_kernel.cu:49-119, host :128-172).
//
// Design (MI355X-first, not the reference's shape).  FPS is a chain of M dependent steps; what a
// step costs is latency, so every variant keeps the state of a scene on ONE CU (or two) and
// touches memory as little as possible:
//   * fps_reg_kernel        N <= 25 600: one 512-thread workgroup per scene, xyz AND the running
//                           min-distances in VGPRs (50 points x 4 floats per lane); a step = the
//                           distance updates, a DPP wave-argmax and ONE barrier with a
//                           double-buffered LDS exchange that also carries the winner's
//                           coordinates (no dependent global load);
//   * fps_pruned_kernel     10 240 < N <= 25 600 (default there): the scene in Morton order, 64-point
//                           groups with a box and an exact maximum each; a step rescans only the
//                           groups the new centroid can still change (exact: the bound uses the
//                           distance contract's own monotone arithmetic); up to four picks per
//                           exchange (candidates re-ranked exactly, fps_block_exchange_multi);
//   * fps_pruned_l2_kernel  25 600 < N <= 51 200 (default there): the same, with only the
//                           min-distances resident and the coordinates of a touched group read
//                           from the Morton-sorted records in L2 -- one CU per scene;
//   * fps_stream_kernel     any other size (and 25 600 < N <= 51 200 without a workspace); the two full-scan
//                           kernels for that range (variants/fps_fullscan_51k.inc: S4G_FPS_MODE=cluster|hybrid)
//                           are compiled into measurement builds only (-DS4G_VARIANTS).
// The reference's tie rule is reproduced exactly through a composite key: maximise d, then
// minimise (bitrev_{log2 bs}(j mod bs) << 23 | j), where bs = clamp(pow2ceil(N),16,512) is the
// REFERENCE's block size (sampling_kernel.cu:34-42,148-167) -- see SURVEY.md Appendix A.1.  In
// the full-scan kernels thread t owns points j = t + THREADS p, so all its points share j mod bs
// and an ascending-p strict '>' scan resolves ties inside a thread.
#include <stdlib.h>
#include <string.h>


#include "s4g_common.h"

namespace s4g {

constexpr int FPS_THREADS = 1024;  // streaming fallback
constexpr int FPS_TIE_PAR = 5;     // tying groups per wave from which the pruned kernel resolves ties lane-parallel
constexpr int FPS_MAX_WAVES = 16;
constexpr uint32_t FPS_JMASK = 0x7FFFFFu;  // 23 bits of point index

typedef float fps_v2f __attribute__((ext_vector_type(2)));

struct FpsSlot {
  uint32_t d;
  uint32_t tie;
  float x, y, z;
  uint32_t d2;      // two-pick exchange: distance bits of the wave's runner-up (else 0)
  uint32_t key2;    // ... and, where the runner-up TIES with the candidate (d2 == d), its tie key; 0 = not known
  uint32_t pad;
};

// Optional per-call extras of the register / pruned kernels: dist (B, M) receives every pick's
// min-distance at the time it was taken (+inf for pick 0); run (B) = 0 marks scenes whose result is
// already known to be the identity prefix 0..M-1 (fps_prefix_check_kernel) -- the workgroup writes
// that and leaves.
struct FpsExtra {
  float* dist;
  const int* run;
};

#ifdef S4G_FPS_STAMPS
// debug build only (make HIPFLAGS_EXTRA=-DS4G_FPS_STAMPS into its own OBJDIR / LIB): per (scene, wave)
// cycle accumulators of fps_pruned_kernel's phases, read back by tools/fps_stamps.py.  The sums are
// kept in (wave-uniform) registers and written ONCE at the end of the kernel: a read-modify-write
// of global memory at every phase boundary would stall the wave for a memory round trip each
// time and end up in the numbers.
__device__ unsigned long long g_fps_acc[64 * 8 * 8];
#define S4G_FPS_T() __builtin_amdgcn_s_memtime()
#define S4G_FPS_ACC_DECL() unsigned long long fps_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define S4G_FPS_ACC(i, v) fps_acc_[(i)] += (unsigned long long)(v)
#define S4G_FPS_ACC_FLUSH()                                                              \
  do {                                                                                   \
    if (lane == 0 && b < 64)                                                             \
      for (int a_ = 0; a_ < 8; ++a_) g_fps_acc[(b * 8 + wave) * 8 + a_] += fps_acc_[a_]; \
  } while (0)
#else
#define S4G_FPS_T() 0ull
#define S4G_FPS_ACC_DECL()
#define S4G_FPS_ACC(i, v)
#define S4G_FPS_ACC_FLUSH()
#endif

// Block-wide argmax exchange.  Input: this wave's (wmax, wtie) and the
// coordinates of its candidate (wave-uniform values).  Output: block winner.
template <int WAVES>
__device__ __forceinline__ void fps_block_exchange(FpsSlot* slots, int wave,
                                                   int lane, uint32_t wmax,
                                                   uint32_t wtie, float sx,
                                                   float sy, float sz,
                                                   int& cur, float& cx,
                                                   float& cy, float& cz,
                                                   uint32_t* dwin = nullptr) {
  if (lane == 0) {
    FpsSlot s;
    s.d = wmax;
    s.tie = wtie;
    s.x = sx;
    s.y = sy;
    s.z = sz;
    s.d2 = s.key2 = s.pad = 0;
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
  if (dwin) *dwin = __builtin_amdgcn_readfirstlane(bmax);   // the pick's min-distance (bits)
  cx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.x), wl));
  cy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.y), wl));
  cz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.z), wl));
}

// Every wave also publishes the distance d2 of its RUNNER-UP (its best point other than its candidate):
// the bound that lets one exchange settle several picks.
// Up to MAXP picks per exchange (round 3).  After the barrier every wave knows all WAVES candidates
// (position, min-distance, tie key) and each wave's runner-up distance d2.  Picks are taken one after
// the other from the candidates ONLY, and each is the sequential algorithm's next pick as long as it
// beats everything that is not a candidate:
//   * a candidate's min-distance after a pick X is known exactly: min(old, dist2(X, candidate)) with
//     the update's own arithmetic (the positions are here), so the candidates can be re-ranked;
//   * the other points of a wave are bounded by its candidate's OLD value while that candidate is
//     untouched (they were below it and only decrease), and by d2 once its candidate has been
//     picked or lowered ("disturbed").
// So pick k (k >= 1) = argmax of the re-ranked candidates by (distance, tie key), valid iff it is
// STRICTLY above the largest d2 of all disturbed waves; ties between candidates resolve by the key
// as in the reference, and a tie with an untouched wave's other points is impossible to lose (their
// keys are larger than their candidate's, which lost to the argmax).  The centroids' updates
// commute, so all picks of an exchange are applied together before the next candidates are taken.
// Everything is wave-uniform and every wave decides alike.  Returns the number of picks (>= 1).
template <int WAVES, bool FMAD, int MAXP, typename IdxT>
__device__ __forceinline__ int fps_block_exchange_multi(FpsSlot* slots, int wave, int lane, uint32_t wmax,
                                                        uint32_t wtie, uint32_t wd2, uint32_t wkey2, float sx, float sy,
                                                        float sz, int limit, int& cur, float& cx, float& cy,
                                                        float& cz, uint32_t& picked_waves,
                                                        IdxT* __restrict__ out_i, float* __restrict__ cout_i,
                                                        int M, float& fx, float& fy, float& fz,
                                                        float* __restrict__ dout_i = nullptr) {
  static_assert(WAVES == 8, "eight entries: the pick x candidate matrix is the wave's 64 lanes");
  if (lane == 0) {
    FpsSlot s;
    s.d = wmax;
    s.tie = wtie;
    s.x = sx;
    s.y = sy;
    s.z = sz;
    s.d2 = wd2;
    s.key2 = wkey2;
    s.pad = 0;
    slots[wave] = s;
  }
  __syncthreads();
  const int ej = lane & 7;
  const FpsSlot s = slots[ej];                   // candidate j = lane % 8 (every row of 8 lanes holds all eight)
  const FpsSlot sp = slots[lane >> 3];           // pick i = lane / 8
  // dist2(pick, candidate): the update's own arithmetic and argument order
  const uint32_t dm = __float_as_uint(dist2<FMAD>(sp.x, sp.y, sp.z, s.x, s.y, s.z));
  auto max8 = [](uint32_t x) {                   // max over the 8 lanes of a row half (all of them end up with it)
    x = max(x, dpp_u32<0xB1>(x));
    x = max(x, dpp_u32<0x4E>(x));
    x = max(x, dpp_u32<0x141>(x));
    return x;
  };
  uint32_t v = s.d;          // this entry's current min-distance bits (0: picked / nothing left)
  uint32_t bnd = 0u;         // largest runner-up distance of the disturbed waves
  int np = 0, xl = 0, xl0 = 0;
  uint32_t pw = 0u;          // picked entries (= wave numbers), 4 bits each
  uint32_t my_d = 0u;
  const bool upper = (lane & 8) != 0;   // lanes 8..15 of a row reduce the runner-ups while lanes 0..7 reduce v
  uint32_t bmax = max8(v);
  // one pick: the best candidate by (distance, key), then -- unless it was the last one this exchange may take --
  // the other candidates' values after it and the two maxima of the next round
  auto take = [&](int k) {
    uint32_t win = (uint32_t)__ballot(v == bmax) & 0xFFu;
    if (win & (win - 1u)) {   // equal distances: the smaller key (sampling_kernel.cu's tie rule)
      uint32_t cand = (v == bmax) ? s.tie : 0xFFFFFFFFu;
      cand = min(cand, dpp_u32<0xB1>(cand));
      cand = min(cand, dpp_u32<0x4E>(cand));
      cand = min(cand, dpp_u32<0x141>(cand));
      win = (uint32_t)__ballot(v == bmax && s.tie == cand) & 0xFFu;
    }
    xl = __ffs(win) - 1;
    if (k == 0) xl0 = xl;
    pw |= (uint32_t)xl << (4 * k);
    np = k + 1;
    my_d = lane == k ? bmax : my_d;   // the pick's min-distance at the time it is taken
    if (k + 1 < MAXP) {
      // the picked wave is disturbed; the other candidates' values after this pick, exactly
      const uint32_t nd = (uint32_t)__builtin_amdgcn_ds_bpermute((xl * 8 + ej) * 4, (int)dm);
      const bool me = ej == xl;
      const bool hit = me || nd < v;             // (v == 0: nothing is below it)
      v = me ? 0u : min(v, nd);
      // ONE reduction for both maxima: the next round's bmax in lanes 0..7, the runner-ups in 8..15
      const uint32_t red = max8(upper ? (hit ? s.d2 : 0u) : v);
      bnd = max(bnd, (uint32_t)__builtin_amdgcn_readlane(red, 8));
      bmax = red;   // (valid in lanes 0..7 of every row: the only lanes the ballots and lane k look at)
    }
  };
  bool at_bound = false;     // the sequence ended on a candidate EQUAL to the disturbed waves' runner-up distance
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    if (k >= limit) break;
    if (k > 0) {
      const uint32_t bmax_s = __builtin_amdgcn_readfirstlane(bmax);
      if (bmax_s == 0u || bmax_s <= bnd) {
        at_bound = bmax_s != 0u && bmax_s == bnd;
        break;
      }
    }
    take(k);
  }
  if (__builtin_expect(at_bound, 0)) {
    // (round 4) Lattice clouds: a whole shell of points holds the maximum, so the best remaining candidate TIES
    // with a disturbed wave's runner-up.  The sequential algorithm takes the smallest key among all holders; a
    // disturbed wave's other holders have keys >= the key2 it reported, so the candidate is the next pick iff
    // its key is below every such key2 (0 = not reported: stop, as before round 4).  Rolled and out of the
    // unrolled loop above: the tie-free exchange is the code of round 3.
#pragma unroll 1
    for (int k = np; k < MAXP && k < limit; ++k) {
      const uint32_t bmax_s = __builtin_amdgcn_readfirstlane(bmax);
      if (bmax_s == 0u || bmax_s < bnd) break;
      if (bmax_s == bnd) {
        uint32_t wkey = (v == bmax) ? s.tie : 0xFFFFFFFFu;
        wkey = min(wkey, dpp_u32<0xB1>(wkey));
        wkey = min(wkey, dpp_u32<0x4E>(wkey));
        wkey = min(wkey, dpp_u32<0x141>(wkey));
        const bool was_hit = v != s.d;            // picked (0) or lowered by a pick of this exchange
        uint32_t kb = (was_hit && s.d2 == bnd) ? slots[ej].key2 : 0xFFFFFFFFu;
        kb = min(kb, dpp_u32<0xB1>(kb));
        kb = min(kb, dpp_u32<0x4E>(kb));
        kb = min(kb, dpp_u32<0x141>(kb));
        if (!((uint32_t)__builtin_amdgcn_readlane(wkey, 0) < (uint32_t)__builtin_amdgcn_readlane(kb, 0))) break;
      }
      take(k);
    }
  }
  // the callers' running centroid = the last pick (and the first one for the MAXP == 2 kernels)
  cur = (int)((uint32_t)__builtin_amdgcn_readlane(s.tie, xl) & FPS_JMASK);
  cx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.x), xl));
  cy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.y), xl));
  cz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.z), xl));
  fx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.x), xl0));
  fy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.y), xl0));
  fz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(s.z), xl0));
  if (wave == 0 && lane < np) {   // lane k stores pick k: ONE store per output array
    const FpsSlot e = slots[(pw >> (4 * lane)) & 15u];
    out_i[lane] = (IdxT)(e.tie & FPS_JMASK);
    if (cout_i) {
      cout_i[lane] = e.x;
      cout_i[M + lane] = e.y;
      cout_i[2 * M + lane] = e.z;
    }
    if (dout_i) dout_i[lane] = __uint_as_float(my_d);
  }
  picked_waves = pw;
  return np;
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
    float* __restrict__ ctr, int lg_bs, int M_run, float* __restrict__ md_out, FpsExtra ex) {
  // M_run <= M steps are computed (outputs keep stride M); md_out (or NULL) receives the
  // running min-distances afterwards -- the hand-over to fps_pruned_kernel
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
  float* __restrict__ dout = ex.dist ? ex.dist + (size_t)b * M : nullptr;
  if (ex.run && ex.run[b] == 0) {
    // proven elsewhere (fps_prefix_check_kernel): this scene's result is the identity prefix
    for (int i = t; i < M_run; i += THREADS) {
      out[i] = (IdxT)i;
      if (cout) {
        cout[i] = px[i];
        cout[M + i] = py[i];
        cout[2 * M + i] = pz[i];
      }
    }
    return;
  }

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
    if (dout) dout[0] = __builtin_inff();
  }

  for (int i = 1; i < M_run; ++i) {
    float best = 0.0f;
    int bestp = -1;
    // two points per instruction where the ISA has packed fp32 forms (v_pk_add / v_pk_mul /
    // v_pk_fma: IEEE per element, no contraction): 8 instead of 10 instructions per point
    const fps_v2f cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
#pragma unroll
    for (int p = 0; p + 1 < PPT; p += 2) {
      const fps_v2f xv = {x[p], x[p + 1]}, yv = {y[p], y[p + 1]}, zv = {z[p], z[p + 1]};
      const fps_v2f dx = xv - cx2, dy = yv - cy2, dz = zv - cz2;
      fps_v2f d;
      if constexpr (FMAD) {
        d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      } else {
        d = (dx * dx + dy * dy) + dz * dz;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float m;  // one v_min_f32 (== (d < md) ? d : md for the non-NaN contract; fminf adds a canonicalising v_max)
        asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(d[e]), "v"(md[p + e]));
        md[p + e] = m;
        if (m > best) {
          best = m;
          bestp = p + e;
        }
      }
    }
    if constexpr (PPT & 1) {
      constexpr int p = PPT - 1;
      const float d = dist2<FMAD>(cx, cy, cz, x[p], y[p], z[p]);
      float m;
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
    uint32_t dwin;
    fps_block_exchange<WAVES>(slots[i & 1], wave, lane, wmax, wtie, sx, sy, sz, cur,
                       cx, cy, cz, &dwin);
    if (t == 0) {
      out[i] = (IdxT)cur;
      if (cout) {  // centroid gather fused in: the winner's xyz is already here
        cout[i] = cx;
        cout[M + i] = cy;
        cout[2 * M + i] = cz;
      }
      if (dout) dout[i] = __uint_as_float(dwin);
    }
  }
  if (md_out) {
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const int j = t + THREADS * p;
      if (j < N) md_out[(size_t)b * N + j] = md[p];
    }
  }
}

#ifdef S4G_VARIANTS
#include "variants/fps_fullscan_51k.inc"
#endif

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


// ---------------------------------------------------------------------------
// Pruned variant (N > 5 120): same per-step exchange, but a step only rescans the
// 64-point groups the new centroid can still change.
//
// The scene is put in a spatial order first (fps_cell_sort_kernel: one launch; round 2:
// bbox / Morton-key kernels + rocPRIM's radix sort); group g = sorted positions [64 g, 64 g + 64) is slot g / 8 of wave
// g % 8, so a group is a compact patch and neighbouring patches sit in different
// waves.  Lane l of a wave keeps, for "its" group (slot l), the bounding box and
// the exact maximum Mg of the group's running min-distances.  A new centroid c
// cannot lower any min-distance of a group whose box is at squared distance
//   LB = ((tx*tx) + (ty*ty)) + (tz*tz) >= Mg,   t = lo - c | hi - c | 0 per axis,
// because every fp32 operation of the distance contract is monotone: LB is a lower
// bound of the ROUNDED distance of every point in the box.  Skipped groups keep
// their min-distances and Mg bit for bit, so the selected index sequence is the
// one of the full scan (tests compare against the oracle).  On the bench scenes a
// centroid touches 5.3 of 400 groups on average (8.7 in round 2's Morton order).  The
// kernel starts by itself from +inf min-distances (md_in == NULL; round 3 -- measured
// equal alone, and one launch of a whole-CU workgroup fewer inside a pipelined step);
// S4G_FPS_DENSE_STEPS = n > 1 has the full-scan kernel run the first n steps and hand its
// min-distances over through the workspace, as rounds 1-2 did.  Updates are dispatched by straight-line
// guarded blocks (one rarely-taken scalar branch per 8 slots, one per slot) with static
// register indices; a group's maximum is only re-reduced when a point that held it came
// closer; the winner's slot is fetched by one walk of a 6-level scalar branch tree.
// 25 600 -> 5 120: 4.8 ms against 10.9 ms for the full scan (S4G_FPS_MODE=dense).
// The winner is resolved lazily: wave max over the Mg lanes, then the lane(s) of
// that group that hold it (static register index through a scalar branch tree),
// original index and tie key from an LDS table.
// ---------------------------------------------------------------------------
constexpr int FPS_DENSE_STEPS = 48;

__device__ __forceinline__ uint32_t f32_ordered(float v) {   // order-preserving float -> uint
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_f32(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}




// The whole pre-pass of the pruned kernels as ONE launch, one workgroup per scene: bounding box,
// a 15-bit cell key per point, a counting sort by cell in LDS, the permutation, the box of every
// 64-point group of that order and (N > 25 600) the sorted (x, y, z, index) records.  (Round 2
// used five kernels around rocprim::radix_sort_pairs: 22 launches per batch on the geometry
// stream.)
//
// The key: the pruned kernels are exact for ANY permutation, the order only decides how tight the
// groups' boxes are (touched groups per pick, simulated on the bench scenes over all 5 119 picks;
// round 2's 10-bit-per-axis Morton key on box-normalised coordinates: 8.7 at 25 600 points).
//  * thin clouds (smallest extent < half the largest: depth-camera scenes are surfaces over a
//    table): a 2-D Hilbert curve over the two long axes, 64 x 64 square cells, the short axis as 3
//    minor bits -- 5.3 touched groups per pick.  Hilbert, not Morton: no long jumps inside a group.
//  * otherwise: 15 Morton bits dealt to the axes by extent -- the next bit always halves the axis
//    whose cells are currently longest -- so cells come out near-cubic (7.0 on the table-top
//    scene, where it would deal 6 + 6 + 3; 15.5 against 16.0 on a uniform box).
// The order INSIDE a cell is whatever order the LDS atomics were served in -- as in the ball
// query's grid build -- so the permutation is not reproducible run to run; the FPS output is.
//
// 2^15 counters as 16-bit halves of 2^14 words (cell c lives in half c >> 14 of word c & 16383;
// a half never exceeds N <= 65 535): one packed prefix sum over the words yields both halves'
// starts, the upper halves shifted by the lower halves' total.
constexpr int FPS_SORT_THREADS = 1024;
constexpr int FPS_SORT_WORDS = 1 << 14;
constexpr int FPS_SORT_BITS = 15;
constexpr int FPS_SORT_AXIS_BITS = 8;   // at most 8 bits per axis (256-entry deposit tables)

__global__ __launch_bounds__(FPS_SORT_THREADS) void fps_cell_sort_kernel(
    const float* __restrict__ xyz, int N, int G, int* __restrict__ perm, float* __restrict__ gbox,
    float4* __restrict__ aos, int aos_cap) {
  extern __shared__ uint32_t sort_lds[];
  constexpr int NW = FPS_SORT_THREADS / 64;
  uint32_t* __restrict__ H = sort_lds;                  // [2^14] packed counters
  uint32_t* __restrict__ dep = H + FPS_SORT_WORDS;      // [3][256] an axis value's bits at their key positions
  uint16_t* __restrict__ hil = reinterpret_cast<uint16_t*>(dep + 3 * 256);   // [64 * 64] Hilbert index of an (a, b) cell
  uint32_t* __restrict__ red = dep + 3 * 256 + 64 * 64 / 2;   // [6][NW] + [NW] + the re-boxed ranges [3][2]
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* __restrict__ p = xyz + (size_t)b * 3 * N;
  int* __restrict__ pm = perm + (size_t)b * N;

  // the three point loops keep U x 3 plane loads in flight (the workgroup is latency-bound)
  constexpr int U = 6;
  auto for_points = [&](auto&& body) {
    for (int j0 = t; j0 < N; j0 += FPS_SORT_THREADS * U) {
      float c[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u * FPS_SORT_THREADS;
        const int jj = j < N ? j : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) c[u][a] = p[(size_t)a * N + jj];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (j0 + u * FPS_SORT_THREADS < N) body(j0 + u * FPS_SORT_THREADS, c[u]);
    }
  };

  for (int i = t; i < FPS_SORT_WORDS; i += FPS_SORT_THREADS) H[i] = 0;
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  for_points([&](int, const float (&c)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const uint32_t v = f32_ordered(c[a]);
      lo[a] = min(lo[a], v);
      hi[a] = max(hi[a], v);
    }
  });
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const uint32_t l = wave_min_u32(lo[a]), h = wave_max_u32(hi[a]);
    if (lane == 0) {
      red[a * NW + wave] = l;
      red[(3 + a) * NW + wave] = h;
    }
  }
  __syncthreads();
  float bl[3], ext[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    uint32_t l = red[a * NW], h = red[(3 + a) * NW];
    for (int w = 1; w < NW; ++w) {
      l = min(l, red[a * NW + w]);
      h = max(h, red[(3 + a) * NW + w]);
    }
    bl[a] = ordered_f32(l);
    ext[a] = ordered_f32(h) - bl[a];
    if (!(ext[a] > 0.f) || !(ext[a] < 3.0e38f)) ext[a] = 0.f;   // degenerate / non-finite axis: never split
  }
  // Robust box (round 4).  A few far outliers -- 50 background points 50 m behind a 0.8 m table-top scene --
  // stretched the box so far that the whole scene fell into a handful of cells: no spatial order, group boxes as
  // large as the scene, every group touched by every pick (25 ms instead of 5.4 for 16 scenes, slower than the full
  // scan).  Per axis a 256-bin histogram over [min, max] gives the range that leaves N / 128 points outside on
  // either side; where that range is under half of the extent the axis is re-boxed to it plus a quarter of its
  // width on both sides (twice: the second histogram refines a first one whose bins were as wide as the scene),
  // and the points outside any re-boxed axis share ONE key -- they sort into groups of their own, which die as soon
  // as FPS has picked them (it picks far points first), instead of poisoning 50 groups of regular points.  Compact
  // clouds (uniform, Gaussian, the bench scenes) keep min / max: a trimmed box there would put the whole rim into
  // the outlier groups.  Ordering only: the result is exact for any box.
  float cut_lo[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  float cut_hi[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  {
    uint32_t* __restrict__ hist = H;               // [3][256], zeroed above; re-zeroed before the cell counters use it
    float* __restrict__ rb = reinterpret_cast<float*>(red + 7 * NW);   // [3][2] the re-boxed ranges
    for (int round = 0; round < 2; ++round) {
      __syncthreads();
      float hinv[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) hinv[a] = ext[a] > 0.f ? 256.0f / ext[a] : 0.f;
      for_points([&](int, const float (&c)[3]) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float sc = (c[a] - bl[a]) * hinv[a];
          if (sc >= 0.f && sc <= 256.0f) atomicAdd(&hist[a * 256 + (int)(sc < 255.f ? sc : 255.f)], 1u);   // (outside an earlier round's box, NaN: not counted)
        }
      });
      __syncthreads();
      if (wave < 3) {   // one wave per axis: 4 bins per lane, inclusive scan, first / last bin past the cut
        const int a = wave;
        uint32_t c4[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          c4[k] = hist[a * 256 + 4 * lane + k];
          tot += c4[k];
        }
        uint32_t incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o2 = __shfl_up(incl, off);
          if (lane >= off) incl += o2;
        }
        const uint32_t all = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t cut = (uint32_t)N >> 7;
        uint32_t run = incl - tot;
        int first = 256, last = -1;            // first bin whose inclusive count exceeds the cut, last bin with > cut behind-or-in it
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t before = run;
          run += c4[k];
          if (run > cut && first == 256) first = 4 * lane + k;
          if (all - before > cut) last = 4 * lane + k;
        }
        const int f = (int)wave_min_u32((uint32_t)first);
        const int l = (int)wave_max_u32((uint32_t)(last + 1)) - 1;
        if (lane == 0) {
          const float e = a == 0 ? ext[0] : (a == 1 ? ext[1] : ext[2]);
          const float b0 = a == 0 ? bl[0] : (a == 1 ? bl[1] : bl[2]);
          const float w = e * (1.0f / 256.0f);
          float nlo = b0, nhi = b0 + e;
          if (e > 0.f && l >= f && (float)(l - f + 1) < 128.0f) {   // the trimmed range is under half of the extent
            const float width = (float)(l - f + 1) * w;
            nlo = fmaxf(b0, b0 + (float)f * w - 0.25f * width);
            nhi = fminf(b0 + e, b0 + (float)(l + 1) * w + 0.25f * width);
          }
          rb[2 * a] = nlo;
          rb[2 * a + 1] = nhi;
        }
      }
      __syncthreads();
      bool any = false;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float nlo = rb[2 * a], nhi = rb[2 * a + 1];
        if (nlo > bl[a] || nhi < bl[a] + ext[a]) {
          any = true;
          cut_lo[a] = nlo;
          cut_hi[a] = nhi;
          bl[a] = nlo;
          ext[a] = nhi - nlo;
        }
      }
      for (int i = t; i < 3 * 256; i += FPS_SORT_THREADS) hist[i] = 0;
      if (!any) break;                          // (uniform: every thread read the same rb)
    }
    __syncthreads();
  }
  const bool reboxed = cut_lo[0] > -__builtin_inff() || cut_lo[1] > -__builtin_inff() || cut_lo[2] > -__builtin_inff() ||
                       cut_hi[0] < __builtin_inff() || cut_hi[1] < __builtin_inff() || cut_hi[2] < __builtin_inff();
  // deal the bits (every thread computes the same sequence): seq = 2 bits per level, MSB level first
  int nb[3] = {0, 0, 0};
  uint32_t seq = 0;
  {
    float edge[3] = {ext[0], ext[1], ext[2]};
#pragma unroll
    for (int i = 0; i < FPS_SORT_BITS; ++i) {
      int a = 0;
      float best = -1.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float e = nb[k] < FPS_SORT_AXIS_BITS ? edge[k] : -1.f;
        if (e > best) {
          best = e;
          a = k;
        }
      }
      seq |= (uint32_t)a << (2 * i);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k == a) {
          ++nb[k];
          edge[k] *= 0.5f;
        }
    }
  }
  // axes by extent: ax_a the longest, ax_c the shortest
  const int ax_a = ext[0] >= ext[1] ? (ext[0] >= ext[2] ? 0 : 2) : (ext[1] >= ext[2] ? 1 : 2);
  const int ax_c = ext[0] < ext[1] ? (ext[0] < ext[2] ? 0 : 2) : (ext[1] < ext[2] ? 1 : 2);
  const int ax_b = ax_a == ax_c ? (ax_a + 1) % 3 : 3 - ax_a - ax_c;   // (all extents equal: a == c == 0)
  const int ax_c2 = ax_a == ax_c ? (ax_a + 2) % 3 : ax_c;
  auto pick = [](const float (&v)[3], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); };
  const bool thin = pick(ext, ax_c2) < 0.5f * pick(ext, ax_a);
  float inv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) inv[a] = ext[a] > 0.f ? (float)(1 << nb[a]) / ext[a] : 0.f;   // ordering only: any rounding will do
  const float inv_ab = pick(ext, ax_a) > 0.f ? 64.0f / pick(ext, ax_a) : 0.f;   // square cells over the two long axes
  const float inv_c = pick(ext, ax_c2) > 0.f ? 8.0f / pick(ext, ax_c2) : 0.f;
  if (thin) {   // Hilbert index of every (x, y) cell of the 64 x 64 grid, four cells per thread
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cell = 4 * t + k;
      int x = cell & 63, y = cell >> 6, d = 0;
#pragma unroll
      for (int sz = 32; sz > 0; sz >>= 1) {
        const int rx = (x & sz) ? 1 : 0, ry = (y & sz) ? 1 : 0;
        d += sz * sz * ((3 * rx) ^ ry);
        if (ry == 0) {
          if (rx == 1) {
            x = 63 - x;
            y = 63 - y;
          }
          const int tmp = x;
          x = y;
          y = tmp;
        }
      }
      hil[cell] = (uint16_t)d;
    }
  } else if (t < 3 * 256) {   // deposit table: value v of axis a -> its bits at the levels dealt to a
    const int a = t >> 8, v = t & 255;
    const int na = a == 0 ? nb[0] : (a == 1 ? nb[1] : nb[2]);
    uint32_t key = 0;
    int used = 0;
#pragma unroll
    for (int i = 0; i < FPS_SORT_BITS; ++i)
      if ((int)((seq >> (2 * i)) & 3u) == a) {
        ++used;
        key |= (uint32_t)((v >> (na - used)) & 1) << (FPS_SORT_BITS - 1 - i);
      }
    dep[t] = key;
  }
  __syncthreads();
  auto quant = [](float sc, float top) -> uint32_t { return (uint32_t)(sc > 0.f ? (sc < top ? sc : top) : 0.f); };   // NaN -> 0
  auto cell_of = [&](const float (&c)[3]) -> uint32_t {
    if (reboxed) {   // outside a re-boxed axis: the outliers' own key (the last cell)
      bool out = false;
#pragma unroll
      for (int a = 0; a < 3; ++a) out = out || c[a] < cut_lo[a] || c[a] > cut_hi[a];
      if (out) return (1u << FPS_SORT_BITS) - 1u;
    }
    if (thin) {
      const uint32_t qa = quant((pick(c, ax_a) - pick(bl, ax_a)) * inv_ab, 63.f);
      const uint32_t qb = quant((pick(c, ax_b) - pick(bl, ax_b)) * inv_ab, 63.f);
      const uint32_t qc = quant((pick(c, ax_c2) - pick(bl, ax_c2)) * inv_c, 7.f);
      return ((uint32_t)hil[qa | (qb << 6)] << 3) | qc;
    }
    uint32_t key = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) key |= dep[a * 256 + quant((c[a] - bl[a]) * inv[a], (float)((1 << nb[a]) - 1))];
    return key;
  };
  for_points([&](int, const float (&c)[3]) {
    const uint32_t cl = cell_of(c);
    atomicAdd(&H[cl & (FPS_SORT_WORDS - 1)], 1u << (16 * (cl >> 14)));
  });
  __syncthreads();
  {  // exclusive prefix over the packed words, 16 consecutive words per thread
    static_assert(FPS_SORT_WORDS == 16 * FPS_SORT_THREADS, "scan layout");
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint4*>(&H[16 * t + 4 * k]);
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t c;
      c = v[k].x; v[k].x = run; run += c;
      c = v[k].y; v[k].y = run; run += c;
      c = v[k].z; v[k].z = run; run += c;
      c = v[k].w; v[k].w = run; run += c;
    }
    uint32_t incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) red[6 * NW + wave] = incl;
    __syncthreads();
    uint32_t base = incl - run, total = 0;
    for (int w = 0; w < NW; ++w) {
      const uint32_t wt = red[6 * NW + w];
      if (w < wave) base += wt;
      total += wt;
    }
    base += (total & 0xFFFFu) << 16;                     // the upper halves start behind all lower ones
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k].x += base; v[k].y += base; v[k].z += base; v[k].w += base;
      *reinterpret_cast<uint4*>(&H[16 * t + 4 * k]) = v[k];
    }
  }
  __syncthreads();
  for_points([&](int j, const float (&c)[3]) {
    const uint32_t cl = cell_of(c);
    const uint32_t sh = 16 * (cl >> 14);
    const uint32_t old = atomicAdd(&H[cl & (FPS_SORT_WORDS - 1)], 1u << sh);
    pm[(old >> sh) & 0xFFFFu] = j;
  });
  __threadfence();                                       // the permutation is read back below
  __syncthreads();

  // boxes: 16 lanes per 64-point group, four points per lane, two groups per lane in flight
  const int l16 = t & 15;
  constexpr int GS = FPS_SORT_THREADS / 16;              // groups per sweep
  for (int g0 = t >> 4; g0 < G; g0 += 2 * GS) {
    int j[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sp = 64 * (g0 + h * GS) + l16 + 16 * k;
        j[h][k] = sp < N ? __hip_atomic_load(&pm[sp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
      }
    float c[2][4][3];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) c[h][k][a] = p[(size_t)a * N + (j[h][k] < 0 ? 0 : j[h][k])];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int g = g0 + h * GS;
      uint32_t bmin[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, bmax[3] = {0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const uint32_t v = f32_ordered(c[h][k][a]);
          bmin[a] = min(bmin[a], j[h][k] < 0 ? 0xFFFFFFFFu : v);
          bmax[a] = max(bmax[a], j[h][k] < 0 ? 0u : v);
        }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const uint32_t l = row16_min_u32(bmin[a]), hh = row16_max_u32(bmax[a]);
        if (l16 == 0 && g < G) {
          float* __restrict__ o = gbox + ((size_t)b * G + g) * 6;
          o[a] = ordered_f32(l);
          o[3 + a] = ordered_f32(hh);
        }
      }
    }
  }
  if (aos) {   // cell-ordered records of fps_pruned_l2_kernel
    float4* __restrict__ out = aos + (size_t)b * aos_cap;
    constexpr int AU = 5;
    for (int sp0 = t; sp0 < aos_cap; sp0 += FPS_SORT_THREADS * AU) {
      int j[AU];
#pragma unroll
      for (int u = 0; u < AU; ++u) {
        const int sp = sp0 + u * FPS_SORT_THREADS;
        j[u] = sp < N ? __hip_atomic_load(&pm[sp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      }
      float c[AU][3];
#pragma unroll
      for (int u = 0; u < AU; ++u)
#pragma unroll
        for (int a = 0; a < 3; ++a) c[u][a] = p[(size_t)a * N + j[u]];
#pragma unroll
      for (int u = 0; u < AU; ++u) {
        const int sp = sp0 + u * FPS_SORT_THREADS;
        if (sp < aos_cap) out[sp] = make_float4(c[u][0], c[u][1], c[u][2], __int_as_float(j[u]));
      }
    }
  }
}

// max(a, b, c) as ONE instruction; NaN operands lose (the boxes of empty groups are NaN)
__device__ __forceinline__ float fps_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// md[pw] of every lane for a wave-uniform slot pw (static register index per leaf)
template <int PPT, int LO, int HI>
__device__ __forceinline__ float fps_pick_md(const float (&md)[PPT], int pw) {
  if constexpr (HI - LO == 1) {
    float v = md[LO];
    asm volatile("; fps pick %1" : "+v"(v) : "n"(LO));
    return v;
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (pw < MID) return fps_pick_md<PPT, LO, MID>(md, pw);
    return fps_pick_md<PPT, MID, HI>(md, pw);
  }
}

// x / y / z / md of slot pw for every lane in one walk of the tree
template <int PPT, int LO, int HI>
__device__ __forceinline__ void fps_pick_slot(const float (&x)[PPT], const float (&y)[PPT],
                                              const float (&z)[PPT], const float (&md)[PPT], int pw,
                                              float& vx, float& vy, float& vz, float& vm) {
  if constexpr (HI - LO == 1) {
    vx = x[LO];
    vy = y[LO];
    vz = z[LO];
    vm = md[LO];
    asm volatile("; fps pick" : "+v"(vx), "+v"(vy), "+v"(vz), "+v"(vm));
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (pw < MID)
      fps_pick_slot<PPT, LO, MID>(x, y, z, md, pw, vx, vy, vz, vm);
    else
      fps_pick_slot<PPT, MID, HI>(x, y, z, md, pw, vx, vy, vz, vm);
  }
}

// update of one group (slot P of this wave): distances, running minimum, new group maximum
template <int PPT, int LO, int HI, bool FMAD>
__device__ __forceinline__ void fps_update_slot(const float (&x)[PPT], const float (&y)[PPT],
                                                const float (&z)[PPT], float (&md)[PPT], int p,
                                                float cx, float cy, float cz, float& gmax) {
  if constexpr (HI - LO == 1) {
    // volatile: keeps this leaf behind its (wave-uniform) branch -- without it the
    // compiler if-converts the tree and evaluates every slot's distance each step
    // (the distance is computed from the asm's output, so it cannot be hoisted either)
    float xl = x[LO];
    asm volatile("; fps slot %1" : "+v"(xl) : "n"(LO));
    const float d = dist2<FMAD>(cx, cy, cz, xl, y[LO], z[LO]);
    float m;
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(d), "v"(md[LO]));
    md[LO] = m;
    const uint32_t g = wave_max_u32(__float_as_uint(m));   // m >= 0: the bit pattern orders like the value
    gmax = __uint_as_float(g);
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (p < MID)
      fps_update_slot<PPT, LO, MID, FMAD>(x, y, z, md, p, cx, cy, cz, gmax);
    else
      fps_update_slot<PPT, MID, HI, FMAD>(x, y, z, md, p, cx, cy, cz, gmax);
  }
}

// MAXP = picks one exchange may settle (1: the plain exchange; 4 by default, see fps_block_exchange_multi)
template <int THREADS, int PPT, bool FMAD, typename IdxT, int MAXP>
__global__ __launch_bounds__(THREADS) void fps_pruned_kernel(const float* __restrict__ xyz,
                                                             const int* __restrict__ perm,
                                                             const float* __restrict__ gbox,
                                                             const float* __restrict__ md_in, int i0,
                                                             int N, int M, IdxT* __restrict__ idx,
                                                             float* __restrict__ ctr, int lg_bs,
                                                             float* __restrict__ dist) {
  constexpr int WAVES = THREADS / 64;
  constexpr int GPL = (PPT + 63) / 64;   // group registers per lane: slot p lives in lane p % 64, reg p / 64
  constexpr bool SPEC = MAXP > 1;
  __shared__ FpsSlot slots[2][FPS_MAX_WAVES];
  extern __shared__ uint16_t orig[];   // [WAVES * PPT * 64] sorted position -> original index
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const float* __restrict__ px = xyz + (size_t)b * 3 * N;
  const float* __restrict__ py = px + N;
  const float* __restrict__ pz = py + N;
  const int* __restrict__ pb = perm + (size_t)b * N;
  IdxT* __restrict__ out = idx + (size_t)b * M;
  float* __restrict__ cout = ctr ? ctr + (size_t)b * 3 * M : nullptr;
  float* __restrict__ dout = dist ? dist + (size_t)b * M : nullptr;   // every pick's min-distance (optional)
  const uint32_t bs_mask = (1u << lg_bs) - 1u;
  auto tie_key = [&](uint32_t j) { return ((__brev(j & bs_mask) >> (32 - lg_bs)) << 23) | j; };

  // slot p of this lane = sorted position 64 * (WAVES * p + wave) + lane
  float x[PPT], y[PPT], z[PPT], md[PPT];
  float blx[GPL], bly[GPL], blz[GPL], bhx[GPL], bhy[GPL], bhz[GPL];   // boxes of this lane's groups
  float mg[GPL];                                                      // their max min-distances
  constexpr int G = WAVES * PPT;           // groups of this scene (the pre-pass's G)
#pragma unroll
  for (int r = 0; r < GPL; ++r) {
    const int slot = 64 * r + lane;          // group slot of this lane: group WAVES * slot + wave
    const float* gb = gbox + ((size_t)b * G + (slot < PPT ? WAVES * slot + wave : 0)) * 6;
    blx[r] = gb[0]; bly[r] = gb[1]; blz[r] = gb[2];
    bhx[r] = gb[3]; bhy[r] = gb[4]; bhz[r] = gb[5];
    mg[r] = 0.f;
  }
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int s = 64 * (WAVES * p + wave) + lane;
    const bool ok = s < N;
    const int j = ok ? pb[s] : 0;
    x[p] = px[j];
    y[p] = py[j];
    z[p] = pz[j];
    // min-distances after the first i0 steps (full-scan kernel); padding lanes sit at 0:
    // never a maximum unless every real point is at 0 too
    md[p] = ok ? (md_in ? md_in[(size_t)b * N + j] : __builtin_inff()) : 0.0f;
    orig[s] = (uint16_t)j;
  }
  const uint32_t rkey = (__brev((uint32_t)t & bs_mask) >> (32 - lg_bs)) << 23;   // all-zero case only
  __syncthreads();

  // md_in == NULL: no full-scan kernel ran before (i0 == 1): point 0 is the first centroid and every
  // min-distance starts at +inf, so the first update touches every group
  int cur = md_in ? (int)out[i0 - 1] : 0;
  float cx = px[cur], cy = py[cur], cz = pz[cur];
  if (!md_in && t == 0) {
    out[0] = 0;
    if (cout) {
      cout[0] = cx;
      cout[M] = cy;
      cout[2 * M] = cz;
    }
    if (dout) dout[0] = __builtin_inff();
  }

  auto publish = [&](int i, uint32_t wmax, uint32_t wtie, float sx, float sy, float sz) {
    uint32_t dwin;
    fps_block_exchange<WAVES>(slots[i & 1], wave, lane, wmax, wtie, sx, sy, sz, cur, cx, cy, cz, &dwin);
    if (t == 0) {
      out[i] = (IdxT)cur;
      if (cout) {
        cout[i] = cx;
        cout[M + i] = cy;
        cout[2 * M + i] = cz;
      }
      if (dout) dout[i] = __uint_as_float(dwin);
    }
  };

  // exact group maxima, once
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const uint32_t g = wave_max_u32(__float_as_uint(md[p]));
    if (lane == (p & 63)) mg[p >> 6] = __uint_as_float(g);
  }

  // ---- pruned phase ---------------------------------------------------------
  // MAXP > 1: an exchange may settle several picks (fps_block_exchange_multi); all their centroids
  // are then applied before the next candidates are taken (the updates commute: each is a running
  // minimum).
  int npend = 1;                       // centroids whose update is still due
  int xpar = 0;                        // LDS exchange buffer of the next exchange
  uint32_t pwaves = 0u;                // their entries in the previous exchange's buffer, 4 bits each
  [[maybe_unused]] float f0x, f0y, f0z;  // (first pick of an exchange: unused here)
  if constexpr (SPEC) {                // the hand-over centroid poses as entry 0 of "the previous exchange"
    if (t == 0) {
      FpsSlot s0;
      s0.d = s0.tie = s0.d2 = s0.key2 = s0.pad = 0u;
      s0.x = cx;
      s0.y = cy;
      s0.z = cz;
      slots[1][0] = s0;
    }
    __syncthreads();
  }
  S4G_FPS_ACC_DECL();
  for (int i = i0; i < M;) {
    uint32_t wmax, wtie, wd2 = 0u, wkey2 = 0u;   // wkey2: the runner-up's tie key where it ties with the candidate
    float sx = cx, sy = cy, sz = cz;
    [[maybe_unused]] const unsigned long long st0 = S4G_FPS_T();
    // 1. groups the new centroid(s) can still change
    uint32_t gbits[GPL];
#pragma unroll 1
    for (int q = 0; q < npend; ++q) {
    float ux = cx, uy = cy, uz = cz;
    if constexpr (SPEC) {   // pick q of the last exchange: its coordinates are still in that buffer
      const FpsSlot& e = slots[xpar ^ 1][(pwaves >> (4 * q)) & 15u];
      ux = e.x;
      uy = e.y;
      uz = e.z;
    }
#pragma unroll
    for (int r = 0; r < GPL; ++r) {
      const bool live = 64 * r + lane < PPT;
      // per axis max(lo - u, u - hi, 0): the distance to the box, one v_max3 instead of two compares + two
      // selects (u - hi where the select form had hi - u: the square is the same bit for bit)
      const float tx = fps_max3(__fsub_rn(blx[r], ux), __fsub_rn(ux, bhx[r]), 0.f);
      const float ty = fps_max3(__fsub_rn(bly[r], uy), __fsub_rn(uy, bhy[r]), 0.f);
      const float tz = fps_max3(__fsub_rn(blz[r], uz), __fsub_rn(uz, bhz[r]), 0.f);
      float lb;
      if constexpr (FMAD) {
        lb = __fmaf_rn(tz, tz, __fmaf_rn(ty, ty, __fmul_rn(tx, tx)));
      } else {
        lb = __fadd_rn(__fadd_rn(__fmul_rn(tx, tx), __fmul_rn(ty, ty)), __fmul_rn(tz, tz));
      }
      const uint64_t need = __ballot(live && lb < mg[r]);
      S4G_FPS_ACC(5, __popcll(need));
      // straight-line dispatch: one (rarely taken) scalar branch per 8 slots, then one per
      // slot; every leaf indexes its registers statically.  The volatile asm pins a leaf's
      // arithmetic behind its branch (the compiler would otherwise evaluate all of them).
      if (need) {
#pragma unroll
        for (int p8 = 0; p8 < 64 && 64 * r + p8 < PPT; p8 += 8) {
          if (__builtin_expect(((need >> p8) & 0xFFull) != 0ull, 0)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              constexpr int dummy = 0;
              (void)dummy;
              const int p = 64 * r + p8 + e;
              if (p < PPT) {
                if ((need >> (p8 + e)) & 1ull) {
                  float xl = x[p];
                  asm volatile("; fps slot" : "+v"(xl));
                  const float d = dist2<FMAD>(ux, uy, uz, xl, y[p], z[p]);
                  float m;
                  const float old = md[p];
                  asm volatile("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(d), "v"(old));
                  md[p] = m;
                  // the group maximum only moves if a point that held it came closer
                  const uint32_t gm = __builtin_amdgcn_readlane(__float_as_uint(mg[r]), p8 + e);
                  if (__ballot(__float_as_uint(old) == gm && m < old)) {
                    const uint32_t g = wave_max_u32(__float_as_uint(m));
                    if (lane == p8 + e) mg[r] = __uint_as_float(g);
                  }
                }
              }
            }
          }
        }
      }
      gbits[r] = live ? __float_as_uint(mg[r]) : 0u;
    }
    }
    [[maybe_unused]] const unsigned long long st1 = S4G_FPS_T();
    // 2. this wave's candidate: the group(s) holding the largest maximum
    uint32_t lmax = gbits[0];
#pragma unroll
    for (int r = 1; r < GPL; ++r) lmax = max(lmax, gbits[r]);
    wmax = wave_max_u32(lmax);
    if (wmax == 0u) {
      wtie = wave_min_u32(rkey | (uint32_t)cur);   // every point of this wave is at distance 0
    } else {
      uint32_t best_key = 0xFFFFFFFFu, second_key = 0xFFFFFFFFu;   // smallest / second-smallest key among the holders
      int nbest = 0;                               // groups / lanes that hold the maximum
      // Tie-heavy clouds (coordinates on a lattice: many groups of a wave hold the SAME maximum): the
      // group-by-group walk below costs ~80 instructions per tying group -- 42 ms instead of 4 for a batch
      // of lattice scenes of 25 600 points (round 4's `mixed_batch` leg found it).  From FPS_TIE_PAR tying
      // groups on, every lane looks through its OWN tying slots instead: 21 ms; with the runner-up's key published
      // to the exchange (wkey2: ties no longer end an exchange after one pick) 12.4 ms.
      bool tie_par = false;
      if constexpr (GPL == 1) {
        const uint64_t gtie = __ballot(gbits[0] == wmax);     // bit = slot (group) of this wave that holds the maximum
        tie_par = __popcll(gtie) > FPS_TIE_PAR;
        if (__builtin_expect(tie_par, 0)) {   // (unlikely: laid out behind the loop -- the step's code is as large as the instruction cache)
          // (through LDS: this thread's min-distances into its own column, then a ROLLED loop over them --
          // unrolled over the registers the block costs the common path 13 spilled registers and 4 %)
          float* __restrict__ mdl = reinterpret_cast<float*>(orig + THREADS * PPT) + t;
#pragma unroll
          for (int pp = 0; pp < PPT; ++pp)
            if ((gtie >> pp) & 1ull) mdl[pp * THREADS] = md[pp];      // (wave-uniform: only the tying slots)
          uint32_t kl = 0xFFFFFFFFu, k2l = 0xFFFFFFFFu;      // this lane's smallest and second-smallest key
          int pl = 0;
          uint64_t gleft = gtie;
#pragma unroll 1
          while (gleft) {
            const int pp = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)gleft) - 1);
            gleft &= gleft - 1;
            const int s = 64 * (WAVES * pp + wave) + lane;
            const uint32_t kk = tie_key(orig[s < N ? s : 0]);
            const bool hit = s < N && __float_as_uint(mdl[pp * THREADS]) == wmax;
            if (hit && kk < kl) {
              k2l = kl;
              kl = kk;
              pl = pp;
            } else if (hit && kk < k2l) {
              k2l = kk;
            }
          }
          const uint32_t kmin = wave_min_u32(kl);
          const int wl = __ffsll((unsigned long long)__ballot(kl == kmin)) - 1;
          wkey2 = wave_min_u32(lane == wl ? k2l : kl);         // the wave's second holder by key
          const int pw = __builtin_amdgcn_readlane(pl, wl);
          float vx, vy, vz, vm;
          fps_pick_slot<PPT, 0, PPT>(x, y, z, md, pw, vx, vy, vz, vm);
          sx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vx), wl));
          sy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vy), wl));
          sz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vz), wl));
          best_key = kmin;
          nbest = 2;                               // several holders of the maximum: no second pick is provable
        }
      }
#pragma unroll
      for (int r = 0; r < GPL; ++r) {
        uint64_t gmask = tie_par ? 0ull : __ballot(gbits[r] == wmax);
        if constexpr (SPEC) nbest += __popcll(gmask);
        const int ngroups = __popcll(gmask);            // (copies of a group's winner only die if it is the wave's winner)
        while (gmask) {                          // one group unless maxima tie across groups
          const int gl = __ffsll((unsigned long long)gmask) - 1;
          // (readfirstlane: the slot number IS wave-uniform, but only a value the compiler knows to be
          // uniform turns the walk to the slot into scalar compares instead of 64-bit vector compares)
          const int pw = __builtin_amdgcn_readfirstlane(64 * r + gl);
          gmask &= gmask - 1;
          const int s = 64 * (WAVES * pw + wave) + lane;
          // every lane reads its original index early (the LDS read overlaps the tree walk)
          const uint32_t kk = tie_key(orig[s < N ? s : 0]);
          float vx, vy, vz, vm;
          fps_pick_slot<PPT, 0, PPT>(x, y, z, md, pw, vx, vy, vz, vm);
          const bool hit = s < N && __float_as_uint(vm) == wmax;
          const uint64_t eq = __ballot(hit);
          const uint32_t k = hit ? kk : 0xFFFFFFFFu;
          uint32_t kmin;
          int wl;
          if (__popcll(eq) == 1) {
            wl = __ffsll((unsigned long long)eq) - 1;
            kmin = __builtin_amdgcn_readlane(k, wl);
          } else {
            kmin = wave_min_u32(k);
            wl = __ffsll((unsigned long long)__ballot(k == kmin)) - 1;
          }
          if constexpr (SPEC) {
            // the wave's runner-up distance: the best of this group's other points and of the
            // other groups' maxima (a second holder of the maximum makes it the maximum itself)
            // (round 4) Several holders of the maximum in the group that are EXACT COPIES of the winner -- clouds
            // subsampled with replacement, as the reference's own harness does (grasp_proposal_test.py:29) -- fall
            // to zero with it: they do not bound the next pick, the runner-up is the best point that is not a copy.
            bool dup = lane == wl;
            if (__builtin_expect(__popcll(eq) > 1 || ngroups > 1, 0)) {   // ties: everything below stays off the tie-free path
              uint32_t k2g = 0xFFFFFFFFu;                       // this group's second holder by key
              if (__popcll(eq) > 1) {
                if (ngroups == 1) {
                  const float wx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vx), wl));
                  const float wy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vy), wl));
                  const float wz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vz), wl));
                  dup = hit && vx == wx && vy == wy && vz == wz;
                }
                if (__ballot(hit && !dup)) {                    // a different point at the same distance
                  nbest = 2;
                  k2g = wave_min_u32((hit && !dup) ? k : 0xFFFFFFFFu);
                }
              }
              if (nbest > 1) {
                if (kmin < best_key) {
                  second_key = min(min(second_key, best_key), k2g);
                } else {
                  second_key = min(second_key, kmin);
                }
              }
            }
            if (nbest == 1) {
              // max(this group's other points, the other groups' maxima): one wave reduction for both
              uint32_t others = (s < N && !dup) ? __float_as_uint(vm) : 0u;
#pragma unroll
              for (int r2 = 0; r2 < GPL; ++r2)
                others = max(others, (r2 == r && lane == gl) ? 0u : gbits[r2]);
              wd2 = wave_max_u32(others);
            }
          }
          if (kmin < best_key) {
            best_key = kmin;
            sx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vx), wl));
            sy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vy), wl));
            sz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vz), wl));
          }
        }
      }
      if constexpr (SPEC) {
        if (__builtin_expect(nbest > 1, 0)) {
          wd2 = wmax;
          if (!tie_par) wkey2 = second_key == 0xFFFFFFFFu ? 0u : second_key;
        }
      }
      wtie = best_key;
    }
    [[maybe_unused]] const unsigned long long st2 = S4G_FPS_T();
    S4G_FPS_ACC(0, st1 - st0);
    S4G_FPS_ACC(1, st2 - st1);
    if constexpr (SPEC) {
      // buffer parity per EXCHANGE, not per step: a multiple pick advances i by more than one, and
      // slots[i & 1] could then be rewritten by a fast wave while a slow one still reads the previous
      // exchange
      npend = fps_block_exchange_multi<WAVES, FMAD, MAXP, IdxT>(slots[xpar], wave, lane, wmax, wtie, wd2, wkey2, sx, sy,
                                                                sz, M - i, cur, cx, cy, cz, pwaves, out + i,
                                                                cout ? cout + i : nullptr, M, f0x, f0y, f0z,
                                                                dout ? dout + i : nullptr);
      xpar ^= 1;
      i += npend;
    } else {
      publish(i, wmax, wtie, sx, sy, sz);
      ++i;
    }
    S4G_FPS_ACC(2, S4G_FPS_T() - st2);
    S4G_FPS_ACC(3, 1);
    S4G_FPS_ACC(4, npend);
  }
  S4G_FPS_ACC_FLUSH();
}


// ---------------------------------------------------------------------------
// Pruned FPS for clouds whose COORDINATES do not fit one CU's registers (25 600 < N <= 51 200).
// Same bookkeeping as fps_pruned_kernel -- Morton groups of 64 points, a box and an exact
// maximum per group, a step only rescans the groups the new centroid can still change, two picks
// per exchange where the runner-up is out of reach -- but only the running min-distances (100
// per lane) stay resident.  The coordinates of a touched group are ONE coalesced 1 KB read of
// the Morton-sorted (x, y, z, original index) records (L2-resident: 0.8 MB per scene), issued
// for a block of eight slots at a time; the candidate's record is read the same way.  A step
// costs two dependent L2 round trips more than the register-resident kernel, but it needs ONE CU
// per scene where the full-scan cluster kernel holds two (3.1 us/step each): a 32-scene batch
// of 51 200-point clouds took 30 % of the chip's CU time for FPS alone.  No dense first phase:
// the min-distances start at +inf, so the first steps touch every group (13 blocks of loads per
// step) and the count decays within a few dozen steps.
// ---------------------------------------------------------------------------

template <int THREADS, int PPT, bool FMAD, typename IdxT, int MAXP>
__global__ __launch_bounds__(THREADS) void fps_pruned_l2_kernel(const float* __restrict__ xyz,
                                                                const float4* __restrict__ sorted,
                                                                const float* __restrict__ gbox, int N, int M,
                                                                IdxT* __restrict__ idx, float* __restrict__ ctr,
                                                                int lg_bs, float* __restrict__ dist) {
  constexpr int WAVES = THREADS / 64;
  constexpr int GPL = (PPT + 63) / 64;
  constexpr int G = WAVES * PPT;
  __shared__ FpsSlot slots[2][FPS_MAX_WAVES];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const float* __restrict__ px = xyz + (size_t)b * 3 * N;
  const float4* __restrict__ srt = sorted + (size_t)b * THREADS * PPT;
  IdxT* __restrict__ out = idx + (size_t)b * M;
  float* __restrict__ cout = ctr ? ctr + (size_t)b * 3 * M : nullptr;
  float* __restrict__ dout = dist ? dist + (size_t)b * M : nullptr;
  const uint32_t bs_mask = (1u << lg_bs) - 1u;
  auto tie_key = [&](uint32_t j) { return ((__brev(j & bs_mask) >> (32 - lg_bs)) << 23) | j; };
  // slot p of this lane = sorted position 64 * (WAVES * p + wave) + lane; its record through a
  // buffer load: one lane offset for every slot, the slot's offset is a scalar
  auto spos = [&](int p) { return 64 * (WAVES * p + wave) + lane; };
  const __amdgpu_buffer_rsrc_t srt_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)srt, 0, THREADS * PPT * 16, 0x00020000);
  const int lane_off = (64 * wave + lane) * 16;
  auto record = [&](int p) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srt_rsrc, lane_off, p * THREADS * 16, 0));
  };

  float md[PPT];
  float blx[GPL], bly[GPL], blz[GPL], bhx[GPL], bhy[GPL], bhz[GPL], mg[GPL];
#pragma unroll
  for (int r = 0; r < GPL; ++r) {
    const int slot = 64 * r + lane;
    const int g = slot < PPT ? WAVES * slot + wave : 0;
    const float* gb = gbox + ((size_t)b * G + g) * 6;
    blx[r] = gb[0]; bly[r] = gb[1]; blz[r] = gb[2];
    bhx[r] = gb[3]; bhy[r] = gb[4]; bhz[r] = gb[5];
    mg[r] = (slot < PPT && 64 * g < N) ? __builtin_inff() : 0.f;   // groups of padding only never lead
  }
#pragma unroll
  for (int p = 0; p < PPT; ++p) md[p] = spos(p) < N ? __builtin_inff() : 0.0f;
  const uint32_t rkey = (__brev((uint32_t)t & bs_mask) >> (32 - lg_bs)) << 23;   // all-zero case only

  int cur = 0;
  float cx = px[0], cy = px[(size_t)N], cz = px[2 * (size_t)N];
  if (t == 0) {
    out[0] = 0;
    if (cout) {
      cout[0] = cx;
      cout[M] = cy;
      cout[2 * M] = cz;
    }
    if (dout) dout[0] = __builtin_inff();
  }

  // (two picks per exchange here: with a touched group costing an L2 round trip the update phase
  // dominates and more picks per exchange only add imbalance -- 9.7 ms against 9.2 for 32 scenes)
  static_assert(MAXP == 2, "the pending pair lives in registers");
  int npend = 1;
  int xpar = 0;
  uint32_t pwaves = 0u;
  float fx = cx, fy = cy, fz = cz;     // first pick of the last exchange; (cx, cy, cz) = its last one
  for (int i = 1; i < M;) {
    uint32_t wmax, wtie, wd2 = 0u, wkey2 = 0u;
    float sx = cx, sy = cy, sz = cz;
    uint32_t gbits[GPL];
#pragma unroll 1
    for (int q = 0; q < npend; ++q) {
      const bool lastp = q == npend - 1;
      const float ux = lastp ? cx : fx, uy = lastp ? cy : fy, uz = lastp ? cz : fz;
#pragma unroll
      for (int r = 0; r < GPL; ++r) {
        const bool live = 64 * r + lane < PPT;
        // per axis max(lo - u, u - hi, 0): the distance to the box, one v_max3 instead of two compares + two
        // selects (u - hi where the select form had hi - u: the square is the same bit for bit)
        const float tx = fps_max3(__fsub_rn(blx[r], ux), __fsub_rn(ux, bhx[r]), 0.f);
        const float ty = fps_max3(__fsub_rn(bly[r], uy), __fsub_rn(uy, bhy[r]), 0.f);
        const float tz = fps_max3(__fsub_rn(blz[r], uz), __fsub_rn(uz, bhz[r]), 0.f);
        float lb;
        if constexpr (FMAD) {
          lb = __fmaf_rn(tz, tz, __fmaf_rn(ty, ty, __fmul_rn(tx, tx)));
        } else {
          lb = __fadd_rn(__fadd_rn(__fmul_rn(tx, tx), __fmul_rn(ty, ty)), __fmul_rn(tz, tz));
        }
        const uint64_t need = __ballot(live && lb < mg[r]);
        if (need) {
#pragma unroll
          for (int p8 = 0; p8 < 64 && 64 * r + p8 < PPT; p8 += 8) {
            if (__builtin_expect(((need >> p8) & 0xFFull) != 0ull, 0)) {
              // the touched slots' records first (all in flight together), then the updates
              float4 rec[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int p = 64 * r + p8 + e;
                if (p < PPT) {
                  if ((need >> (p8 + e)) & 1ull) rec[e] = record(p);
                }
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int p = 64 * r + p8 + e;
                if (p < PPT) {
                  if ((need >> (p8 + e)) & 1ull) {
                    const float d = dist2<FMAD>(ux, uy, uz, rec[e].x, rec[e].y, rec[e].z);
                    float m;
                    const float old = md[p];
                    asm volatile("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(d), "v"(old));
                    md[p] = m;
                    // the group maximum only moves if a point that held it came closer
                    const uint32_t gm = __builtin_amdgcn_readlane(__float_as_uint(mg[r]), p8 + e);
                    if (__ballot(__float_as_uint(old) == gm && m < old)) {
                      const uint32_t g = wave_max_u32(__float_as_uint(m));
                      if (lane == p8 + e) mg[r] = __uint_as_float(g);
                    }
                  }
                }
              }
            }
          }
        }
        gbits[r] = live ? __float_as_uint(mg[r]) : 0u;
      }
    }
    // this wave's candidate and the distance of its runner-up
    uint32_t lmax = gbits[0];
#pragma unroll
    for (int r = 1; r < GPL; ++r) lmax = max(lmax, gbits[r]);
    wmax = wave_max_u32(lmax);
    if (wmax == 0u) {
      wtie = wave_min_u32(rkey | (uint32_t)cur);   // every point of this wave is at distance 0
    } else {
      uint32_t best_key = 0xFFFFFFFFu, second_key = 0xFFFFFFFFu;   // (second key: see fps_pruned_kernel)
      int nbest = 0;
      int ngroups = 0;
#pragma unroll
      for (int r = 0; r < GPL; ++r) ngroups += __popcll(__ballot(gbits[r] == wmax));
#pragma unroll
      for (int r = 0; r < GPL; ++r) {
        uint64_t gmask = __ballot(gbits[r] == wmax);
        nbest += __popcll(gmask);
        while (gmask) {
          const int gl = __ffsll((unsigned long long)gmask) - 1;
          // (readfirstlane: the slot number IS wave-uniform, but only a value the compiler knows to be
          // uniform turns the walk to the slot into scalar compares instead of 64-bit vector compares)
          const int pw = __builtin_amdgcn_readfirstlane(64 * r + gl);
          gmask &= gmask - 1;
          const int s = 64 * (WAVES * pw + wave) + lane;
          const float4 me = record(pw);                         // this lane's point of the group (zeros past the end)
          const float vm = fps_pick_md<PPT, 0, PPT>(md, pw);
          const uint32_t kk = tie_key((uint32_t)__float_as_int(me.w));
          const bool hit = s < N && __float_as_uint(vm) == wmax;
          const uint64_t eq = __ballot(hit);
          const uint32_t k = hit ? kk : 0xFFFFFFFFu;
          uint32_t kmin;
          int wl;
          if (__popcll(eq) == 1) {
            wl = __ffsll((unsigned long long)eq) - 1;
            kmin = __builtin_amdgcn_readlane(k, wl);
          } else {
            kmin = wave_min_u32(k);
            wl = __ffsll((unsigned long long)__ballot(k == kmin)) - 1;
          }
          bool dup = lane == wl;                               // exact copies of the winner: see fps_pruned_kernel
          uint32_t k2g = 0xFFFFFFFFu;
          if (__builtin_expect(__popcll(eq) > 1, 0)) {
            if (ngroups == 1) {
              const float wx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.x), wl));
              const float wy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.y), wl));
              const float wz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.z), wl));
              dup = hit && me.x == wx && me.y == wy && me.z == wz;
            }
            if (__ballot(hit && !dup)) {
              nbest = 2;
              k2g = wave_min_u32((hit && !dup) ? k : 0xFFFFFFFFu);
            }
          }
          if (__builtin_expect(ngroups > 1 || nbest > 1, 0)) {
            if (kmin < best_key) {
              second_key = min(min(second_key, best_key), k2g);
            } else {
              second_key = min(second_key, kmin);
            }
          }
          if (nbest == 1) {
            uint32_t others = (s < N && !dup) ? __float_as_uint(vm) : 0u;
#pragma unroll
            for (int r2 = 0; r2 < GPL; ++r2) others = max(others, (r2 == r && lane == gl) ? 0u : gbits[r2]);
            wd2 = wave_max_u32(others);
          }
          if (kmin < best_key) {
            best_key = kmin;
            sx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.x), wl));
            sy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.y), wl));
            sz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.z), wl));
          }
        }
      }
      if (__builtin_expect(nbest > 1, 0)) {
        wd2 = wmax;
        wkey2 = second_key == 0xFFFFFFFFu ? 0u : second_key;
      }
      wtie = best_key;
    }
    npend = fps_block_exchange_multi<WAVES, FMAD, MAXP, IdxT>(slots[xpar], wave, lane, wmax, wtie, wd2, wkey2, sx, sy, sz,
                                                              M - i, cur, cx, cy, cz, pwaves, out + i,
                                                              cout ? cout + i : nullptr, M, fx, fy, fz,
                                                              dout ? dout + i : nullptr);
    xpar ^= 1;   // per exchange (a multiple pick advances i by more than one)
    i += npend;
  }
}

constexpr int FPS_L2_CAP = 512 * 100;   // points per scene of fps_pruned_l2_kernel<512, 100>
constexpr int FPS_L2_CAP_BIG = 65535;  // ... of <512, 128> (round 4: 51 200 < N <= 65 535; the pre-pass's 16-bit positions end there)
static int fps_l2_slots(int64_t N) { return N <= FPS_L2_CAP ? 100 : 128; }

struct FpsSortWs {
  float* gbox;      // [B][G][6] boxes of the 64-point groups
  float* md;        // [B][N] min-distances handed over by the full-scan kernel (S4G_FPS_DENSE_STEPS > 1)
  int* val_out;     // [B][N] the pre-pass's permutation
  float4* aos;      // cell-ordered (x, y, z, index) records: fps_pruned_l2_kernel only (N > 25 600)
  size_t total;
};

static FpsSortWs fps_sort_ws(void* base, int64_t B, int64_t N) {
  FpsSortWs w;
  const size_t n = (size_t)B * (size_t)N;
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return r;
  };
  // fps_pruned_l2_kernel's box pass always writes 8 * 100 groups per scene (fewer real groups for
  // 25 600 < N < 34 816): size for whichever is larger
  const int64_t l2_groups = 8 * (int64_t)fps_l2_slots(N);
  const int64_t gbox_groups = (N + 63) / 64 + 256 > l2_groups ? (N + 63) / 64 + 256 : l2_groups;
  w.gbox = (float*)take(sizeof(float) * 6 * B * gbox_groups);
  w.md = (float*)take(sizeof(float) * n);
  w.val_out = (int*)take(sizeof(int) * n);
  w.aos = N > (int64_t)512 * 50 ? (float4*)take(sizeof(float4) * (size_t)B * 512 * fps_l2_slots(N)) : nullptr;
  w.total = off;
  return w;
}

// 25 600 < N <= 51 200: the one-CU pruned kernel with coordinates in L2 (default),
// S4G_FPS_MODE=cluster the two-CU full scan (opt-in, B <= 128), =hybrid the one-CU full scan
static bool fps_use_pruned_l2(int64_t N, int64_t M) {
  if (N <= (int64_t)512 * 50 || N > FPS_L2_CAP_BIG || M < 64) return false;
  const char* e = s4g::knob("S4G_FPS_MODE");
  return !(e && (e[0] == 'c' || e[0] == 'h' || e[0] == 'd'));
}

// M < 0: "may the pruned kernel run for this N" (workspace sizing, which does not know M)
static bool fps_use_pruned(int64_t N, int64_t M) {
  // S4G_FPS_MODE=dense|pruned (read per call).  Default: pruned where a lane holds more
  // than 20 points (N > 10 240) -- below that the full scan is cheaper than the
  // bookkeeping (SA2 size, 5 120 -> 1 024: 0.94 ms dense vs 1.28 ms pruned) -- unless the chain is long:
  // from 2 048 picks on the pruned kernel wins from 10 points per lane (round 4, 16 scenes in step,
  // M = 5 120: N = 6 000 4.2 vs 7.3 ms, 8 192 4.3 vs 7.4, 10 240 4.5 vs 7.2)
  const char* e = s4g::knob("S4G_FPS_MODE");
  if (e && e[0] == 'd') return false;
  if (N > (int64_t)512 * 50) return false;
  if (e && e[0] == 'p') return N > 512 * 5;
  return N > 512 * 20 || (N > 512 * 10 && (M < 0 || M >= 2048));
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

// The two-CU kernel is OPT-IN (S4G_FPS_MODE=cluster) and only taken when both workgroups of every
// scene can be resident at once (2 B workgroups of 512 threads, one per CU, on 256 CUs): its two
// halves spin on each other, and HIP guarantees no co-residency beyond what fits the chip.  A
// partner that never answers sets the error word AND turns every later index of that scene into
// -1 (the caller sees an impossible index instead of a plausible wrong one).  Everything else in
// this size range that cannot take the pruned kernel (M < 64, workspace too small) runs the
// single-CU hybrid kernel, which has no such requirement.
#ifdef S4G_VARIANTS
static bool fps_use_cluster(int64_t B) {
  const char* e = s4g::knob("S4G_FPS_MODE");
  return e && e[0] == 'c' && 2 * B <= 256;
}
static size_t fps_cluster_ws_bytes(int64_t B) { return (size_t)B * 4 * sizeof(FpsXch) + 64; }
#else
static size_t fps_cluster_ws_bytes(int64_t) { return 0; }
#endif

// ---------------------------------------------------------------------------
// FPS of an FPS-ordered set is its own prefix.  Let c_0 .. c_{M1-1} be the picks of one FPS run in
// pick order, D_k the min-distance pick k had when it was taken.  A second FPS over THAT set (the
// next set-abstraction level: modules.py:80-83 samples the previous level's centroids) starts at c_0
// and at step k looks for the point of the set farthest from {c_0 .. c_{k-1}}: c_k was the farthest
// point of the whole cloud, the set is a subset that contains it, so c_k attains the maximum of the
// set too -- with the same value D_k, the same fp32 arithmetic.  If it is the ONLY point that attains
// it, the second run picks position k whatever the tie rule; by induction its output is 0, 1, 2, ...
// This kernel checks the "only": lane j keeps m = min_{i < k} d(c_i, c_j) (the running min-distance
// the second run would hold for position j) and the scene fails if some j > k reaches D_k (or D_k
// is not a positive finite number).  M2 - 1 steps per lane, no communication: 20 us where the
// sequential run takes 0.87 ms (5 120 -> 1 024).  A scene that fails is sampled by the real kernel
// (fps_reg_kernel's `run` flag), so the result is the reference's in every case; a check over the
// first M2 steps also covers every further level that samples a prefix of this one.
// ---------------------------------------------------------------------------
template <bool FMAD>
__global__ __launch_bounds__(256) void fps_prefix_check_kernel(const float* __restrict__ ctr,
                                                               const float* __restrict__ dist, int M1,
                                                               int M2, int* __restrict__ run) {
  __shared__ float4 step[256];   // (c_k, D_{k+1}) of 256 steps at a time
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const float* __restrict__ cx = ctr + (size_t)b * 3 * M1;
  const float* __restrict__ cy = cx + M1;
  const float* __restrict__ cz = cy + M1;
  const float* __restrict__ D = dist + (size_t)b * M1;
  const int jj = j < M1 ? j : 0;
  const float sx = cx[jj], sy = cy[jj], sz = cz[jj];
  float m = __builtin_inff();
  bool bad = false;
  for (int k0 = 0; k0 + 1 < M2; k0 += 256) {
    const int kk = k0 + threadIdx.x;
    __syncthreads();
    if (kk + 1 < M2) step[threadIdx.x] = make_float4(cx[kk], cy[kk], cz[kk], D[kk + 1]);
    __syncthreads();
    const int n = min(256, M2 - 1 - k0);
#pragma unroll 4
    for (int q = 0; q < n; ++q) {
      const float4 c = step[q];
      const float d = dist2<FMAD>(c.x, c.y, c.z, sx, sy, sz);   // (centroid, point): the kernels' order
      m = d < m ? d : m;
      const int k = k0 + q;
      // element k + 1's own running min-distance must BE the reported D_{k+1} (bitwise: the sampler that
      // produced D and this kernel then provably share one arithmetic), every later element stays below it
      bad |= !(c.w > 0.f) || !(c.w < __builtin_inff()) || (j > k + 1 && j < M1 && !(m < c.w)) ||
             (j == k + 1 && __float_as_uint(m) != __float_as_uint(c.w));
    }
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(&run[b], 1);
}

static int launch_fps_cell_sort(const float* xyz, int64_t B, int64_t N, int G, int* perm, float* gbox,
                                float4* aos, int aos_cap, hipStream_t stream) {
  constexpr size_t lds = sizeof(uint32_t) * (FPS_SORT_WORDS + 3 * 256 + 64 * 64 / 2 + 7 * (FPS_SORT_THREADS / 64) + 8);
  static LdsAttrCache lds_cache;
  if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(&fps_cell_sort_kernel), lds, lds_cache)) return rc;
  hipLaunchKernelGGL(fps_cell_sort_kernel, dim3((unsigned)B), dim3(FPS_SORT_THREADS), lds, stream, xyz, (int)N, G,
                     perm, gbox, aos, aos_cap);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

template <bool FMAD, typename IdxT>
static int launch_fps(const float* xyz, int64_t B, int64_t N, int64_t M,
                      IdxT* idx, float* ctr, void* ws, size_t ws_bytes,
                      hipStream_t stream, FpsExtra ex = FpsExtra{nullptr, nullptr}) {
  const int lg = ref_block_lg(N);
  const dim3 grid((unsigned)B);

  // opt-in pruned variant: Morton order + group boxes, the first FPS_DENSE_STEPS steps by
  // the full-scan kernel (which leaves its min-distances in the workspace), the rest pruned
  FpsSortWs w = {};
  bool pruned = false;
  if (fps_use_pruned(N, M) && M > FPS_DENSE_STEPS && B < (1 << 16)) {
    w = fps_sort_ws(ws, B, N);
    pruned = ws && ws_bytes >= w.total;
  }
  int m_run = (int)M;
  float* md_out = nullptr;
  // steps the full-scan kernel runs in front of the pruned one (S4G_FPS_DENSE_STEPS; 0 = none: the pruned
  // kernel starts from +inf min-distances by itself)
  int dense_steps = 0;
  if (const char* e = s4g::knob("S4G_FPS_DENSE_STEPS")) dense_steps = atoi(e) > 1 ? atoi(e) : 0;
  if (dense_steps >= M) pruned = false;
  if (pruned) {
    const int G = 8 * (N <= 512 * 10 ? 10 : N <= 512 * 20 ? 20 : N <= 512 * 32 ? 32 : 50);   // groups of the pruned launch below
    if (int rc = launch_fps_cell_sort(xyz, B, N, G, w.val_out, w.gbox, nullptr, 0, stream)) return rc;
    m_run = dense_steps;
    md_out = w.md;
  }

  bool launched = pruned && dense_steps == 0;   // (no full-scan launch in front of the pruned kernel)
  // the per-scene skip flags are honoured by the full-scan kernel when it IS the sampler (N <= 10 240);
  // as the dense front of the pruned kernel it must run every scene, or the pruned steps would start
  // from min-distances nobody wrote
  const FpsExtra ex_reg = pruned ? FpsExtra{ex.dist, nullptr} : ex;
#define S4G_FPS_CASE(T, P)                                                   \
  if (!launched && N <= (int64_t)T * P) {                                    \
    hipLaunchKernelGGL((fps_reg_kernel<T, P, FMAD, IdxT>), grid, dim3(T), 0, \
                       stream, xyz, (int)N, (int)M, idx, ctr, lg, m_run,     \
                       md_out, ex_reg);                                      \
    S4G_LAUNCH_CHECK();                                                      \
    launched = true;                                                         \
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
  if (launched && pruned) {
  // up to four picks per exchange (one and two -- rounds 1 and 2 -- measured slower: 6.7 / 6.0 / 4.0 ms)
#define S4G_FPS_PRUNED_LAUNCH(T, P, S)                                                             \
  {                                                                                                \
    static LdsAttrCache lds_cache;                                                                 \
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(&fps_pruned_kernel<T, P, FMAD, IdxT, S>), lds, lds_cache)) return rc;  \
    hipLaunchKernelGGL((fps_pruned_kernel<T, P, FMAD, IdxT, S>), grid, dim3(T), lds, stream, xyz,  \
                       w.val_out, w.gbox, dense_steps ? w.md : nullptr, dense_steps ? dense_steps : 1, \
                       (int)N, (int)M, idx, ctr, lg, ex.dist);                                     \
  }
#define S4G_FPS_PRUNED(T, P)                                                                       \
  if (N <= (int64_t)T * P) {                                                                       \
    const size_t lds = (sizeof(uint16_t) + sizeof(float)) * T * P;   /* original indices + the tie path's min-distance columns */ \
    S4G_FPS_PRUNED_LAUNCH(T, P, 4)                                                                 \
    S4G_LAUNCH_CHECK();                                                                            \
    return S4G_OK;                                                                                 \
  }
    S4G_FPS_PRUNED(512, 10)
    S4G_FPS_PRUNED(512, 20)
    S4G_FPS_PRUNED(512, 32)
    S4G_FPS_PRUNED(512, 50)
#undef S4G_FPS_PRUNED
#undef S4G_FPS_PRUNED_LAUNCH
  }
  if (launched) return S4G_OK;
  if (fps_use_pruned_l2(N, M) && B < (1 << 16)) {
    const FpsSortWs w2 = fps_sort_ws(ws, B, N);
    if (ws && ws_bytes >= w2.total) {
      const int slots = fps_l2_slots(N);
      if (int rc = launch_fps_cell_sort(xyz, B, N, 8 * slots, w2.val_out, w2.gbox, w2.aos, 512 * slots, stream)) return rc;
      if (slots == 100)
        hipLaunchKernelGGL((fps_pruned_l2_kernel<512, 100, FMAD, IdxT, 2>), grid, dim3(512), 0, stream, xyz,
                           w2.aos, w2.gbox, (int)N, (int)M, idx, ctr, lg, ex.dist);
      else
        hipLaunchKernelGGL((fps_pruned_l2_kernel<512, 128, FMAD, IdxT, 2>), grid, dim3(512), 0, stream, xyz,
                           w2.aos, w2.gbox, (int)N, (int)M, idx, ctr, lg, ex.dist);
      S4G_LAUNCH_CHECK();
      return S4G_OK;
    }
  }
  if (ex.dist) return S4G_EUNSUPPORTED;   // the remaining kernels do not report pick distances
#ifdef S4G_VARIANTS
  // opt-in: two workgroups per scene, all points in registers, winners exchanged through L2 once
  // per step (see fps_use_cluster for the co-residency requirement)
  if (N <= (int64_t)512 * 100 && fps_use_cluster(B) && ws && ws_bytes >= fps_cluster_ws_bytes(B)) {
    FpsXch* xch = (FpsXch*)ws;
    int* err = (int*)((char*)ws + (size_t)B * 4 * sizeof(FpsXch));
    const hipError_t e = hipMemsetAsync(ws, 0, fps_cluster_ws_bytes(B), stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((fps_cluster_kernel<512, 50, FMAD, IdxT>), dim3((unsigned)(2 * B)), dim3(512), 0,
                       stream, xyz, (int)N, (int)M, idx, ctr, lg, xch, err);
    S4G_LAUNCH_CHECK();
    return S4G_OK;
  }
  if (N <= (int64_t)512 * 100) {   // x + min-distance in registers, y / z streamed from L2
    hipLaunchKernelGGL((fps_hybrid_kernel<512, 100, 10, FMAD, IdxT>), grid, dim3(512), 0, stream, xyz,
                       (int)N, (int)M, idx, ctr, lg);
    S4G_LAUNCH_CHECK();
    return S4G_OK;
  }
#endif
  // (default build: what the pruned kernel cannot take in this range -- M < 64, no workspace -- streams)
  if (ws_bytes < (size_t)B * (size_t)N * sizeof(float) || ws == nullptr)
    return S4G_EWORKSPACE;
  hipLaunchKernelGGL((fps_stream_kernel<FMAD, IdxT>), grid, dim3(FPS_THREADS), 0, stream,
                     xyz, (int)N, (int)M, idx, ctr, (float*)ws, lg);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

size_t fps_workspace_bytes(int64_t B, int64_t N) {
  if (N <= 0 || B <= 0) return 0;
  if (N <= (int64_t)512 * 50) return fps_use_pruned(N, -1) ? fps_sort_ws(nullptr, B, N).total : 0;
  if (N <= FPS_L2_CAP_BIG) {   // sort buffers + sorted records of the pruned kernel / the cluster kernel's exchange slots
    const size_t c = fps_cluster_ws_bytes(B);
    const size_t l2 = fps_sort_ws(nullptr, B, N).total;
    return l2 > c ? l2 : c;
  }
  return (size_t)B * (size_t)N * sizeof(float);
}

}  // namespace s4g

#ifdef S4G_FPS_STAMPS
extern "C" int s4g_debug_fps_stamps(unsigned long long* host_out_64x8x8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host_out_64x8x8, HIP_SYMBOL(s4g::g_fps_acc), sizeof(unsigned long long) * 64 * 8 * 8);
  if (e == hipSuccess && reset) {
    static unsigned long long zeros[64 * 8 * 8];
    e = hipMemcpyToSymbol(HIP_SYMBOL(s4g::g_fps_acc), zeros, sizeof(zeros));
  }
  return (int)e;
}
#endif

extern "C" int s4g_fps_gather_ex_i32(const float* xyz_b3n, int64_t B, int64_t N, int64_t M,
                                     int32_t* idx_bm, float* ctr_b3m, float* dist_bm,
                                     const int32_t* run_b, void* ws, size_t ws_bytes, int flags,
                                     s4g_stream_t stream) {
  if (B < 0 || M <= 0 || N < M || N >= (1 << 23)) return S4G_EINVAL;
  if (B == 0) return S4G_OK;
  if (!xyz_b3n || !idx_bm || !ctr_b3m) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const s4g::FpsExtra ex{dist_bm, run_b};
  if (flags & S4G_FLAG_FMAD)
    return s4g::launch_fps<true, int32_t>(xyz_b3n, B, N, M, idx_bm, ctr_b3m, ws, ws_bytes, st, ex);
  return s4g::launch_fps<false, int32_t>(xyz_b3n, B, N, M, idx_bm, ctr_b3m, ws, ws_bytes, st, ex);
}

extern "C" int s4g_fps_prefix_check_f32(const float* ctr_b3m, const float* dist_bm, int64_t B,
                                        int64_t M1, int64_t M2, int32_t* run_b, int flags,
                                        s4g_stream_t stream) {
  if (B < 0 || M1 <= 0 || M2 <= 0 || M2 > M1 || M1 >= (1 << 23) || B > 65535) return S4G_EINVAL;
  if (B == 0) return S4G_OK;
  if (!ctr_b3m || !dist_bm || !run_b) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(run_b, 0, sizeof(int32_t) * (size_t)B, st);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((unsigned)((M1 + 255) / 256), (unsigned)B);
  if (flags & S4G_FLAG_FMAD)
    hipLaunchKernelGGL(s4g::fps_prefix_check_kernel<true>, grid, dim3(256), 0, st, ctr_b3m, dist_bm, (int)M1,
                       (int)M2, run_b);
  else
    hipLaunchKernelGGL(s4g::fps_prefix_check_kernel<false>, grid, dim3(256), 0, st, ctr_b3m, dist_bm, (int)M1,
                       (int)M2, run_b);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_fps_prepass_f32(const float* xyz_b3n, int64_t B, int64_t N, int64_t G,
                                   int32_t* perm_bn, float* gbox_bg6, s4g_stream_t stream) {
  if (B < 0 || N <= 0 || N > 65535 || B > 65535 || G < (N + 63) / 64 || G > (1 << 20)) return S4G_EINVAL;
  if (B == 0) return S4G_OK;
  if (!xyz_b3n || !perm_bn || !gbox_bg6) return S4G_EINVAL;
  return s4g::launch_fps_cell_sort(xyz_b3n, B, N, (int)G, perm_bn, gbox_bg6, nullptr, 0, (hipStream_t)stream);
}

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
