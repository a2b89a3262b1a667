// The four per-point heads of the S4G network as ONE launch.
//
// Reference: PointNet2_tcls.py:83-95 (definitions) and :126-140 (forward): four SharedMLP stacks
// 256 -> 512 -> 256 -> 256 -> 128 (conv1x1 -> BN -> ReLU each, nn_utils/conv.py:24-34) that all
// read the same (B, 256, N) feature tensor, each followed by a Conv1d(128 -> c, bias) with
// c = 3 / 9 / 4 / 5 and a sigmoid on the last head.  36 % of the network's multiply-adds.
//
// Layer by layer this path wrote a (B N, 2048) tensor (heads.0), read it back, wrote (B N, 512)
// and read that for the logits: 5.7 GB of the 10.5 GB a 16-scene step moves, 33 GB of a
// 32 x 51 200-point step.  Here a workgroup owns 64 positions and walks ALL layers of ALL heads:
//   X (64 x 256, loaded, scaled and split once)                       -> LDS panel A, stays
//   per head g:  H0 = relu(W0 X)      512 wide, as two 256-wide halves -> LDS panel B
//                H1 = relu(W1 H0)     accumulated over the two halves  -> panel B
//                H2 = relu(W2 H1), H3 = relu(W3 H2) (128 wide)         -> panel B
//                logits = Wl H3 + b (sigmoid on head 3)                -> (B, c, N) tensors
// Nothing but X (once) and the 21 output channels crosses HBM.  Eight waves: wave w owns
// channels 32 w .. 32 w + 31 of a 256-wide layer for all 64 positions (2 accumulator blocks);
// W never touches LDS -- the host stores it in MFMA-fragment order and a wave streams its
// fragments through a two-deep register ring exactly like mlp_chain_kernel.  Operands are
// swapped (D = W A^T), so a lane ends up with 4 consecutive channels of ONE position: panel
// writes are 8-byte LDS stores and the logits leave as 128-byte rows of one channel.
//   PL = 2: fp32-class f16x2 arithmetic (two fp16 planes, three products, per-tile
//           power-of-two scales found by a wave-max + one barrier per layer);
//   PL = 1: one bf16 plane, one product, no scales (S4G_GEMM_BF16, configs[4]).
#include <stdlib.h>

#include "mlp_common.h"

// measurement builds only (make HIPFLAGS_EXTRA=-DS4G_HEADS_ABLATE=bits, tools/ablate_kernels.sh): the
// kernel with parts REMOVED, to see what each costs -- 1 no W refill, 2 no LDS operand reads in
// the strips, 4 no panel epilogue, 8 nor its barriers.  Results are garbage.
#ifndef S4G_HEADS_ABLATE
#define S4G_HEADS_ABLATE 0
#endif

namespace s4g {

__device__ __forceinline__ void heads_keep_alive(const f32x16& v) { asm volatile("" ::"v"(v)); }

struct HeadsParams {
  int P, N;                // positions (B * N), points per scene
  const float* X;          // (P, 256) channels-last fp32
  int ldx;
  const uint16_t* W[5];    // fragment-ordered planes: heads.0 (2048 x 256), heads.1 (4, 256, 512),
                           // heads.2 (4, 256, 256), heads.3 (4, 128, 256), logits (4, 32, 128)
  const float* bias[5];
  const float* wsc[5];     // per-channel inverse weight scales (PL == 2)
  float* out[4];
  int ch[4];
  long long obs[4];        // floats between two scenes' blocks of out[h] (ch[h] * N, or the packed tensor's C_total * N)
  int head_mask;           // bit h: evaluate head h (1 .. 15; the others' outputs are not touched)
  int sigmoid_head;
  const float* a_amax;     // per-scene maxima of X (PL == 2); PRE: of the sparse features
  float a_floor;
  int rps;
  // PRE: the last feature-propagation level's tail in front of the heads (see s4g_heads_desc_t):
  // panel A = relu(interp(sparse) (+ dense) + lbias), then two 256 -> 256 layers A -> B -> A
  const uint16_t* Wp[2];
  const float* bias_p[2];
  const float* wsc_p[2];
  const int* nidx;
  const float* nw;
  const float* sparse;
  const float* dense;
  const float* lbias;
  const float* a_amax2;
  int N2;
};

// NRBT = 32-position blocks per workgroup: 2 (64 positions; the f16x2 panels fill the LDS) or 4
// (128 positions, single-plane bf16 only: every W fragment then feeds four MFMAs instead of two,
// which halves the W bytes streamed per position -- the single product makes that stream three
// times as heavy per MFMA as in the f16x2 form).
template <int PL, int HD_RING, int NRBT, bool PRE>
__global__ __launch_bounds__(512, 2) void mlp_heads_kernel(const HeadsParams p) {
  constexpr int BM = 32 * NRBT, C = 256, NW = 8, astr = C + 8, aplane = BM * astr;
  constexpr int FB = PL * 1024;            // bytes of one (32 channels x 16 k) fragment block
  extern __shared__ __attribute__((aligned(16))) float smemf[];
  uint16_t* PA = reinterpret_cast<uint16_t*>(smemf);   // X       [PL][BM][C + 8]
  uint16_t* PB = PA + PL * aplane;                      // hidden  [PL][BM][C + 8]
  float* scr = reinterpret_cast<float*>(PB + PL * aplane);   // [NW][32 scale | 32 bias], then NW maxima

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int p0 = blockIdx.x * BM;
  const uint32_t wf_lane = (uint32_t)lane * 16u;

  // activation scale of X (per scene, see amax_rows)
  float sa = 1.f, inv_sa = 1.f;
  if constexpr (PL == 2) {
    float amax = p.a_floor;
    if constexpr (PRE) {   // the loader SUMS its inputs: |bias| + |interpolated| (+ |dense|)
      if (p.a_amax) amax += amax_rows(p.a_amax, lane, p0, min(p0 + BM, p.P) - 1, p.rps);
      if (p.a_amax2) amax += amax_rows(p.a_amax2, lane, p0, min(p0 + BM, p.P) - 1, p.rps);
    } else if (p.a_amax) {
      amax = fmaxf(amax, amax_rows(p.a_amax, lane, p0, min(p0 + BM, p.P) - 1, p.rps));
    }
    uint32_t ex = __float_as_uint(amax) >> 23;
    ex = ex < 15u ? 15u : (ex > 240u ? 240u : ex);
    ex = __builtin_amdgcn_readfirstlane(ex);
    sa = __uint_as_float((268u - ex) << 23);
    inv_sa = __uint_as_float((ex - 14u) << 23);
  }

  // fragment block (n32, ks) of layer L whose rows are nks blocks deep.  The W streams are read
  // with buffer loads: resource descriptor and block offset are scalars, the lane offset is one
  // constant VGPR -- no per-load 64-bit address arithmetic on the vector pipe.
  WRef wrs[5];
#pragma unroll
  for (int L = 0; L < 5; ++L) wrs[L] = wref(p.W[L], 0);
  auto woff = [&](int n32, int nks, int ks) { return (size_t)((n32 * nks + ks) * FB); };
  auto w0 = [&](int g, int h) { return wrs[0] + woff(g * 16 + h * 8 + wv, 16, 0); };
  auto w1 = [&](int g, int h) { return wrs[1] + woff(g * 8 + wv, 32, h * 16); };
  auto w2 = [&](int g) { return wrs[2] + woff(g * 8 + wv, 16, 0); };
  auto w3 = [&](int g) { return wrs[3] + woff(g * 4 + (wv & 3), 16, 0); };
  auto wl = [&](int g) { return wrs[4] + woff(g, 8, 0); };
  auto wload = [&](const WRef& w, uint32_t byte_off, int pl) { return wref_load(w, wf_lane + pl * 1024, byte_off); };
  // PRE: the two 256 -> 256 layers in front of the heads (this wave's 32 channels: block wv, 16 steps)
  const WRef wp0 = PRE ? wref(p.Wp[0], woff(wv, 16, 0)) : wrs[0];
  const WRef wp1 = PRE ? wref(p.Wp[1], woff(wv, 16, 0)) : wrs[0];

  // the heads this launch evaluates, in ascending order (head_mask: all four by default; a serving path may ask for the
  // score head alone on every point and for the pose heads on the points it keeps: FusedPointNet2(..., topk=))
  const int hmask = p.head_mask;
  const int g_first = __builtin_ctz(hmask);
  uint4 ring[HD_RING][PL];
  {
    const WRef w = PRE ? wp0 : w0(g_first, 0);
#pragma unroll
    for (int d = 0; d < HD_RING; ++d)
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) ring[d][pl] = wload(w, d * FB, pl);
  }

  // ---- prologue: X -> panel A (one round trip per 64 rows: 8 x 16 bytes per thread in flight)
#pragma unroll 1
  for (int r0 = 0; r0 < BM; r0 += 64) {
    const int row = r0 + (t >> 3), chunk = t & 7;
    const bool ok = p0 + row < p.P;
    float4 ra[8];
    if constexpr (PRE) {
      // X0 = relu(sum_k w_k S[idx_k] (+ dense) + bias): the three neighbour rows of the sparse level
      // (L2-resident: N2 x 256 floats per scene), 4 columns x 8 K-tiles per thread, same arithmetic
      // order as the chain kernel's loader and interp_add_cl_kernel
      const int pos = ok ? p0 + row : 0;
      const int bq = pos / p.N;
      int i3[3];
      float w3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        i3[k] = bq * p.N2 + p.nidx[(size_t)pos * 3 + k];
        w3[k] = p.nw[(size_t)pos * 3 + k];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // two half panels: 4 K-tiles x 3 (4) loads in flight
        float4 a[4], b[4], c[4], y[4], bb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int col = (h * 4 + kk) * 32 + chunk * 4;
          a[kk] = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[0] * 256 + col);
          b[kk] = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[1] * 256 + col);
          c[kk] = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[2] * 256 + col);
          bb[kk] = *reinterpret_cast<const float4*>(p.lbias + col);
          y[kk] = p.dense ? *reinterpret_cast<const float4*>(p.dense + (size_t)pos * 256 + col)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const float av[4] = {a[kk].x, a[kk].y, a[kk].z, a[kk].w}, bv[4] = {b[kk].x, b[kk].y, b[kk].z, b[kk].w};
          const float cv[4] = {c[kk].x, c[kk].y, c[kk].z, c[kk].w}, yv[4] = {y[kk].x, y[kk].y, y[kk].z, y[kk].w};
          const float bi[4] = {bb[kk].x, bb[kk].y, bb[kk].z, bb[kk].w};
          float r[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float acc = __fmul_rn(av[e], w3[0]);
            acc = __fadd_rn(acc, __fmul_rn(bv[e], w3[1]));
            acc = __fadd_rn(acc, __fmul_rn(cv[e], w3[2]));
            r[e] = ok ? fmaxf(__fadd_rn(__fadd_rn(yv[e], acc), bi[e]), 0.f) : 0.f;
          }
          ra[h * 4 + kk] = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    } else {
    const float* src = p.X + (size_t)(ok ? p0 + row : 0) * p.ldx + chunk * 4;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
      ra[kt] = ok ? nt_load4(src + kt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);   // read once: do not displace the W set in L2
    }
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      uint16_t* dst = PA + row * astr + kt * 32 + chunk * 4;
      if constexpr (PL == 2) {
        uint2 h, l;
        split2_h<false>(ra[kt], sa, h, l);
        *reinterpret_cast<uint2*>(dst) = h;
        *reinterpret_cast<uint2*>(dst + aplane) = l;
      } else {
        *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pk_bf16(ra[kt].x, ra[kt].y), cvt_pk_bf16(ra[kt].z, ra[kt].w));
      }
    }
  }
  __syncthreads();

  // NKS 16-deep steps of ACC[rb] += W(32 channels) . PANEL(rows ROWOFF + 32 rb ..)^T.  W fragments
  // come through the ring, refilled HD_RING steps ahead from this strip (WCUR) or the next one
  // (WNEXT); the first step starts from a literal zero accumulator (no register zeroing).
#define S4G_HD_TERM(NRB, PA_, PB_, ZERO)                                                            \
  _Pragma("unroll") for (int rb = 0; rb < NRB; ++rb)                                                \
    acc[rb] = chain_mfma<PL>(bf[PB_], af[rb][PA_], (ZERO) ? zero16 : acc[rb]);
#define S4G_HD_STRIP(PANEL, NRB, ROWOFF, WCUR, NKS, WNEXT)                                          \
  {                                                                                                 \
    const uint16_t* a_lane = (PANEL) + ((ROWOFF) + li) * astr + 8 * lh;                             \
    uint4 afn[NRBT][PL];                                                                            \
    _Pragma("unroll") for (int rb = 0; rb < NRB; ++rb) _Pragma("unroll") for (int pl = 0; pl < PL; ++pl) \
      afn[rb][pl] = *reinterpret_cast<const uint4*>(a_lane + pl * aplane + rb * 32 * astr);         \
    const WRef wcur_ = (WCUR);                                                                      \
    const WRef wnext_ = (WNEXT);                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                            \
      const int d = ks % HD_RING;                                                                   \
      const int ksn = ks + 1 == NKS ? 0 : ks + 1;                                                   \
      uint4 af[NRBT][PL], bf[PL];                                                                   \
      _Pragma("unroll") for (int rb = 0; rb < NRB; ++rb) _Pragma("unroll") for (int pl = 0; pl < PL; ++pl) { \
        af[rb][pl] = afn[rb][pl];                                                                   \
        if (!(S4G_HEADS_ABLATE & 2))                                                                \
          afn[rb][pl] = *reinterpret_cast<const uint4*>(a_lane + pl * aplane + rb * 32 * astr + ksn * 16); \
      }                                                                                             \
      _Pragma("unroll") for (int pl = 0; pl < PL; ++pl) bf[pl] = ring[d][pl];                       \
      {                                                                                             \
        const int kr = ks + HD_RING;                                                                \
        _Pragma("unroll") for (int pl = 0; pl < PL; ++pl)                                           \
          if (!(S4G_HEADS_ABLATE & 1))                                                              \
            ring[d][pl] = kr < NKS ? wload(wcur_, kr * FB, pl) : wload(wnext_, (kr - NKS) * FB, pl); \
      }                                                                                             \
      if constexpr (PL == 2) {                                                                      \
        S4G_HD_TERM(NRB, 0, 1, ks == 0)                                                             \
        S4G_HD_TERM(NRB, 1, 0, false)                                                               \
        S4G_HD_TERM(NRB, 0, 0, false)                                                               \
      } else {                                                                                      \
        S4G_HD_TERM(NRB, 0, 0, ks == 0)                                                             \
      }                                                                                             \
      _Pragma("unroll") for (int q = 0; q < PL; ++q) {                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                          \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                               \
  }

  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
  f32x16 acc[NRBT];
  float* epi_s = scr + wv * 64;

  // scale | bias of this wave's 32 channels, staged once per phase (read back as float4 per
  // register quad: lane holds channels 8 j + 4 lh + (0..3))
  auto stage_sb2 = [&](const float* wsc, const float* bias, int gch, float mul) {
    epi_s[lane] = lane < 32 ? (PL == 2 ? mul * wsc[gch + lane] : 1.f) : bias[gch + lane - 32];
  };
  auto stage_sb = [&](int L, int gch, float mul) { stage_sb2(p.wsc[L], p.bias[L], gch, mul); };

  // acc (NRB blocks of 32 positions x this wave's 32 channels) -> relu(acc * scale + bias) ->
  // tile maximum (one barrier: also the point after which nobody reads the old panel B) ->
  // split with the tile's own power-of-two scale -> panel B rows ROWOFF.., columns PCH0..
  float inv_sh = 1.f;
  auto panel_epilogue_to = [&](uint16_t* PD, auto nrb_tag, int rowoff, int pch0) {
    constexpr int NRB = decltype(nrb_tag)::value;
    if (S4G_HEADS_ABLATE & 4) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) heads_keep_alive(acc[rb]);
      if (!(S4G_HEADS_ABLATE & 8)) {
        __syncthreads();
        __syncthreads();
      }
      return;
    }
    float tmax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 sc4 = *reinterpret_cast<const float4*>(epi_s + 8 * j + 4 * lh);
      const float4 b4 = *reinterpret_cast<const float4*>(epi_s + 32 + 8 * j + 4 * lh);
      const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
      const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        // two v_pk_fma_f32 per four values (each half rounded once, as __fmaf_rn); the maximum as max3s (x >= 0)
        const f32x2 x01 = pk_fma(f32x2{acc[rb][4 * j], acc[rb][4 * j + 1]}, f32x2{scv[0], scv[1]}, f32x2{bv[0], bv[1]});
        const f32x2 x23 = pk_fma(f32x2{acc[rb][4 * j + 2], acc[rb][4 * j + 3]}, f32x2{scv[2], scv[3]}, f32x2{bv[2], bv[3]});
        const float x[4] = {fmaxf(x01.x, 0.f), fmaxf(x01.y, 0.f), fmaxf(x23.x, 0.f), fmaxf(x23.y, 0.f)};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rb][4 * j + e] = x[e];
        if constexpr (PL == 2) {
          tmax = fmaxf(fmaxf(tmax, x[0]), x[1]);
          tmax = fmaxf(fmaxf(tmax, x[2]), x[3]);
        }
      }
    }
    float sh = 1.f;
    if constexpr (PL == 2) {
      const uint32_t wm = wave_max_u32(__float_as_uint(tmax));
      if (lane == 0) scr[NW * 64 + wv] = __uint_as_float(wm);
    }
    __syncthreads();
    if constexpr (PL == 2) {
      float hmax = scr[NW * 64];
#pragma unroll
      for (int w = 1; w < NW; ++w) hmax = fmaxf(hmax, scr[NW * 64 + w]);
      uint32_t exh = __float_as_uint(hmax) >> 23;
      exh = exh < 15u ? 15u : (exh > 240u ? 240u : exh);
      exh = __builtin_amdgcn_readfirstlane(exh);
      sh = __uint_as_float((268u - exh) << 23);
      inv_sh = __uint_as_float((exh - 14u) << 23);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        uint16_t* dst = PD + (rowoff + rb * 32 + li) * astr + pch0 + 8 * j + 4 * lh;
        if constexpr (PL == 2) {
          const float4 v = make_float4(acc[rb][4 * j], acc[rb][4 * j + 1], acc[rb][4 * j + 2], acc[rb][4 * j + 3]);
          uint2 h, l;
          split2_h<false>(v, sh, h, l);
          *reinterpret_cast<uint2*>(dst) = h;
          *reinterpret_cast<uint2*>(dst + aplane) = l;
        } else {
          *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pk_bf16(acc[rb][4 * j], acc[rb][4 * j + 1]),
                                                      cvt_pk_bf16(acc[rb][4 * j + 2], acc[rb][4 * j + 3]));
        }
      }
    __syncthreads();
  };
  auto panel_epilogue = [&](auto nrb_tag, int rowoff, int pch0) { panel_epilogue_to(PB, nrb_tag, rowoff, pch0); };
  using full = std::integral_constant<int, NRBT>;       // all of this workgroup's position blocks
  using half = std::integral_constant<int, NRBT / 2>;

  if constexpr (PRE) {
    // fp2.1: A -> B, fp2.2: B -> A; X then sits in panel A with the tile's own power-of-two scale
    stage_sb2(p.wsc_p[0], p.bias_p[0], wv * 32, inv_sa);
    S4G_HD_STRIP(PA, NRBT, 0, wp0, 16, wp1)
    panel_epilogue(full{}, 0, wv * 32);
    stage_sb2(p.wsc_p[1], p.bias_p[1], wv * 32, inv_sh);
    S4G_HD_STRIP(PB, NRBT, 0, wp1, 16, w0(g_first, 0))
    panel_epilogue_to(PA, full{}, 0, wv * 32);
    inv_sa = inv_sh;
  }

  for (int g = g_first, gn; g >= 0; g = gn) {
    const int rest = hmask >> (g + 1);
    gn = rest ? g + 1 + __builtin_ctz(rest) : -1;      // the next head of this launch
    const int gnx = gn >= 0 ? gn : g;                  // whose first fragments the last strips prefetch (the last head: its own again)
    f32x16 acc1[NRBT];   // heads.1 accumulated over the two halves of its 512 inputs, in units of 1 / w_scale
    // ---- heads.0 half 0 -> B;  heads.1 over that half
    stage_sb(0, g * 512 + wv * 32, inv_sa);
    S4G_HD_STRIP(PA, NRBT, 0, w0(g, 0), 16, w1(g, 0))
    panel_epilogue(full{}, 0, wv * 32);
    S4G_HD_STRIP(PB, NRBT, 0, w1(g, 0), 16, w0(g, 1))
#pragma unroll
    for (int rb = 0; rb < NRBT; ++rb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {      // (packed fp32: two values per issue slot)
        const f32x2 v = f32x2{acc[rb][r], acc[rb][r + 1]} * f32x2{inv_sh, inv_sh};
        acc1[rb][r] = v.x;
        acc1[rb][r + 1] = v.y;
      }
    // ---- heads.0 half 1 -> B;  heads.1 over that half, then its epilogue -> B
    stage_sb(0, g * 512 + 256 + wv * 32, inv_sa);
    S4G_HD_STRIP(PA, NRBT, 0, w0(g, 1), 16, w1(g, 1))
    panel_epilogue(full{}, 0, wv * 32);
    S4G_HD_STRIP(PB, NRBT, 0, w1(g, 1), 16, w2(g))
#pragma unroll
    for (int rb = 0; rb < NRBT; ++rb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 v = pk_fma(f32x2{acc[rb][r], acc[rb][r + 1]}, f32x2{inv_sh, inv_sh}, f32x2{acc1[rb][r], acc1[rb][r + 1]});
        acc[rb][r] = v.x;
        acc[rb][r + 1] = v.y;
      }
    stage_sb(1, g * 256 + wv * 32, 1.f);
    panel_epilogue(full{}, 0, wv * 32);
    // ---- heads.2 -> B
    stage_sb(2, g * 256 + wv * 32, inv_sh);
    S4G_HD_STRIP(PB, NRBT, 0, w2(g), 16, w3(g))
    panel_epilogue(full{}, 0, wv * 32);
    // ---- heads.3 (128 wide): wave = (half of the position blocks, wv >> 2) x (32-channel block wv & 3) -> B
    {
      const int row3 = (wv >> 2) * (BM / 2), cg = wv & 3;
      stage_sb(3, g * 128 + cg * 32, inv_sh);
      const WRef after = wv < NRBT ? wl(g) : w0(gnx, 0);
      S4G_HD_STRIP(PB, NRBT / 2, row3, w3(g), 16, after)
      panel_epilogue(half{}, row3, cg * 32);
    }
    // ---- logits: waves 0 .. NRBT - 1, one 32-position block each; (B, c, N) channel-first stores
    if (wv < NRBT) {
      stage_sb(4, g * 32, inv_sh);
      S4G_HD_STRIP(PB, 1, wv * 32, wl(g), 8, w0(gnx, 0))
      const int row = p0 + wv * 32 + li;
      const int b = row / p.N, pt = row - b * p.N;
      const int nch = p.ch[g];
      float* __restrict__ base = p.out[g] + (size_t)b * (size_t)p.obs[g] + pt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 sc4 = *reinterpret_cast<const float4*>(epi_s + 8 * j + 4 * lh);
        const float4 b4 = *reinterpret_cast<const float4*>(epi_s + 32 + 8 * j + 4 * lh);
        const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
        const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 8 * j + 4 * lh + e;
          float x = __fmaf_rn(acc[0][4 * j + e], scv[e], bv[e]);
          if (g == p.sigmoid_head) x = 1.0f / (1.0f + expf(-x));
          if (c < nch && row < p.P) base[(size_t)c * p.N] = x;
        }
      }
    }
  }
#undef S4G_HD_STRIP
#undef S4G_HD_TERM
}

template <int PL, int HD_RING, int NRBT, bool PRE>
static int launch_heads_cfg(const HeadsParams& p, hipStream_t st) {
  constexpr int BM = 32 * NRBT;
  constexpr size_t lds = sizeof(uint16_t) * 2 * PL * BM * (256 + 8) + sizeof(float) * (8 * 64 + 16);
  static_assert(lds <= 160 * 1024, "one workgroup's panels must fit a CU's LDS");
  static LdsAttrCache lds_cache;
  if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(&mlp_heads_kernel<PL, HD_RING, NRBT, PRE>), lds, lds_cache)) return rc;
  hipLaunchKernelGGL((mlp_heads_kernel<PL, HD_RING, NRBT, PRE>), dim3((unsigned)((p.P + BM - 1) / BM)), dim3(512), lds, st, p);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}
template <int PL, int HD_RING, int NRBT>
static int launch_heads(const HeadsParams& p, hipStream_t st) {
  return p.Wp[0] ? launch_heads_cfg<PL, HD_RING, NRBT, true>(p, st) : launch_heads_cfg<PL, HD_RING, NRBT, false>(p, st);
}

}  // namespace s4g

extern "C" int s4g_heads_chain_f32(const s4g_heads_desc_t* d, s4g_stream_t stream) {
  using namespace s4g;
  if (!d || d->P < 0 || d->N <= 0) return S4G_EINVAL;
  const bool pre = d->pre_W_frag[0] != nullptr;
  if (!pre && (!d->X || (d->ldx & 3) || d->ldx < 256 || ((uintptr_t)d->X & 15))) return S4G_EINVAL;
  if (pre && (!d->pre_W_frag[1] || !d->pre_bias[0] || !d->pre_bias[1] || !d->pre_nidx || !d->pre_nw ||
              !d->pre_sparse || !d->pre_lbias || d->pre_N2 < 3 || ((uintptr_t)d->pre_sparse & 15) ||
              ((uintptr_t)d->pre_dense & 15) || ((uintptr_t)d->pre_lbias & 15) ||
              (d->precision == S4G_GEMM_F16X2 && (!d->pre_w_inv_scale[0] || !d->pre_w_inv_scale[1]))))
    return S4G_EINVAL;
  if (d->precision != S4G_GEMM_F16X2 && d->precision != S4G_GEMM_BF16) return S4G_EINVAL;
  if (d->C != 256 || d->H0 != 512 || d->H1 != 256 || d->H2 != 256 || d->H3 != 128) return S4G_EUNSUPPORTED;
  HeadsParams p;
  p.P = d->P;
  p.N = d->N;
  p.X = d->X;
  p.ldx = d->ldx;
  for (int l = 0; l < 5; ++l) {
    if (!d->W_frag[l] || !d->bias[l] || (d->precision == S4G_GEMM_F16X2 && !d->w_inv_scale[l])) return S4G_EINVAL;
    p.W[l] = (const uint16_t*)d->W_frag[l];
    p.bias[l] = d->bias[l];
    p.wsc[l] = d->w_inv_scale[l];
  }
  p.head_mask = d->head_mask ? d->head_mask : 15;
  if (p.head_mask < 0 || p.head_mask > 15) return S4G_EINVAL;
  for (int h = 0; h < 4; ++h) {
    p.out[h] = nullptr;
    p.ch[h] = 0;
    p.obs[h] = 0;
    if (!((p.head_mask >> h) & 1)) continue;
    if (!d->out[h] || d->channels[h] <= 0 || d->channels[h] > 32) return S4G_EINVAL;
    p.out[h] = d->out[h];
    p.ch[h] = d->channels[h];
    if (d->out_batch_stride != 0 && d->out_batch_stride < (int64_t)d->channels[h] * d->N) return S4G_EINVAL;
    p.obs[h] = d->out_batch_stride != 0 ? (long long)d->out_batch_stride : (long long)d->channels[h] * d->N;
  }
  p.sigmoid_head = d->sigmoid_head;
  p.a_amax = d->a_amax;
  p.a_floor = d->a_amax_floor;
  p.rps = d->rows_per_scene > 0 ? d->rows_per_scene : 0;
  for (int l = 0; l < 2; ++l) {
    p.Wp[l] = pre ? (const uint16_t*)d->pre_W_frag[l] : nullptr;
    p.bias_p[l] = pre ? d->pre_bias[l] : nullptr;
    p.wsc_p[l] = pre ? d->pre_w_inv_scale[l] : nullptr;
  }
  p.nidx = pre ? d->pre_nidx : nullptr;
  p.nw = pre ? d->pre_nw : nullptr;
  p.sparse = pre ? d->pre_sparse : nullptr;
  p.dense = pre ? d->pre_dense : nullptr;
  p.lbias = pre ? d->pre_lbias : nullptr;
  p.a_amax2 = pre ? d->pre_a_amax2 : nullptr;
  p.N2 = pre ? d->pre_N2 : 0;
  if (d->precision == S4G_GEMM_F16X2 && !d->a_amax && !(d->a_amax_floor > 0.f)) return S4G_EINVAL;
  if (d->P == 0) return S4G_OK;
  hipStream_t st = (hipStream_t)stream;
  // W fragment ring four 16-deep steps deep (two and eight measured 1-4 % slower in rounds 3 and 4); the
  // single-plane bf16 form owns 128 positions per workgroup (its 64-position form spilled)
  if (d->precision == S4G_GEMM_F16X2) return launch_heads<2, 4, 2>(p, st);
  return launch_heads<1, 4, 4>(p, st);
}
