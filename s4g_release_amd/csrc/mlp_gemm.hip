// Shared-MLP contraction on the matrix cores of gfx950.
//
// Replaces, for inference, what the reference spreads over cuDNN/cuBLAS and
// ATen kernels per layer (reference nn_utils/conv.py:28-34,68-74: 1x1 conv ->
// BatchNorm -> ReLU; pointnet2_utils/modules.py:42-50 group + concat,
// :242-243 max over neighbours, :118-127 interpolate + concat):
//
//   Y[p][n] = act( sum_k A[p][k] * W[n][k] + bias[n] )
//
// with BatchNorm folded into W/bias on the host, positions p as GEMM rows and
// all activations stored CHANNELS-LAST ([position][channel]) so that
//   * both MFMA operands are K-contiguous (16-byte LDS reads feed 4 MFMAs),
//   * a grouped neighbour row is ONE contiguous C*4-byte gather,
//   * the max over the K=64 neighbours of a centroid is an in-register max
//     over the accumulator rows of one wave (the (B,C,M,K) pre-max tensor,
//     335 MB/scene at SA1, is never written).
// The A operand is produced by a fused loader:
//   PLAIN   rows of a channels-last activation (optionally a column slice)
//   GATHER  set-abstraction grouping: row = [feat[idx[p]] | xyz[idx[p]] - ctr]
//           (modules.py:42-50; K order is [feat, xyz] -- W is permuted to match)
//   INTERP  feature propagation: row = [sum_k w_k * sparse[idx_k] | dense[p]]
//           (modules.py:118-127, interpolate_kernel.cu:160-174)
//   GATHER_MLP1 / GATHER_ADD / INTERP_ADD  the level's (linear) first layer applied before the
//           grouping / interpolation and finished inside this loader (see ALoader)
//
// Kernels in this file, oldest first (each has its own header further down):
//   mlp_gemm_kernel                  fp32 MFMA (v_mfma_f32_32x32x2_f32), the exact mode described here
//   mlp_gemm_bf16x3_kernel           three bf16 planes, six products: exact-class alternative
//   mlp_gemm_f16x2_kernel            two fp16 planes, three products (fp32-class; PL = 1: one bf16
//                                    plane): the tiled single-layer form, both operands through LDS
//   mlp_gemm_f16x2_resident_kernel   short contractions with the A panel resident in LDS (measurement
//                                    builds only: variants/gemm_resident.inc)
//   mlp_chain_kernel                 two or three layers per launch, intermediates in LDS, W streamed
//                                    in MFMA-fragment order: the kernels the shipped network runs on
// (the four heads as one launch live in mlp_heads.hip, helpers shared by both in mlp_common.h).
//
// Arithmetic of the fp32 kernel: v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulate
// (bit-for-bit a k-ordered fmaf chain), so results stay within fp32 round-off
// of the reference's fp32 convolution (tested to 1e-4 abs on the outputs).
//
// Tiling: 128 positions x 128 channels per 256-thread workgroup, 4 waves as
// 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).  K is walked
// in 32-wide tiles staged global -> registers -> LDS (double buffered, one
// barrier per tile; next tile's global loads are issued before the MFMAs of
// the current one).  LDS rows are padded to 36 floats: conflict-free
// ds_read_b128 for the 16-lane groups of gfx950.  Workgroup ids are remapped
// so that the N-tiles sharing an A panel run back to back on one XCD (L2).
#include <stdlib.h>

#include "mlp_common.h"

#include <type_traits>

#ifndef S4G_CHAIN_ABLATE
#define S4G_CHAIN_ABLATE 0
#endif

namespace s4g {


constexpr int GM_BM = 128;
constexpr int GM_BN = 128;
constexpr int GM_BK = 32;
constexpr int GM_LDS = 36;  // padded row stride (floats)
constexpr int GM_THREADS = 256;

enum { LOAD_PLAIN = 0, LOAD_GATHER = 1, LOAD_INTERP = 2, LOAD_GATHER_MLP1 = 3, LOAD_GATHER_ADD = 4,
       LOAD_INTERP_ADD = 5,
       // internal (never in a descriptor): GATHER_MLP1 on pre-gathered rel_xyz4 records in the f16x2 chain kernel --
       // the 3 -> C first layer runs on the matrix cores too (mlp_chain_kernel, "phase 0")
       LOAD_REL_MLP1 = 6 };
enum { EPI_STORE = 0, EPI_MAX = 1, EPI_CHANNEL_FIRST = 2 };

struct GemmParams {
  // problem
  int P, Cin, Kpad, Cout, relu;
  const float* W;     // [groups][Cout][Kpad]
  const float* bias;  // [groups][Cout]
  // PLAIN: A + p*lda + a_coff + g*a_gcol
  const float* A;
  int lda, a_coff, a_gcol;
  // GATHER: feat (B*N, Cf) channels-last, xyz (B,3,N), ctr (B,3,M), idx (B*M*K) int32
  const int* gidx;
  const float* feat;
  const float* xyz;
  const float* ctr;
  int Cf, N, M, K;
  // GATHER_MLP1: first SA layer (xyz only) evaluated in the loader:
  // A[p][k] = relu(w1[k].x*rx + w1[k].y*ry + w1[k].z*rz + w1[k].w), k < Cin
  const float4* mlp1;
  const float4* rel4;   // optional: the rows' (xyz_j - ctr_m, 0) records, read instead of following gidx
  // optional (mlp_chain_kernel, GATHER_MLP1 + rel4 + MAX): rel4 holds every centroid's DISTINCT rows only
  // (s4g_group_rel_xyz_unique_i32): seg4[row / 4] = output row of each group of 4 rows (-1: none),
  // seg_rows[scene] = rows the scene occupies behind its base scene * rps
  const int* seg4;
  const int* seg_rows;
  // optional second output tensor of a plain single layer (two layers that read the SAME input as one launch,
  // W / bias concatenated): output channels >= split_n (a multiple of 256) go to out2 (row stride ldc2, column
  // n - split_n) and publish their maxima to out_amax2
  float* out2;
  int ldc2, split_n;
  uint32_t* out_amax2;
  // INTERP: sparse (B*N2, C2), dense (B*N1, C1), nidx/nw (B*N1, 3)
  const int* nidx;
  const float* nw;
  const float* sparse;
  const float* dense;
  int C2, C1, N2, N1;
  // output
  float* out;  // STORE: out + p*ldc + c_coff + g*c_gcol ; MAX: row = p / K
  int ldc, c_coff, c_gcol;
  int w_gstride, b_gstride;
  // CHANNEL_FIRST: out[b][c][n] split over up to 4 tensors
  float* cf_ptr[4];
  int cf_start[5];
  int cf_sigmoid_from;  // channels >= this get a sigmoid
  int cf_N;             // points per batch element
  int mtiles, ntiles;
  // bf16x3 path: W split into three bf16 planes [3][groups][Cout][Kpad16]
  const uint16_t* W3;
  int Kpad16;
  int bf16_single;  // 1: plain bf16 contraction (hi planes only), reduced precision
  size_t w3_plane;  // elements between planes
  // f16x2 path: W split into two fp16 planes of W * w_scale[n] (power-of-two
  // per-channel scales), activations scaled by a power of two derived from
  // the running maxima below so that both fit fp16's range
  const uint16_t* Wh2;         // [2][groups][Cout][Kpad16] fp16
  const float* w_inv_scale;    // [groups][Cout]
  const float* a_amax;         // 64 slots or NULL
  const float* a_amax2;        // 64 slots or NULL
  float a_amax_floor;
  uint32_t* out_amax;          // 64 slots or NULL
  int rps;                     // loader rows per scene: slot row s of a_amax / out_amax belongs to scene s (0: one row)
  const uint16_t* Wfrag;       // f16x2 planes in MFMA-fragment order (resident-A kernel) or NULL
  const float* lbias;          // INTERP_ADD loader: bias of the layer whose output the loader forms
  // fused second layer (mlp_chain_kernel): out = max_K relu(bn(W2 relu(bn(W A))))
  const uint16_t* Wfrag2;
  const float* w_inv_scale2;
  const float* bias2;
  int Cout2, relu2;
  // optional third layer of the chain (then layer 2 is as wide as layer 1 and also stays in LDS)
  const uint16_t* Wfrag3;
  const float* w_inv_scale3;
  const float* bias3;
  int Cout3, relu3;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// scene b and centroid m of grouped row pos = p0 + r (rows are [scene][centroid][neighbour]):
// ONE division per workgroup (p0 is uniform) instead of two per row -- the rows of a tile can only
// step over a scene boundary a few times, and K is a power of two wherever the fast path groups
__device__ __forceinline__ void gather_row_bm(int M, int K, int p0, int r, int& b, int& m) {
  const int MK = M * K;
  const int b0 = __builtin_amdgcn_readfirstlane(p0 / MK);
  int rem = p0 - b0 * MK + r;
  b = b0;
  while (rem >= MK) {
    rem -= MK;
    ++b;
  }
  m = (K & (K - 1)) == 0 ? rem >> (31 - __clz(K)) : rem / K;
}

template <int LOADER, int RPT = 4, int RS = 32>
struct ALoader {
  // per-thread: RPT rows (t>>3)+RS*s, one 4-float chunk (t&7)
  const float* src0[RPT];  // PLAIN: row base; GATHER: feat row base; INTERP: unused
  bool ok[RPT];
  // GATHER tail
  float rel[RPT][3];
  // INTERP
  int i3[RPT][3];
  float w3[RPT][3];
  size_t drow[RPT];

  __device__ __forceinline__ void init(const GemmParams& p, int p0, int g, int t) {
    constexpr bool GATHERS = LOADER == LOAD_GATHER_MLP1 || LOADER == LOAD_GATHER_ADD || LOADER == LOAD_GATHER;
    // Pass 1: every row's position and -- for the gathering loaders -- its neighbour index.  The
    // index loads are unconditional (a row past the end reads row 0's) and all issued before the
    // first dependent gather: one round trip for the thread's rows instead of one per row (a
    // load behind a per-row guard is not hoisted over the previous row's wait).
    int pp_[RPT], j_[RPT], b_[RPT], m_[RPT];
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
      const int r = (t >> 3) + RS * s;
      const int pos = p0 + r;
      ok[s] = pos < p.P;
      pp_[s] = ok[s] ? pos : 0;
      j_[s] = b_[s] = m_[s] = 0;
      if constexpr (GATHERS) {
        if (!(LOADER == LOAD_GATHER_MLP1 && p.rel4)) {
          gather_row_bm(p.M, p.K, p0, ok[s] ? r : 0, b_[s], m_[s]);   // (a row past the end: the tile's first row)
          j_[s] = p.gidx[pp_[s]];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
      const int pp = pp_[s], j = j_[s], b = b_[s], m = m_[s];
      if constexpr (LOADER == LOAD_PLAIN) {
        src0[s] = p.A + (size_t)pp * p.lda + p.a_coff + g * p.a_gcol;
      } else if constexpr (LOADER == LOAD_GATHER_MLP1) {
        if (p.rel4) {   // pre-gathered by s4g_group_rel_xyz_i32: one coalesced 16-byte read per row
          const float4 r4 = p.rel4[pp];
          rel[s][0] = r4.x;
          rel[s][1] = r4.y;
          rel[s][2] = r4.z;
          continue;
        }
        const float* x = p.xyz + (size_t)b * 3 * p.N;
        const float* c = p.ctr + (size_t)b * 3 * p.M;
        rel[s][0] = __fsub_rn(x[j], c[m]);
        rel[s][1] = __fsub_rn(x[p.N + j], c[p.M + m]);
        rel[s][2] = __fsub_rn(x[2 * p.N + j], c[2 * p.M + m]);
      } else if constexpr (LOADER == LOAD_GATHER_ADD) {
        // first SA layer applied to the level's features BEFORE the grouping (it is linear):
        // A[p][k] = relu(F[b*N + j][k] + w1[k] . (xyz_j - ctr_m, 1))
        src0[s] = p.feat + ((size_t)b * p.N + j) * p.Cf;
        const float* x = p.xyz + (size_t)b * 3 * p.N;
        const float* c = p.ctr + (size_t)b * 3 * p.M;
        rel[s][0] = __fsub_rn(x[j], c[m]);
        rel[s][1] = __fsub_rn(x[p.N + j], c[p.M + m]);
        rel[s][2] = __fsub_rn(x[2 * p.N + j], c[2 * p.M + m]);
      } else if constexpr (LOADER == LOAD_GATHER) {
        src0[s] = p.feat ? p.feat + ((size_t)b * p.N + j) * p.Cf : nullptr;
        if ((t & 7) == ((p.Cf & 31) >> 2)) {   // the lane whose 4-float chunk holds the xyz columns
          const float* x = p.xyz + (size_t)b * 3 * p.N;
          const float* c = p.ctr + (size_t)b * 3 * p.M;
          // group_xyz -= new_xyz (modules.py:44): one rounded subtraction
          rel[s][0] = __fsub_rn(x[j], c[m]);
          rel[s][1] = __fsub_rn(x[p.N + j], c[p.M + m]);
          rel[s][2] = __fsub_rn(x[2 * p.N + j], c[2 * p.M + m]);
        }
      } else {   // INTERP, INTERP_ADD
        const int bq = pp / p.N1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          i3[s][k] = bq * p.N2 + p.nidx[(size_t)pp * 3 + k];
          w3[s][k] = p.nw[(size_t)pp * 3 + k];
        }
        drow[s] = (size_t)pp;
      }
    }
  }

  // 4-float chunk of logical A row s at columns k0..k0+3
  __device__ __forceinline__ float4 load(const GemmParams& p, int s, int k0, int t) const {
    if constexpr (LOADER == LOAD_GATHER_MLP1) {
      // the weight loads are unconditional (rows past the end and columns past Cin are zeroed by
      // a select afterwards): the rows of a thread share them, and a guarded load is not merged
      const bool live = ok[s] && k0 < p.Cin;
      const int kk = k0 < p.Cin ? k0 : 0;
      float4 r;
      float* rp = reinterpret_cast<float*>(&r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 w = p.mlp1[kk + e];
        const float v = __fmaf_rn(w.z, rel[s][2], __fmaf_rn(w.y, rel[s][1], __fmaf_rn(w.x, rel[s][0], w.w)));
        rp[e] = live ? fmaxf(v, 0.f) : 0.f;
      }
      return r;
    }
    if constexpr (LOADER == LOAD_GATHER_ADD) {
      // (unconditional loads as above; a row past the end reads row 0's feature row)
      const bool live = ok[s] && k0 < p.Cin;
      const int kk = k0 < p.Cin ? k0 : 0;
      const float4 f = *reinterpret_cast<const float4*>(src0[s] + kk);
      const float fv[4] = {f.x, f.y, f.z, f.w};
      float4 r;
      float* rp = reinterpret_cast<float*>(&r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 w = p.mlp1[kk + e];
        const float v = __fmaf_rn(w.z, rel[s][2], __fmaf_rn(w.y, rel[s][1], __fmaf_rn(w.x, rel[s][0], w.w)));
        rp[e] = live ? fmaxf(__fadd_rn(fv[e], v), 0.f) : 0.f;
      }
      return r;
    }
    if (!ok[s]) return f4zero();
    if constexpr (LOADER == LOAD_PLAIN) {
      if (k0 >= p.Cin) return f4zero();
      return *reinterpret_cast<const float4*>(src0[s] + k0);
    } else if constexpr (LOADER == LOAD_INTERP_ADD) {
      // first FP layer applied before the interpolation: A = relu(y + bias + sum_k w_k S[idx_k])
      if (k0 >= p.Cin) return f4zero();
      const float4 a = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[s][0] * p.C2 + k0);
      const float4 b = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[s][1] * p.C2 + k0);
      const float4 c = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[s][2] * p.C2 + k0);
      const float4 bb = *reinterpret_cast<const float4*>(p.lbias + k0);
      float4 y = f4zero();
      if (p.dense) y = *reinterpret_cast<const float4*>(p.dense + drow[s] * p.C2 + k0);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w};
      const float yv[4] = {y.x, y.y, y.z, y.w}, bi[4] = {bb.x, bb.y, bb.z, bb.w};
      float4 r;
      float* rp = reinterpret_cast<float*>(&r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = __fmul_rn(av[e], w3[s][0]);
        acc = __fadd_rn(acc, __fmul_rn(bv[e], w3[s][1]));
        acc = __fadd_rn(acc, __fmul_rn(cv[e], w3[s][2]));
        rp[e] = fmaxf(__fadd_rn(__fadd_rn(yv[e], acc), bi[e]), 0.f);   // same order as interp_add_cl_kernel
      }
      return r;
    } else if constexpr (LOADER == LOAD_GATHER) {
      if (k0 < p.Cf) return *reinterpret_cast<const float4*>(src0[s] + k0);
      if (k0 == p.Cf) return make_float4(rel[s][0], rel[s][1], rel[s][2], 0.f);
      return f4zero();
    } else {
      if (k0 < p.C2) {
        const float4 a = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[s][0] * p.C2 + k0);
        const float4 b = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[s][1] * p.C2 + k0);
        const float4 c = *reinterpret_cast<const float4*>(p.sparse + (size_t)i3[s][2] * p.C2 + k0);
        float4 r;
        // acc = 0; acc += f_k * w_k, k = 0,1,2 (interpolate_kernel.cu:160-174)
        r.x = __fadd_rn(__fadd_rn(__fmul_rn(a.x, w3[s][0]), __fmul_rn(b.x, w3[s][1])), __fmul_rn(c.x, w3[s][2]));
        r.y = __fadd_rn(__fadd_rn(__fmul_rn(a.y, w3[s][0]), __fmul_rn(b.y, w3[s][1])), __fmul_rn(c.y, w3[s][2]));
        r.z = __fadd_rn(__fadd_rn(__fmul_rn(a.z, w3[s][0]), __fmul_rn(b.z, w3[s][1])), __fmul_rn(c.z, w3[s][2]));
        r.w = __fadd_rn(__fadd_rn(__fmul_rn(a.w, w3[s][0]), __fmul_rn(b.w, w3[s][1])), __fmul_rn(c.w, w3[s][2]));
        return r;
      }
      const int kd = k0 - p.C2;
      if (kd < p.C1) return *reinterpret_cast<const float4*>(p.dense + drow[s] * p.C1 + kd);
      return f4zero();
    }
  }
};

// Shared epilogue.  D layout of every 32x32 MFMA: col = lane & 31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).

// NCB = 32-wide column blocks per wave (2: four waves as 2x2; 1: eight waves as 2x4).
// `wave` indexes the staging slab, wr / wc the 64-row half and the column strip.
template <int EPI, int NCB>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[2][NCB],
                                              const float* __restrict__ bg, int g, int p0,
                                              int n0, int wave, int wr, int wc, int li, int lh,
                                              float* __restrict__ lds_stage) {
  constexpr int WCOLS = 32 * NCB;        // columns per wave
  constexpr int GE_STRIDE = WCOLS + 4;   // floats per staged output row
  if constexpr (EPI == EPI_STORE) {
    // Wide stores: each wave stages 32 rows x 64 channels of its tile in (now
    // idle) LDS and writes them back as float4 -- 16 store instructions per
    // wave instead of 64 dword stores (the tail is store-ISSUE bound).
    const bool vec_ok = ((p.ldc | p.c_coff | p.c_gcol | p.Cout) & 3) == 0 &&
                        ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
    if (vec_ok) {
      const int lane = li + 32 * lh;
      float* st = lds_stage + wave * 32 * GE_STRIDE;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int n = n0 + wc * WCOLS + cb * 32 + li;
          const float bias = n < p.Cout ? bg[n] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[rb][cb][r] + bias;
            if (p.relu) v = fmaxf(v, 0.f);
            st[((r & 3) + 8 * (r >> 2) + 4 * lh) * GE_STRIDE + cb * 32 + li] = v;
          }
        }
        constexpr int LPR = WCOLS / 4;   // lanes per staged row
        const int c4 = (lane % LPR) * 4;
        const int col = n0 + wc * WCOLS + c4;
#pragma unroll
        for (int i = 0; i < 32 * LPR / 64; ++i) {
          const int rl = lane / LPR + (64 / LPR) * i;
          const float4 v = *reinterpret_cast<const float4*>(st + rl * GE_STRIDE + c4);
          const int row = p0 + wr * 64 + rb * 32 + rl;
          if (row < p.P && col < p.Cout)
            *reinterpret_cast<float4*>(p.out + (size_t)row * p.ldc + p.c_coff + g * p.c_gcol + col) = v;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int n = n0 + wc * WCOLS + cb * 32 + li;
    const bool nok = n < p.Cout;
    const float bias = nok ? bg[n] : 0.f;
    if constexpr (EPI == EPI_STORE) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = p0 + wr * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float v = acc[rb][cb][r] + bias;
          if (p.relu) v = fmaxf(v, 0.f);
          if (nok && row < p.P)
            p.out[(size_t)row * p.ldc + p.c_coff + g * p.c_gcol + n] = v;
        }
    } else if constexpr (EPI == EPI_MAX) {
      // max over groups of K consecutive rows, K in {16, 32, 64}; then
      // bias + ReLU (monotone, so max-then-activate == activate-then-max).
      if (p.K == 64) {
        float m = acc[0][cb][0];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[rb][cb][r]);
        m = fmaxf(m, __shfl_xor(m, 32));
        float v = m + bias;
        if (p.relu) v = fmaxf(v, 0.f);
        const int grp = (p0 + wr * 64) >> 6;
        if (nok && lh == 0 && p0 + wr * 64 < p.P)
          p.out[(size_t)grp * p.ldc + p.c_coff + n] = v;
      } else if (p.K == 32) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          float m = acc[rb][cb][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[rb][cb][r]);
          m = fmaxf(m, __shfl_xor(m, 32));
          float v = m + bias;
          if (p.relu) v = fmaxf(v, 0.f);
          const int row0 = p0 + wr * 64 + rb * 32;
          if (nok && lh == 0 && row0 < p.P)
            p.out[(size_t)(row0 >> 5) * p.ldc + p.c_coff + n] = v;
        }
      } else {  // K == 16: rows 0-15 are regs with (r>>2) in {0,1}, rows 16-31 {2,3}
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            float m = acc[rb][cb][8 * hh];
#pragma unroll
            for (int r = 1; r < 8; ++r) m = fmaxf(m, acc[rb][cb][8 * hh + r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            float v = m + bias;
            if (p.relu) v = fmaxf(v, 0.f);
            const int row0 = p0 + wr * 64 + rb * 32 + 16 * hh;
            if (nok && lh == 0 && row0 < p.P)
              p.out[(size_t)(row0 >> 4) * p.ldc + p.c_coff + n] = v;
          }
      }
    } else {  // EPI_CHANNEL_FIRST: out[b][c][n_pt], 4 consecutive points per store
      int head = 0;
#pragma unroll
      for (int h2 = 1; h2 < 4; ++h2)
        if (n >= p.cf_start[h2]) head = h2;
      const int cl = n - p.cf_start[head];
      const int ch = p.cf_start[head + 1] - p.cf_start[head];
      float* base = p.cf_ptr[head];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int row = p0 + wr * 64 + rb * 32 + 8 * r4 + 4 * lh;
          if (nok && n < p.cf_start[4] && row < p.P) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[rb][cb][4 * r4 + e] + bias;
              if (p.relu) x = fmaxf(x, 0.f);
              if (n >= p.cf_sigmoid_from) x = 1.0f / (1.0f + expf(-x));
              v[e] = x;
            }
            const int b = row / p.cf_N;
            const int pt = row - b * p.cf_N;
            float* dst = base + ((size_t)b * ch + cl) * p.cf_N + pt;
            if (pt + 3 < p.cf_N && (p.cf_N & 3) == 0) {
              *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int row_e = row + e;
                if (row_e < p.P) {
                  const int be = row_e / p.cf_N;
                  const int pe = row_e - be * p.cf_N;
                  base[((size_t)be * ch + cl) * p.cf_N + pe] = v[e];
                }
              }
            }
          }
        }
    }
  }
}

template <int LOADER, int EPI>
__global__ __launch_bounds__(GM_THREADS, 2) void mlp_gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                              // [2][BM][36]
  float* Ws = smem + 2 * GM_BM * GM_LDS;         // [2][BN][36]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int g = blockIdx.y;

  // XCD-aware bijective remap: consecutive ids on one XCD walk the N-tiles of
  // one M-tile, so the shared A panel is an L2 hit.
  const int nb = p.mtiles * p.ntiles;
  int id = blockIdx.x;
  {
    const int q = nb >> 3, r = nb & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = id / p.ntiles;
  const int nt = id - mt * p.ntiles;
  const int p0 = mt * GM_BM;
  const int n0 = nt * GM_BN;

  const float* __restrict__ Wg = p.W + (size_t)g * p.w_gstride;
  const float* __restrict__ bg = p.bias + (size_t)g * p.b_gstride;

  ALoader<LOADER> ld;
  ld.init(p, p0, g, t);
  const int chunk = t & 7;
  const int srow = t >> 3;
  const float* wrow[4];
  bool wok[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int n = n0 + srow + 32 * s;
    wok[s] = n < p.Cout;
    wrow[s] = Wg + (size_t)(wok[s] ? n : 0) * p.Kpad;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int a_off = (wr * 64 + li) * GM_LDS + 4 * lh;
  const int b_off = (wc * 64 + li) * GM_LDS + 4 * lh;
  const int st_off = srow * GM_LDS + chunk * 4;

  float4 ra[4], rw[4];
  const int ntile_k = (p.Kpad + GM_BK - 1) / GM_BK;

  auto gload = [&](int kt) {
    const int k0 = kt * GM_BK + chunk * 4;
    const bool live = k0 < p.Kpad;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ra[s] = live ? ld.load(p, s, k0, t) : f4zero();
      rw[s] = (live && wok[s]) ? *reinterpret_cast<const float4*>(wrow[s] + k0) : f4zero();
    }
  };
  auto lstore = [&](int buf) {
    float* a = As + buf * GM_BM * GM_LDS + st_off;
    float* w = Ws + buf * GM_BN * GM_LDS + st_off;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      *reinterpret_cast<float4*>(a + 32 * s * GM_LDS) = ra[s];
      *reinterpret_cast<float4*>(w + 32 * s * GM_LDS) = rw[s];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < ntile_k; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < ntile_k) gload(kt + 1);
    const int krem = p.Kpad - kt * GM_BK;
    const int nq = krem >= GM_BK ? 4 : (krem >> 3);
    const float* a = As + buf * GM_BM * GM_LDS + a_off;
    const float* w = Ws + buf * GM_BN * GM_LDS + b_off;
    for (int q = 0; q < nq; ++q) {
      const float4 a0 = *reinterpret_cast<const float4*>(a + 8 * q);
      const float4 a1 = *reinterpret_cast<const float4*>(a + 32 * GM_LDS + 8 * q);
      const float4 b0 = *reinterpret_cast<const float4*>(w + 8 * q);
      const float4 b1 = *reinterpret_cast<const float4*>(w + 32 * GM_LDS + 8 * q);
      const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
      const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][r], bv[0][r], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][r], bv[1][r], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][r], bv[0][r], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][r], bv[1][r], acc[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < ntile_k) lstore(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<EPI, 2>(p, acc, bg, g, p0, n0, wave, wr, wc, li, lh, smem);
}

// ---------------------------------------------------------------------------
// bf16x3 variant: fp32-equivalent contraction at the bf16 matrix-core rate.
//
// Every fp32 operand is split EXACTLY into three bf16 numbers,
//   x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (8 + 8 + 8 = 24 significand bits: nothing of the fp32 value is lost), and
//   a*b ~= a1*b1 + (a1*b2 + a2*b1) + (a1*b3 + a2*b2 + a3*b1)
// drops only the terms below 2^-24 |a||b| -- the size of one fp32 rounding.
// Products of bf16 pairs are exact in fp32 and v_mfma_f32_32x32x16_bf16
// accumulates in fp32, so the result carries fp32-class error (measured
// against fp64 in tests/test_fused_gpu.py next to the exact-fp32 kernel) while
// six bf16 MFMAs per 16-deep step cost 6 x 32 cycles against 8 x 64 for the
// fp32-input MFMA: 2.67x less matrix-pipe time.  Weights are split once on the
// host; activations stay fp32 in HBM and are split by the loader while it
// stages them into LDS.
// ---------------------------------------------------------------------------


// split 4 floats into three planes of 4 bf16 (packed as uint2 each)
__device__ __forceinline__ void split3(const float4 v, uint2& h, uint2& m, uint2& l) {
  h.x = cvt_pk_bf16(v.x, v.y);
  h.y = cvt_pk_bf16(v.z, v.w);
  const float r0 = v.x - __uint_as_float(h.x << 16), r1 = v.y - __uint_as_float(h.x & 0xFFFF0000u);
  const float r2 = v.z - __uint_as_float(h.y << 16), r3 = v.w - __uint_as_float(h.y & 0xFFFF0000u);
  m.x = cvt_pk_bf16(r0, r1);
  m.y = cvt_pk_bf16(r2, r3);
  const float s0 = r0 - __uint_as_float(m.x << 16), s1 = r1 - __uint_as_float(m.x & 0xFFFF0000u);
  const float s2 = r2 - __uint_as_float(m.y << 16), s3 = r3 - __uint_as_float(m.y & 0xFFFF0000u);
  l.x = cvt_pk_bf16(s0, s1);
  l.y = cvt_pk_bf16(s2, s3);
}

// A stays fp32 in LDS (plain staging copy, no VALU in the synchronous part);
// each wave splits its own A fragments in registers right before use, in the
// issue shadow of the MFMAs.  W arrives pre-split (host).  One LDS stage of
// 49 KB + ~168 VGPRs => three workgroups (12 waves) per CU, so while one
// workgroup stages or waits at its barrier two others keep the matrix pipe fed.

__device__ __forceinline__ void split3_frag(const float4 lo, const float4 hi, bf16x8& h,
                                            bf16x8& m, bf16x8& l) {
  uint2 h0, m0, l0, h1, m1, l1;
  split3(lo, h0, m0, l0);
  split3(hi, h1, m1, l1);
  h = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
  m = __builtin_bit_cast(bf16x8, make_uint4(m0.x, m0.y, m1.x, m1.y));
  l = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

template <int LOADER, int EPI, int NS, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (NS == 1 && WAVES == 4 && LOADER != LOAD_INTERP && LOADER != LOAD_INTERP_ADD) ? 3 : (NS == 1 ? WAVES / 2 : 1)) void mlp_gemm_bf16x3_kernel(const GemmParams p) {
  constexpr int THREADS = 64 * WAVES;
  constexpr int WC = WAVES / 2;          // column strips (wave grid is 2 x WC)
  constexpr int NCB = 4 / WC;            // 32-wide column blocks per wave
  constexpr int RPT = 1024 / THREADS;    // A rows per thread
  constexpr int RS = THREADS / 8;        // A row step between them
  constexpr int WPT = 512 / THREADS;     // W rows per thread (per plane)
  constexpr int WRS = THREADS / 4;
  constexpr int BK = 32 * NS;
  constexpr int ASTR = BK + 4;   // fp32 A row stride
  constexpr int WSTR = BK + 8;   // bf16 W row stride
  extern __shared__ __attribute__((aligned(16))) float smemf[];
  float* Af = smemf;                                                     // [BM][36] fp32
  uint16_t* Ws = reinterpret_cast<uint16_t*>(smemf + GM_BM * ASTR);   // [3][BN][WSTR] bf16

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int g = blockIdx.y;
  const int nb = p.mtiles * p.ntiles;
  int id = blockIdx.x;
  {
    const int q = nb >> 3, r = nb & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = id / p.ntiles;
  const int nt = id - mt * p.ntiles;
  const int p0 = mt * GM_BM;
  const int n0 = nt * GM_BN;
  const float* __restrict__ bg = p.bias + (size_t)g * p.b_gstride;
  const uint16_t* __restrict__ W3g = p.W3 + (size_t)g * p.Cout * p.Kpad16;

  ALoader<LOADER, RPT, RS> ld;
  ld.init(p, p0, g, t);
  const int chunk = t & 7;    // A: 4-float chunk of the 32-wide K tile
  const int srow = t >> 3;    // A: rows srow + RS s
  const int wchunk = t & 3;   // W: 8-bf16 chunk
  const int wrow = t >> 2;    // W: rows wrow + WRS s
  bool wok[WPT];
  size_t woff[WPT];
#pragma unroll
  for (int s = 0; s < WPT; ++s) {
    const int n = n0 + wrow + WRS * s;
    wok[s] = n < p.Cout;
    woff[s] = (size_t)(wok[s] ? n : 0) * p.Kpad16 + wchunk * 8;
  }

  f32x16 acc[2][NCB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wr = wave / WC, wc = wave % WC;
  const int li = lane & 31, lh = lane >> 5;
  const int a_off = (wr * 64 + li) * ASTR + 8 * lh;
  const int b_off = (wc * 32 * NCB + li) * WSTR + 8 * lh;

  float4 ra[NS][RPT];
  uint4 rw[NS][3][WPT];
  const int ntile_k = (p.Kpad16 + BK - 1) / BK;

  auto gload = [&](int kt) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int k0 = kt * BK + u * 32 + chunk * 4;
      const bool live = k0 < p.Kpad16;
#pragma unroll
      for (int s = 0; s < RPT; ++s) ra[u][s] = live ? ld.load(p, s, k0, t) : f4zero();
      const int kw = kt * BK + u * 32 + wchunk * 8;
      const bool wlive = kw < p.Kpad16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int s = 0; s < WPT; ++s)
          rw[u][pl][s] = (wlive && wok[s])
                             ? *reinterpret_cast<const uint4*>(W3g + pl * p.w3_plane + woff[s] + kt * BK + u * 32)
                             : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
#pragma unroll
      for (int s = 0; s < RPT; ++s)
        *reinterpret_cast<float4*>(Af + (srow + RS * s) * ASTR + u * 32 + chunk * 4) = ra[u][s];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int s = 0; s < WPT; ++s)
          *reinterpret_cast<uint4*>(Ws + pl * GM_BN * WSTR + (wrow + WRS * s) * WSTR + u * 32 + wchunk * 8) =
              rw[u][pl][s];
    }
  };

  gload(0);
  for (int kt = 0; kt < ntile_k; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < ntile_k) gload(kt + 1);
    const int krem = p.Kpad16 - kt * BK;
    const int nks = krem >= BK ? 2 * NS : (krem + 15) / 16;
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8 af[2][3];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const float* ap = Af + a_off + rb * 32 * ASTR + ks * 16;
        split3_frag(*reinterpret_cast<const float4*>(ap), *reinterpret_cast<const float4*>(ap + 4),
                    af[rb][0], af[rb][1], af[rb][2]);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        bf16x8 bf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          bf[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
              Ws + pl * GM_BN * WSTR + b_off + cb * 32 * WSTR + ks * 16));
        // six products per tile, smallest terms first, the two row blocks interleaved
#define S4G_X3_TERM(PA, PB)                                                              \
  acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][PA], bf[PB], acc[0][cb], 0, 0, 0); \
  acc[1][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][PA], bf[PB], acc[1][cb], 0, 0, 0);
        if (!p.bf16_single) {
          S4G_X3_TERM(0, 2)
          S4G_X3_TERM(1, 1)
          S4G_X3_TERM(2, 0)
          S4G_X3_TERM(0, 1)
          S4G_X3_TERM(1, 0)
        }
        S4G_X3_TERM(0, 0)
#undef S4G_X3_TERM
      }
    }
    __syncthreads();
  }
  gemm_epilogue<EPI, NCB>(p, acc, bg, g, p0, n0, wave, wr, wc, li, lh, smemf);
}

// ---------------------------------------------------------------------------
// f16x2 variant: fp32-class contraction in THREE fp16 MFMAs per 16-deep step.
//
// fp16 carries 11 significand bits, so two planes hold 22 of fp32's 24:
//   xs = x * s (s a power of two bringing the tensor's max below 2^15),
//   x1 = fp16(xs), x2 = fp16(xs - x1)           (both round-to-nearest-even)
//   a*b ~= (a1*b1 + (a1*b2 + a2*b1)) / (s_a s_b)
// The representation error is 2^-22 |x| per operand and the dropped a2*b2 term
// is below 2^-22 |a||b|: against fp64 the result is as accurate as a plain
// fp32 dot product (rms 2.9e-8 vs 2.8e-8 of sum|a||b| at K=128..1024, see
// tests/test_fused_gpu.py), at half the matrix-pipe time of the bf16x3 form.
// Scales: W per output channel, fixed on the host; activations per tensor,
// from the maximum the PRODUCING launch left in `out_amax` (64 atomicMax
// slots) -- no extra pass over the data.  Elements below 2^-18 of the
// tensor's maximum start losing low-order bits (fp16 subnormals of the second
// plane), which is far under the fp32 round-off of any sum they take part in.
// A is split once per workgroup while it is staged into LDS as two fp16
// planes (same LDS bytes as fp32); W arrives pre-split.
// ---------------------------------------------------------------------------

// LDS rows are 32 halves (64 B, no padding); the 16-byte chunk c of row r sits at
// position c ^ ((r >> 2) & 3).  A 16-lane b128 read phase (16 consecutive rows,
// one chunk) and a 16-lane b128 / 32-lane b64 write phase (4 consecutive rows, all
// chunks) then touch all 64 banks exactly once.
constexpr int GH_STR = 32;

// NCB = 32-wide column blocks per wave: 2 -> 128 x 128 tile (three workgroups per
// CU), 4 -> 128 x 256 tile, wave tile 64 x 128 (two workgroups per CU): 25 % less
// LDS and L1 traffic per MFMA and twice the matrix work per barrier; used when
// Cout > 128.
// PL = 2: the f16x2 split; PL = 1: ONE bf16 plane per operand and one product (S4G_GEMM_BF16: W is
// the hi plane of W_bf16x3, no scales) -- the same tile loop at a third of the matrix work.
template <int LOADER, int EPI, int NCB, int PL>
__global__ __launch_bounds__(256, (NCB == 4 || LOADER == LOAD_INTERP || LOADER == LOAD_INTERP_ADD) ? 2 : 3) void mlp_gemm_f16x2_kernel(
    const GemmParams p) {
  constexpr int BN = 64 * NCB;
  constexpr int RPT = 4, RS = 32, WPT = BN / 64, WRS = 64, BK = 32;
  constexpr int APLANE = GM_BM * GH_STR;  // halves per A plane
  constexpr int WPLANE = BN * GH_STR;     // halves per W plane
  extern __shared__ __attribute__((aligned(16))) float smemf[];
  uint16_t* Ah = reinterpret_cast<uint16_t*>(smemf);  // [2][BM][40]
  uint16_t* Wh = Ah + PL * APLANE;                    // [PL][BN][32]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int g = blockIdx.y;
  const int nb = p.mtiles * p.ntiles;
  int id = blockIdx.x;
  {
    const int q = nb >> 3, r = nb & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = id / p.ntiles;
  const int nt = id - mt * p.ntiles;
  const int p0 = mt * GM_BM;
  const int n0 = nt * BN;
  const float* __restrict__ bg = p.bias + (size_t)g * p.b_gstride;
  const uint16_t* __restrict__ Wg = (PL == 2 ? p.Wh2 : p.W3) + (size_t)g * p.Cout * p.Kpad16;
  // two output tensors (split_n is a multiple of the tile width): this tile's columns belong to one of them
  GemmParams po = p;
  if (EPI == EPI_STORE && p.split_n > 0 && n0 >= p.split_n) {
    po.out = p.out2;
    po.ldc = p.ldc2;
    po.c_coff = -p.split_n;
    po.out_amax = p.out_amax2;
  }

  // activation scale: the power of two that puts the tensor maximum in [2^14, 2^15)
  float amax = PL == 2 ? p.a_amax_floor : 1.f;
  constexpr bool ADD_BOUNDS = LOADER == LOAD_GATHER_ADD || LOADER == LOAD_INTERP_ADD;   // the loader SUMS its inputs
  const int p_hi = min(p0 + GM_BM, p.P) - 1;   // rows of this tile: [p0, p_hi]
  if (PL == 2 && p.a_amax) {
    const float m = amax_rows(p.a_amax, lane, p0, p_hi, p.rps);
    amax = ADD_BOUNDS ? amax + m : fmaxf(amax, m);
  }
  if (PL == 2 && p.a_amax2) {
    const float m = amax_rows(p.a_amax2, lane, p0, p_hi, p.rps);
    amax = ADD_BOUNDS ? amax + m : fmaxf(amax, m);
  }
  uint32_t ex = __float_as_uint(amax) >> 23;
  ex = ex < 15u ? 15u : (ex > 240u ? 240u : ex);
  ex = __builtin_amdgcn_readfirstlane(ex);
  const float sa = PL == 2 ? __uint_as_float((268u - ex) << 23) : 1.f;       // 2^(141 - ex)
  const float inv_sa = PL == 2 ? __uint_as_float((ex - 14u) << 23) : 1.f;   // 2^(ex - 141)

  ALoader<LOADER, RPT, RS> ld;
  ld.init(p, p0, g, t);
  const int chunk = t & 7;    // A: 4-float chunk of the 32-wide K tile
  const int srow = t >> 3;    // A: rows srow + 32 s
  const int wchunk = t & 3;   // W: 8-half chunk
  const int wrow = t >> 2;    // W: rows wrow + 64 s
  bool wok[WPT];
  size_t woff[WPT];
#pragma unroll
  for (int s = 0; s < WPT; ++s) {
    const int n = n0 + wrow + WRS * s;
    wok[s] = n < p.Cout;
    woff[s] = (size_t)(wok[s] ? n : 0) * p.Kpad16 + wchunk * 8;
  }

  f32x16 acc[2][NCB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int fswz = (li >> 2) & 3;   // row + 32 k keeps bits 2..3: one swizzle per lane
  const int a_row = (wr * 64 + li) * GH_STR;
  const int b_row = (wc * 32 * NCB + li) * GH_STR;
  const int f_off[2] = {((0 + lh) ^ fswz) * 8, ((2 + lh) ^ fswz) * 8};   // chunk 2 ks + lh

  float4 ra[RPT];
  uint4 rw[PL][WPT];
  const int ntile_k = (p.Kpad16 + BK - 1) / BK;

  auto gload = [&](int kt) {
    const int k0 = kt * BK + chunk * 4;
    const bool live = k0 < p.Kpad16;
#pragma unroll
    for (int s = 0; s < RPT; ++s) ra[s] = live ? ld.load(p, s, k0, t) : f4zero();
    const bool wlive = kt * BK + wchunk * 8 < p.Kpad16;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int s = 0; s < WPT; ++s)
        rw[pl][s] = (wlive && wok[s])
                        ? *reinterpret_cast<const uint4*>(Wg + pl * p.w3_plane + woff[s] + kt * BK)
                        : make_uint4(0u, 0u, 0u, 0u);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
      uint16_t* dst = Ah + (srow + RS * s) * GH_STR + (((chunk >> 1) ^ ((srow >> 2) & 3)) * 8) + (chunk & 1) * 4;
      if constexpr (PL == 2) {
        uint2 h, l;
        split2_h<LOADER == LOAD_GATHER>(ra[s], sa, h, l);
        *reinterpret_cast<uint2*>(dst) = h;
        *reinterpret_cast<uint2*>(dst + APLANE) = l;
      } else {
        *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pk_bf16(ra[s].x, ra[s].y), cvt_pk_bf16(ra[s].z, ra[s].w));
      }
    }
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int s = 0; s < WPT; ++s)
        *reinterpret_cast<uint4*>(Wh + pl * WPLANE + (wrow + WRS * s) * GH_STR +
                                  ((wchunk ^ ((wrow >> 2) & 3)) * 8)) = rw[pl][s];
  };

#define S4G_H2_TERM(PA, PB)                                                        \
  acc[0][cp] = chain_mfma<PL>(af[0][PA], bf[0][PB], acc[0][cp]);                     \
  acc[0][cp + 1] = chain_mfma<PL>(af[0][PA], bf[1][PB], acc[0][cp + 1]);             \
  acc[1][cp] = chain_mfma<PL>(af[1][PA], bf[0][PB], acc[1][cp]);                     \
  acc[1][cp + 1] = chain_mfma<PL>(af[1][PA], bf[1][PB], acc[1][cp + 1]);
  auto compute_ks = [&](int ks) {
    uint4 af[2][PL];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int pl = 0; pl < PL; ++pl)
        af[rb][pl] = *reinterpret_cast<const uint4*>(Ah + pl * APLANE + a_row + rb * 32 * GH_STR + f_off[ks]);
#pragma unroll
    for (int cp = 0; cp < NCB; cp += 2) {
      uint4 bf[2][PL];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
          bf[cb][pl] = *reinterpret_cast<const uint4*>(Wh + pl * WPLANE + b_row + (cp + cb) * 32 * GH_STR + f_off[ks]);
      // three products per tile, the small cross terms first (one product for a single plane)
      if constexpr (PL == 2) {
        S4G_H2_TERM(0, 1)
        S4G_H2_TERM(1, 0)
      }
      S4G_H2_TERM(0, 0)
    }
  };

  gload(0);
  for (int kt = 0; kt < ntile_k; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < ntile_k) gload(kt + 1);
    const int krem = p.Kpad16 - kt * BK;
    const int nks = krem >= BK ? 2 : 1;
    for (int ks = 0; ks < nks; ++ks) compute_ks(ks);
    __syncthreads();
  }
#undef S4G_H2_TERM

  // undo the scales (exact: powers of two) and leave this tile's maximum for
  // the consumer of the output tensor
  float tmax = 0.f;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int n = n0 + wc * 32 * NCB + cb * 32 + li;
    const bool nok = n < p.Cout;
    const float sc = nok ? (PL == 2 ? inv_sa * p.w_inv_scale[(size_t)g * p.b_gstride + n] : 1.f) : 0.f;
    const float bias = nok ? bg[n] : 0.f;
    float mx = -__builtin_inff(), mn = __builtin_inff();
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[rb][cb][r] * sc;
        acc[rb][cb][r] = v;
        mx = fmaxf(mx, v);
        mn = fminf(mn, v);
      }
    const float hi = mx + bias, lo = mn + bias;
    tmax = fmaxf(tmax, p.relu ? hi : fmaxf(fabsf(hi), fabsf(lo)));
  }
  if (PL == 2 && po.out_amax) {
    const uint32_t wm = wave_max_u32(__float_as_uint(fmaxf(tmax, 0.f)));
    if (lane == 0) amax_publish(po.out_amax, wm, blockIdx.x * 4 + wave, p0, p_hi, p.rps);
  }
  gemm_epilogue<EPI, NCB>(po, acc, bg, g, p0, n0, wave, wr, wc, li, lh, smemf);
}

template <int LOADER, int EPI, int NCB, int PL>
static int launch_gemm_f16x2_cfg(GemmParams p, int groups, hipStream_t st) {
  constexpr int BN = 64 * NCB;
  p.ntiles = (p.Cout + BN - 1) / BN;
  size_t lds = sizeof(uint16_t) * PL * (GM_BM + BN) * GH_STR;
  const size_t epi = sizeof(float) * 4 * 32 * (32 * NCB + 4);   // staged epilogue stores
  if (lds < epi) lds = epi;
  static LdsAttrCache lds_cache;
  if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(&mlp_gemm_f16x2_kernel<LOADER, EPI, NCB, PL>), lds, lds_cache)) return rc;
  const dim3 grid((unsigned)(p.mtiles * p.ntiles), (unsigned)groups);
  hipLaunchKernelGGL((mlp_gemm_f16x2_kernel<LOADER, EPI, NCB, PL>), grid, dim3(256), lds, st, p);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

#ifdef S4G_VARIANTS
#define S4G_VARIANT_PART 1
#include "variants/gemm_resident.inc"
#undef S4G_VARIANT_PART
#endif

// ---------------------------------------------------------------------------
// f16x2, two fused layers (the last two layers of an SA level whose widths allow it:
// K = 128 -> 128 -> Cout2, max over the 64 neighbours): the resident-A kernel run twice
// on ONE LDS panel.  A workgroup owns 128 positions (two centroids).  Phase 1 contracts
// the loader's panel with W (operands swapped: a lane ends up with 4 consecutive channels
// of one position), applies scale / bias / ReLU, finds the TILE maximum (one barrier --
// which is also the point where nobody reads the old panel any more), splits the
// activations with the tile's own power-of-two scale and writes them over the panel.
// Phase 2 is the resident kernel's strip loop on that panel with the max epilogue.  The
// 128-channel intermediate never leaves the CU: no 2.7 GB store + load + re-split, no
// amax round trip through HBM (a per-tile scale is as exact as the per-tensor one: both
// are powers of two undone in the epilogue).
// ---------------------------------------------------------------------------
// measurement builds only (make HIPFLAGS_EXTRA=-DS4G_CHAIN_ABLATE=bits, tools/ablate_kernels.sh): the chain
// kernel with parts REMOVED, to see what each costs -- 1 no W refill, 2 no LDS operand reads in
// the strips, 4 no panel epilogue, 8 no final epilogue, 16 no loader.  Results are garbage.
__device__ __forceinline__ void chain_keep_alive(const f32x16& v) { asm volatile("" ::"v"(v)); }

#ifdef S4G_CHAIN_STAMPS
// debug build only (make HIPFLAGS_EXTRA=-DS4G_CHAIN_STAMPS): s_memtime stamps of wave 0 of the
// 512 workgroups from the middle of the grid at the phase boundaries of mlp_chain_kernel, read back by tools/chain_stamps.py
__device__ unsigned long long g_chain_stamps[512 * 16];
#define S4G_STAMP(i)                                                                       \
  do {                                                                                     \
    const unsigned sb_ = gridDim.x >= 1024 ? blockIdx.x - gridDim.x / 2 : blockIdx.x;   /* steady state: the middle of the grid */ \
    if (threadIdx.x == 0 && sb_ < 512u && blockIdx.y == 0)                                 \
      g_chain_stamps[sb_ * 16 + (i)] = __builtin_amdgcn_s_memtime();                      \
  } while (0)
#else
#define S4G_STAMP(i)
#endif
constexpr int GF_RING_F16X2 = 2;
// ring depth of the single-plane (bf16) form: 4 measured no faster than 2 (configs[4]: sa0
// 2.67 vs 2.57 ms), so the W stream's latency is not what parks its waves
// W fragment prefetch depth of the chain kernel: 2 measured equal to 4 (+0.5 %) with 32 registers
// less -- no spills in the deep-first-layer and eight-wave forms
#ifndef S4G_CHAIN_RING1
#define S4G_CHAIN_RING1 2
#endif
// 32-row blocks a wave of the single-plane (bf16) form owns: 4 (round 4: a W fragment then feeds four MFMAs
// instead of two -- with ONE product per MAC the W stream is the form's bottleneck) or 2 (rounds 2-3)
#ifndef S4G_CHAIN_BF16_NRB
#define S4G_CHAIN_BF16_NRB 4
#endif
// (The f16x2 form stays at two: four blocks there mean a 133 KB panel and ONE workgroup per CU at 128- and 256-wide
// layers, and losing the second workgroup's phase overlap costs more than the halved W stream returns -- measured in
// round 4, one box, in step: sa0 chain 1.15 -> 1.64 ms, sa1 chain 1.07 -> 1.20 ms, the chain-form single layers
// 0.57 -> 0.67 ms, 1 988 -> 1 863 scenes/s.)

// PL = planes per operand: 2 = the f16x2 split above (three fp16 products per step, power-of-two
// scales), 1 = ONE bf16 plane and one product (S4G_GEMM_BF16, the reduced-precision roofline
// configuration: bf16 has fp32's exponent range, so there are no scales, no maxima and no
// barrier for them; the panel is half as large, so twice as many workgroups fit a CU).
template <int LOADER, int EPI2, int RW, int KC, int PL>
__global__ __launch_bounds__(RW == 8 ? 512 : 256, (PL == 1 && RW != 8 && S4G_CHAIN_BF16_NRB == 2) ? 3 : 2) void mlp_chain_kernel(const GemmParams p) {
  // RW = 2: 128 positions, 128-wide layers, 4 waves; RW = 1: 64 positions, 256-wide layers, 4 waves;
  // RW = 8: 64 positions, 512-wide layers, EIGHT waves (one workgroup per CU: its 133 KB panel).
  // A wave owns NRB 32-row blocks x 64 channels: 2 blocks in the f16x2 form (the positions above), 4 in
  // the single-plane form (twice the positions: the same LDS bytes, half the W bytes per MFMA).
  constexpr int GF_RING = PL == 2 ? GF_RING_F16X2 : S4G_CHAIN_RING1;
  constexpr int NRB = PL == 1 ? S4G_CHAIN_BF16_NRB : 2, WROWS = 32 * NRB;
  constexpr int CW = RW == 8 ? 8 : 4 / RW, RWN = RW == 8 ? 1 : RW;
  constexpr int BM = WROWS * RWN, K = 64 * CW, NW = RWN * CW;
  constexpr int RS = 8 * NW, RPT = BM / RS;   // loader: 8 lanes per row, RS rows per pass
  constexpr int astr = K + 8, aplane = BM * astr, KS = K >> 4;
  extern __shared__ __attribute__((aligned(16))) float smemf[];
  uint16_t* Ah = reinterpret_cast<uint16_t*>(smemf);   // [PL][BM][K + 8]
  float* scr = reinterpret_cast<float*>(Ah + PL * aplane);   // [NW waves][128] scale | bias, then NW tile maxima
  const int g = blockIdx.y;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;

  constexpr bool ADD_BOUNDS = LOADER == LOAD_GATHER_ADD || LOADER == LOAD_INTERP_ADD;   // the loader SUMS its inputs
  constexpr bool MFMA0 = LOADER == LOAD_REL_MLP1;   // the xyz-only first layer as one 16-deep MFMA step (f16x2 form only)
  static_assert(!MFMA0 || (PL == 2 && KC == 1), "phase 0 on the matrix cores: f16x2, one panel");
  constexpr bool SEGMAX = (LOADER == LOAD_GATHER_MLP1 || MFMA0) && EPI2 == EPI_MAX;   // may run on distinct rows only (seg4)
  const int tile = blockIdx.x;
  const int p0 = tile * BM;
  bool seg_scene = false;   // this tile's scene is in the distinct-row layout (a mostly-full scene keeps the plain one)
  if constexpr (SEGMAX) {
    // distinct-row form: a scene's rows end before its base + rps; the tiles behind them have nothing to do
    if (p.seg_rows) {
      const int sc = p0 / p.rps;
      const int nrows = p.seg_rows[sc];
      if (p0 - sc * p.rps >= nrows) return;
      seg_scene = nrows < p.rps;
    }
  }
  const int p_hi = min(p0 + BM, p.P) - 1;   // rows of this tile: [p0, p_hi]
  // The input maxima of this tile's scene are REQUESTED here and reduced after the W prefetch and
  // the loader's own loads have been issued (vector loads return in order: asked for first, they
  // cost no round trip of their own; reduced first, each was one before anything else started).
  const int am_s0 = p.rps > 0 ? p0 / p.rps : 0, am_s1 = p.rps > 0 ? p_hi / p.rps : 0;
  float am_v1 = 0.f, am_v2 = 0.f;
  if (PL == 2 && p.a_amax) am_v1 = p.a_amax[(size_t)am_s0 * 64 + lane];
  if (PL == 2 && p.a_amax2) am_v2 = p.a_amax2[(size_t)am_s0 * 64 + lane];

  const int wr = wave / CW, wc = wave % CW;
  const int li = lane & 31, lh = lane >> 5;
  const int wc_u = __builtin_amdgcn_readfirstlane(wc);
  const uint32_t wf_lane = (uint32_t)lane * 16u;
  constexpr size_t cb_stride = (size_t)KS * PL * 1024;          // bytes between n32 and n32 + 1
  constexpr size_t strip_stride = (size_t)CW * 2 * cb_stride;   // bytes between 128-channel strips
  // layer 1 may contract over kchunks * K inputs: its panel is loaded K columns at a time and
  // the accumulators run through all chunks before the first epilogue
  constexpr int kchunks = KC;   // Kpad16 / K
  const size_t cbs1 = (size_t)kchunks * cb_stride;              // layer 1's bytes between n32 blocks
  // W streams go through buffer loads: descriptor and block offset are scalars, the lane offset
  // one constant VGPR, so a fragment load costs no 64-bit address arithmetic on the vector pipe
  const WRef w1 = wref(p.Wfrag + (size_t)g * p.Cout * p.Kpad16 * PL, (size_t)(wc_u * 2) * cbs1);
  // chain: layer 1 -> [layer 2 when a third layer follows] -> final layer (2 or 3)
  const bool tri = p.Wfrag3 != nullptr;
  const WRef wmid = wref(p.Wfrag2 + (size_t)g * p.Cout2 * K * PL, (size_t)(wc_u * 2) * cb_stride);
  const int CoutF = tri ? p.Cout3 : p.Cout2;
  const float* __restrict__ scF = PL == 2 ? (tri ? p.w_inv_scale3 : p.w_inv_scale2) + (size_t)g * CoutF : nullptr;
  const float* __restrict__ bgF = (tri ? p.bias3 : p.bias2) + (size_t)g * CoutF;
  const WRef w2 = tri ? wref(p.Wfrag3 + (size_t)g * CoutF * K * PL, (size_t)(wc_u * 2) * cb_stride) : wmid;
  const int nstrip2 = (CoutF + 64 * CW - 1) / (64 * CW);
  // a wave whose 64 channels lie past the final Cout (last, partial strip) sits the final phase
  // out; its ring must not prefetch fragments that do not exist
  const bool active0 = wc_u * 64 < CoutF;

  uint4 ring[GF_RING][2][PL];
#pragma unroll
  for (int d = 0; d < GF_RING; ++d)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int pl = 0; pl < PL; ++pl)
        ring[d][cb][pl] = wref_load(w1, wf_lane + pl * 1024, (uint32_t)(cb * cbs1 + (size_t)d * PL * 1024));

  // (the loader's per-row state stays live through the first layer only when that layer is
  // more than one panel deep: KC > 1)
  S4G_STAMP(0);
  ALoader<LOADER, RPT, RS> ld;
  // phase 0 on the matrix cores: this wave's rows' (xyz_j - ctr_m, 0) records and this lane's channel of the first
  // layer, requested before anything waits (buffer loads: a row past P reads zeros)
  float4 rv0[NRB];
  float4 wv0 = f4zero();
  if constexpr (MFMA0) {
    const int rows_left = p.P - p0;
    const __amdgpu_buffer_rsrc_t rrel = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4*>(p.rel4 + p0), 0, (rows_left < BM ? rows_left : BM) * 16, 0x00020000);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
      rv0[rb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                               rrel, (wr * WROWS + rb * 32 + li) * 16, 0, 0));
    const int ch = wc * 64 + lane;
    if (ch < p.Cin) wv0 = p.mlp1[ch];
  } else {
    ld.init(p, p0, g, t);
  }
  S4G_STAMP(1);
  // activation scale of the loader's rows (per scene; a tile that straddles scenes joins their rows)
  float amax = PL == 2 ? p.a_amax_floor : 1.f;
  if (PL == 2 && p.a_amax) {
    float m = __uint_as_float(wave_max_u32(__float_as_uint(am_v1)));
    for (int sc = am_s0 + 1; sc <= am_s1; ++sc) m = fmaxf(m, amax_slots(p.a_amax + (size_t)sc * 64, lane));
    amax = ADD_BOUNDS ? amax + m : fmaxf(amax, m);
  }
  if (PL == 2 && p.a_amax2) {
    float m = __uint_as_float(wave_max_u32(__float_as_uint(am_v2)));
    for (int sc = am_s0 + 1; sc <= am_s1; ++sc) m = fmaxf(m, amax_slots(p.a_amax2 + (size_t)sc * 64, lane));
    amax = ADD_BOUNDS ? amax + m : fmaxf(amax, m);
  }
  uint32_t ex = __float_as_uint(amax) >> 23;
  ex = ex < 15u ? 15u : (ex > 240u ? 240u : ex);
  ex = __builtin_amdgcn_readfirstlane(ex);
  const float sa = PL == 2 ? __uint_as_float((268u - ex) << 23) : 1.f;
  const float inv_sa = PL == 2 ? __uint_as_float((ex - 14u) << 23) : 1.f;
  // columns [kc * K, kc * K + K) of the loader's rows -> two fp16 planes.  DEPTH K-tiles of
  // loads are in flight at a time: the whole panel in the prologue (one round trip), two
  // tiles for the later chunks of a deep first layer (accumulators and ring are live then)
  auto load_panel = [&](int kc, auto depth_tag) {
    constexpr int DEPTH = decltype(depth_tag)::value;
    const int chunk = t & 7, srow = t >> 3;
    constexpr int NKT = K / 32;
#pragma unroll
    for (int kt0 = 0; kt0 < NKT; kt0 += DEPTH) {
      float4 ra[DEPTH][RPT];
#pragma unroll
      for (int kt = 0; kt < DEPTH; ++kt)
#pragma unroll
        for (int s = 0; s < RPT; ++s) ra[kt][s] = ld.load(p, s, kc * K + (kt0 + kt) * 32 + chunk * 4, t);
#pragma unroll
      for (int kt = 0; kt < DEPTH; ++kt)
#pragma unroll
        for (int s = 0; s < RPT; ++s) {
          uint16_t* dst = Ah + (srow + RS * s) * astr + (kt0 + kt) * 32 + chunk * 4;
          if constexpr (PL == 2) {
            uint2 h, l;
            split2_h<false>(ra[kt][s], sa, h, l);
            *reinterpret_cast<uint2*>(dst) = h;
            *reinterpret_cast<uint2*>(dst + aplane) = l;
          } else {
            *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pk_bf16(ra[kt][s].x, ra[kt][s].y),
                                                        cvt_pk_bf16(ra[kt][s].z, ra[kt][s].w));
          }
        }
      if (DEPTH < NKT) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // (64 loaded values per thread in flight at most: one round trip for the 2-block form, two for the 4-block one)
  const uint16_t* a_lane = Ah + (wr * WROWS + li) * astr + 8 * lh;
  float* epi_s = scr + wave * 128;
  f32x16 acc[2][NRB];   // [32-channel block][32-row block], in every phase
  if constexpr (MFMA0) {
    // ---- phase 0: H0 = relu(W1 (rel, 1)) for this wave's WROWS rows x 64 channels as ONE 16-deep step of the
    // f16x2 contraction (12 MFMAs) instead of 3 FMAs + ReLU + select per element on the vector ALU.
    //   position operand: lane (li, lh) holds (rx, ry, rz, ONE) s_a at k = 8 lh .. 8 lh + 3 -- BOTH half-waves carry
    //     the row's record, so the k = 0..3 and k = 8..11 slots are copies; ONE = the power of two above the wave's
    //     largest |coordinate| (measured here: no bound is assumed), s_a ONE = 2^15
    //   channel operand: lane = channel wc 64 + lane = block lh, column li: (wx, wy, wz, b / ONE) s_w at ITS
    //     half's k slots and zeros in the other block's, so block cb's fragment is "mine where lh == cb, else 0" --
    //     no lane exchange; s_w = the channel's own power-of-two scale (its largest entry in [2^14, 2^15))
    // acc = s_a s_w (w . rel + b); the epilogue applies relu, then splits with f = sa / (s_a s_w) per channel.
    float m = 0.f;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(rv0[rb].x), fabsf(rv0[rb].y)), fabsf(rv0[rb].z)));
    uint32_t ea = wave_max_u32(__float_as_uint(m)) >> 23;
    ea = ea < 90u ? 90u : (ea > 240u ? 240u : ea);      // (ONE >= 2^-36: coordinates below that are zero to fp32 sums)
    ea = __builtin_amdgcn_readfirstlane(ea);
    const float s_a = __uint_as_float((268u - ea) << 23), inv_a = __uint_as_float((ea - 14u) << 23);
    const float one = __uint_as_float((ea + 1u) << 23), inv_one = __uint_as_float((253u - ea) << 23);
    const float bq = wv0.w * inv_one;
    const float mw = fmaxf(fmaxf(fabsf(wv0.x), fabsf(wv0.y)), fmaxf(fabsf(wv0.z), fabsf(bq)));
    uint32_t ew = __float_as_uint(mw) >> 23;
    ew = ew < 15u ? 15u : (ew > 240u ? 240u : ew);
    const float s_w = __uint_as_float((268u - ew) << 23);
    epi_s[lane] = __uint_as_float((ew - 14u) << 23) * inv_a * sa;
    uint32_t wh01, wl01, wh23, wl23;
    split_quad_h(wv0.x, wv0.y, wv0.z, bq, s_w, wh01, wl01, wh23, wl23);
    uint4 af0[NRB][2], bf0[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      bf0[cb][0] = make_uint4(lh == cb ? wh01 : 0u, lh == cb ? wh23 : 0u, 0u, 0u);
      bf0[cb][1] = make_uint4(lh == cb ? wl01 : 0u, lh == cb ? wl23 : 0u, 0u, 0u);
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const float one_r = p0 + wr * WROWS + rb * 32 + li < p.P ? one : 0.f;
      uint32_t h01, l01, h23, l23;
      split_quad_h(rv0[rb].x, rv0[rb].y, rv0[rb].z, one_r, s_a, h01, l01, h23, l23);
      af0[rb][0] = make_uint4(h01, h23, 0u, 0u);
      af0[rb][1] = make_uint4(l01, l23, 0u, 0u);
    }
    f32x16 z16;
#pragma unroll
    for (int r = 0; r < 16; ++r) z16[r] = 0.f;
    // The fragments above come out of INLINE ASM (v_fma_mix*): the compiler's hazard recognizer does not see an asm
    // statement as a vector-ALU write, so it does not pad the VALU-write -> MFMA-read distance -- the first MFMA read
    // the ONE slot's half before the v_fma_mixhi two instructions earlier had landed (found by the test: the bias
    // term's low plane missing in one row block).  Everywhere else the split's results go through LDS first.
    // The wait is TIED to the registers it protects: the s_nop's asm statement takes every asm-produced fragment word as
    // an in/out operand, so the splits must be scheduled before it and the MFMAs (which read its outputs) after it -- by
    // data dependence, not by where the statement happens to sit between two scheduling barriers.
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
      asm volatile("s_nop 7" : "+v"(af0[rb][0].x), "+v"(af0[rb][0].y), "+v"(af0[rb][1].x), "+v"(af0[rb][1].y));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        acc[cb][rb] = chain_mfma<PL>(bf0[cb][1], af0[rb][0], z16);
        acc[cb][rb] = chain_mfma<PL>(bf0[cb][0], af0[rb][1], acc[cb][rb]);
        acc[cb][rb] = chain_mfma<PL>(bf0[cb][0], af0[rb][0], acc[cb][rb]);
      }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 f4 = *reinterpret_cast<const float4*>(epi_s + nb * 32 + 8 * j + 4 * lh);
#pragma unroll
        for (int pb = 0; pb < NRB; ++pb) {
          uint16_t* dst = Ah + (wr * WROWS + pb * 32 + li) * astr + wc * 64 + nb * 32 + 8 * j + 4 * lh;
          uint2 h, l;
          split_quad_h2(fmaxf(acc[nb][pb][4 * j], 0.f), fmaxf(acc[nb][pb][4 * j + 1], 0.f),
                        fmaxf(acc[nb][pb][4 * j + 2], 0.f), fmaxf(acc[nb][pb][4 * j + 3], 0.f), f4.x, f4.y, f4.z, f4.w,
                        h.x, l.x, h.y, l.y);
          *reinterpret_cast<uint2*>(dst) = h;
          *reinterpret_cast<uint2*>(dst + aplane) = l;
        }
      }
  } else {
    if (!(S4G_CHAIN_ABLATE & 16)) load_panel(0, std::integral_constant<int, (K / 32) / (NRB / 2)>{});
  }
  S4G_STAMP(2);
  __syncthreads();
  S4G_STAMP(3);

  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NRB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  uint4 afn[NRB][PL];
  auto prime_a = [&]() {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int pl = 0; pl < PL; ++pl)
        afn[rb][pl] = *reinterpret_cast<const uint4*>(a_lane + pl * aplane + rb * 32 * astr);
  };

  // one 128-deep strip: 8 steps of 12 MFMAs; W fragments through the ring, refilled RING
  // steps ahead from this strip (wcur) or the next one (wnext)
#define S4G_F2_STRIP(SWAPPED, ZFIRST, wcur, cbs_cur, wnext, cbs_next)                                                     \
  _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                  \
    const int d = ks % GF_RING;                                                                        \
    const int ksn = ks + 1 == KS ? 0 : ks + 1;                                                         \
    uint4 af[NRB][PL], bf[2][PL];                                                                      \
    _Pragma("unroll") for (int rb = 0; rb < NRB; ++rb) _Pragma("unroll") for (int pl = 0; pl < PL; ++pl) { \
      af[rb][pl] = afn[rb][pl];                                                                        \
      if (!(S4G_CHAIN_ABLATE & 2))                                                                     \
        afn[rb][pl] = *reinterpret_cast<const uint4*>(a_lane + pl * aplane + rb * 32 * astr + ksn * 16); \
    }                                                                                                  \
    _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) _Pragma("unroll") for (int pl = 0; pl < PL; ++pl)  \
      bf[cb][pl] = ring[d][cb][pl];                                                                    \
    {                                                                                                  \
      const int kr = ks + GF_RING;                                                                     \
      const WRef src = kr < KS ? (wcur) : (wnext);                                                     \
      const size_t cbs = kr < KS ? (cbs_cur) : (cbs_next);                                             \
      const int kk = kr < KS ? kr : kr - KS;                                                           \
      _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) _Pragma("unroll") for (int pl = 0; pl < PL; ++pl) \
        if (!(S4G_CHAIN_ABLATE & 1))                                                                   \
          ring[d][cb][pl] = wref_load(src, wf_lane + pl * 1024, (uint32_t)(cb * cbs + (size_t)kk * PL * 1024)); \
    }                                                                                                  \
    if constexpr (PL == 2) {                                                                           \
      S4G_F2_TERM(SWAPPED, 0, 1, (ZFIRST) && ks == 0)                                                  \
      S4G_F2_TERM(SWAPPED, 1, 0, false)                                                                \
      S4G_F2_TERM(SWAPPED, 0, 0, false)                                                                \
    } else {                                                                                           \
      S4G_F2_TERM(SWAPPED, 0, 0, (ZFIRST) && ks == 0)                                                  \
    }                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 2 * PL; ++q) {                                               \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                               \
    }                                                                                                  \
    if constexpr (PL == 2) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                          \
    if constexpr (PL == 1 && NRB == 4) {   /* 8 MFMAs, 4 LDS reads, 2 fragment loads per step */          \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                             \
      }                                                                                                \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  }
// Z: start from a literal zero accumulator (the MFMA's C operand as an inline constant: no
// register zeroing); only ever true in a fully unrolled first step
#define S4G_F2_TERM(SWAPPED, PA, PB, Z)                                                                \
  _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) _Pragma("unroll") for (int rb = 0; rb < NRB; ++rb)  \
    if constexpr (SWAPPED)                                                                             \
      acc[cb][rb] = chain_mfma<PL>(bf[cb][PB], af[rb][PA], (Z) ? zero16 : acc[cb][rb]);                \
    else                                                                                               \
      acc[cb][rb] = chain_mfma<PL>(af[rb][PA], bf[cb][PB], (Z) ? zero16 : acc[cb][rb]);
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

  // ---- panel phases: H = relu(bn(W A)), C channels = one strip, operands swapped
  WRef wcur = w1;
  const float* __restrict__ scp = PL == 2 ? p.w_inv_scale + (size_t)g * p.b_gstride : nullptr;
  const float* __restrict__ bp = p.bias + (size_t)g * p.b_gstride;
  int relu_ph = p.relu;
  float inv_in = inv_sa, inv_sh = 1.f;
  const int npanel = tri ? 2 : 1;
  for (int ph = 0; ph < npanel; ++ph) {
  const WRef wnxt = ph + 1 < npanel ? wmid : (active0 ? w2 : wcur);
  prime_a();
  {
    const int n = wc * 64 + lane;   // channel whose scale / bias this lane stages for its wave
    epi_s[lane] = PL == 2 ? inv_in * scp[n] : 1.f;
    epi_s[64 + lane] = bp[n];
  }
  if (KC > 1 && ph == 0) {
    // layer 1: every K-column chunk of the loader's rows through the same accumulators
    zero_acc();
    for (int kc = 0; kc < kchunks; ++kc) {
      if (kc > 0) {
        __syncthreads();      // everybody is done reading the previous chunk's panel
        load_panel(kc, std::integral_constant<int, 2>{});
        __syncthreads();
        prime_a();
      }
      const WRef wc1 = w1 + (size_t)kc * cb_stride;           // this chunk's 16 steps of each n32 block
      const bool lastc = kc + 1 == kchunks;
      const WRef wn1 = lastc ? wnxt : wc1 + cb_stride;
      const size_t cbsn = lastc ? cb_stride : cbs1;
      S4G_F2_STRIP(true, false, wc1, cbs1, wn1, cbsn)
    }
  } else {
    S4G_F2_STRIP(true, true, wcur, cb_stride, wnxt, cb_stride)
  }
  S4G_STAMP(4 + 4 * ph);
  if (S4G_CHAIN_ABLATE & 4) {
    chain_keep_alive(acc[0][0]); chain_keep_alive(acc[0][1]); chain_keep_alive(acc[1][0]); chain_keep_alive(acc[1][1]);
    __syncthreads();
    __syncthreads();
  } else {
  float tmax = 0.f;
  const float relu_lo = relu_ph ? 0.f : -__builtin_inff();
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 sc4 = *reinterpret_cast<const float4*>(epi_s + nb * 32 + 8 * j + 4 * lh);
      const float4 b4 = *reinterpret_cast<const float4*>(epi_s + 64 + nb * 32 + 8 * j + 4 * lh);
      const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
      const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int pb = 0; pb < NRB; ++pb) {
        // two v_pk_fma_f32 per four values (each half rounded once, as __fmaf_rn), the tile maximum as max3s
        const f32x2 x01 = pk_fma(f32x2{acc[nb][pb][4 * j], acc[nb][pb][4 * j + 1]}, f32x2{scv[0], scv[1]}, f32x2{bv[0], bv[1]});
        const f32x2 x23 = pk_fma(f32x2{acc[nb][pb][4 * j + 2], acc[nb][pb][4 * j + 3]}, f32x2{scv[2], scv[3]}, f32x2{bv[2], bv[3]});
        // (ReLU as a max with 0 or -inf: a per-element branch on the runtime flag became select chains)
        const float x[4] = {fmaxf(x01.x, relu_lo), fmaxf(x01.y, relu_lo), fmaxf(x23.x, relu_lo), fmaxf(x23.y, relu_lo)};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nb][pb][4 * j + e] = x[e];
        if constexpr (PL == 2) {
          tmax = fmaxf(fmaxf(tmax, fabsf(x[0])), fabsf(x[1]));
          tmax = fmaxf(fmaxf(tmax, fabsf(x[2])), fabsf(x[3]));
        }
      }
    }
  if constexpr (LOADER == LOAD_PLAIN && EPI2 == EPI_STORE) {
    if (!p.Wfrag2) {
      // ONE layer (the plain single-layer launches of a step run here instead of on the tiled kernel where
      // that is faster: the A panel is staged K columns at a time, W streams through the register ring, two
      // barriers per K-wide chunk instead of two per 32 columns): the activated values above ARE the output --
      // this workgroup's positions x this group's K channels (the caller maps 256-channel strips of a wider
      // layer to groups that share A).  Operands are swapped: a lane holds 4 consecutive channels of a position.
      // (two output tensors: this group's 256 channels belong to one of them)
      const bool second = p.split_n > 0 && g * p.c_gcol >= p.split_n;
      float* __restrict__ obase = second ? p.out2 - p.split_n : p.out + p.c_coff;
      const int old = second ? p.ldc2 : p.ldc;
      uint32_t* __restrict__ oamax = second ? p.out_amax2 : p.out_amax;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nn = wc * 64 + nb * 32 + 8 * j + 4 * lh;
#pragma unroll
          for (int pb = 0; pb < NRB; ++pb) {
            const int row = p0 + wr * WROWS + pb * 32 + li;
            if (row < p.P)
              *reinterpret_cast<float4*>(obase + (size_t)row * old + g * p.c_gcol + nn) =
                  make_float4(acc[nb][pb][4 * j], acc[nb][pb][4 * j + 1], acc[nb][pb][4 * j + 2], acc[nb][pb][4 * j + 3]);
          }
        }
      if (PL == 2 && oamax) {
        const uint32_t wm = wave_max_u32(__float_as_uint(tmax));
        if (lane == 0) amax_publish(oamax, wm, (tile * 4 + wave) * 5 + g, p0, p_hi, p.rps);
      }
      return;
    }
  }
  float sh = 1.f;
  if constexpr (PL == 2) {
    const uint32_t wm = wave_max_u32(__float_as_uint(tmax));
    if (lane == 0) scr[NW * 128 + wave] = __uint_as_float(wm);
  }
  S4G_STAMP(5 + 4 * ph);
  __syncthreads();   // every wave is done with the old panel; the four maxima are visible
  S4G_STAMP(6 + 4 * ph);
  if constexpr (PL == 2) {
    float hmax = scr[NW * 128];
#pragma unroll
    for (int w = 1; w < NW; ++w) hmax = fmaxf(hmax, scr[NW * 128 + w]);
    uint32_t exh = __float_as_uint(hmax) >> 23;
    exh = exh < 15u ? 15u : (exh > 240u ? 240u : exh);
    exh = __builtin_amdgcn_readfirstlane(exh);
    sh = __uint_as_float((268u - exh) << 23);
    inv_sh = __uint_as_float((exh - 14u) << 23);
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int pb = 0; pb < NRB; ++pb) {
        uint16_t* dst = Ah + (wr * WROWS + pb * 32 + li) * astr + wc * 64 + nb * 32 + 8 * j + 4 * lh;
        if constexpr (PL == 2) {
          const float4 v = make_float4(acc[nb][pb][4 * j], acc[nb][pb][4 * j + 1], acc[nb][pb][4 * j + 2],
                                       acc[nb][pb][4 * j + 3]);
          uint2 h, l;
          split2_h<false>(v, sh, h, l);
          *reinterpret_cast<uint2*>(dst) = h;
          *reinterpret_cast<uint2*>(dst + aplane) = l;
        } else {
          *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pk_bf16(acc[nb][pb][4 * j], acc[nb][pb][4 * j + 1]),
                                                      cvt_pk_bf16(acc[nb][pb][4 * j + 2], acc[nb][pb][4 * j + 3]));
        }
      }
  S4G_STAMP(7 + 4 * ph);
  __syncthreads();
  }
  // the next panel phase (three-layer chains) reads this panel through layer 2's weights
  inv_in = inv_sh;
  wcur = wmid;
  scp = PL == 2 ? p.w_inv_scale2 + (size_t)g * p.Cout2 : nullptr;
  bp = p.bias2 + (size_t)g * p.Cout2;
  relu_ph = p.relu2;
  }

  // ---- phase 2: the strip loop on the new panel, max over the 64 rows of a centroid
  GemmParams q = p;
  q.Cout = CoutF;
  q.relu = tri ? p.relu3 : p.relu2;
  prime_a();
  WRef wstrip = w2;
  const float* __restrict__ bg2 = bgF;
  // distinct-row form: the output rows of this half-wave's eight 4-row groups per 64 rows (rows WROWS wr + 64 c + 32 lh + 4 i ..)
  int sidv[NRB / 2][8] = {};
  if constexpr (SEGMAX) {
    if (seg_scene) {
#pragma unroll
      for (int c = 0; c < NRB / 2; ++c) {
        const int4* sp = reinterpret_cast<const int4*>(p.seg4 + ((p0 + wr * WROWS + c * 64 + lh * 32) >> 2));
        const int4 s0 = sp[0], s1 = sp[1];
        sidv[c][0] = s0.x; sidv[c][1] = s0.y; sidv[c][2] = s0.z; sidv[c][3] = s0.w;
        sidv[c][4] = s1.x; sidv[c][5] = s1.y; sidv[c][6] = s1.z; sidv[c][7] = s1.w;
      }
    }
  }
  for (int strip = 0; strip < nstrip2; ++strip, wstrip = wstrip + strip_stride) {
    if ((strip * CW + wc_u) * 64 >= CoutF) break;
    const int n = (strip * CW + wc) * 64 + lane;
    const float e_sc = PL == 2 ? inv_sh * scF[n] : 1.f;
    const float e_bias = bg2[n];
    const WRef wnext = ((strip + 1) * CW + wc_u) * 64 < CoutF ? wstrip + strip_stride : wstrip;
    S4G_STAMP(12 + 2 * (strip & 1));
    S4G_F2_STRIP(EPI2 == EPI_STORE, true, wstrip, cb_stride, wnext, cb_stride)
    S4G_STAMP(13 + 2 * (strip & 1));
    if (S4G_CHAIN_ABLATE & 8) {
      chain_keep_alive(acc[0][0]); chain_keep_alive(acc[0][1]); chain_keep_alive(acc[1][0]); chain_keep_alive(acc[1][1]);
      continue;
    }
    const int n0 = (strip * CW + wc) * 64;
    float omax = 0.f;
    if constexpr (EPI2 == EPI_STORE) {
      // operands swapped: registers 4 j .. 4 j + 3 = channels n0 + 32 nb + 8 j + 4 lh + (0..3)
      epi_s[lane] = e_sc;
      epi_s[64 + lane] = e_bias;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nn = n0 + nb * 32 + 8 * j + 4 * lh;
          const float4 sc4 = *reinterpret_cast<const float4*>(epi_s + nb * 32 + 8 * j + 4 * lh);
          const float4 b4 = *reinterpret_cast<const float4*>(epi_s + 64 + nb * 32 + 8 * j + 4 * lh);
          const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
          const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int pb = 0; pb < NRB; ++pb) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = __fmaf_rn(acc[nb][pb][4 * j + e], scv[e], bv[e]);
              if (q.relu) x = fmaxf(x, 0.f);
              v[e] = x;
              omax = fmaxf(omax, fabsf(x));
            }
            const int row = p0 + wr * WROWS + pb * 32 + li;
            if (row < p.P)
              *reinterpret_cast<float4*>(p.out + (size_t)row * p.ldc + p.c_coff + g * p.c_gcol + nn) =
                  make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      if (p.out_amax) {
        const uint32_t wm = wave_max_u32(__float_as_uint(omax));
        if (lane == 0) amax_publish(p.out_amax, wm, tile * 4 + wave + strip, p0, p_hi, p.rps);
      }
    } else if (SEGMAX && seg_scene) {
      // distinct-row form: the wave's 64 rows are pieces of several centroids, each a run of 4-row groups
      // with one output row (seg4).  Per channel block: the group maxima (4 -> 1 in registers), one
      // half-exchange per pair (v_permlane32_swap: afterwards the lower half-wave holds the 8 groups of
      // row block 0 for its channel, the upper half those of block 1), then every half walks ITS eight
      // groups; where the output row changes the run's maximum is scaled, biased, clamped (post-ReLU
      // values are >= 0: their bit patterns order like unsigned integers) and merged into the
      // zero-initialised output with atomicMax -- a centroid's rows may continue in the other half, the
      // next wave's rows or the next tile.  max is exact and order-free: same values as the 64-row form.
      uint32_t* __restrict__ outu = reinterpret_cast<uint32_t*>(p.out);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const float sc = __shfl(e_sc, cb * 32 + li);
        const float bias = __shfl(e_bias, cb * 32 + li);
        const int nn = n0 + cb * 32 + li;
#pragma unroll
        for (int c = 0; c < NRB / 2; ++c) {      // every 64 rows of the wave: two 32-row blocks
        float v[8];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          // rows 32 rb + 8 qd + 4 lh + (0..3) = group 8 rb + 2 qd + lh of the 64 rows' 16
          const float o0 = fmaxf(fmaxf(acc[cb][2 * c][4 * qd], acc[cb][2 * c][4 * qd + 1]),
                                 fmaxf(acc[cb][2 * c][4 * qd + 2], acc[cb][2 * c][4 * qd + 3]));
          const float o1 = fmaxf(fmaxf(acc[cb][2 * c + 1][4 * qd], acc[cb][2 * c + 1][4 * qd + 1]),
                                 fmaxf(acc[cb][2 * c + 1][4 * qd + 2], acc[cb][2 * c + 1][4 * qd + 3]));
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o0), __float_as_uint(o1), false, false);
          v[2 * qd] = __uint_as_float(r[0]);       // lower half: block 0's group 2 qd;     upper: block 1's
          v[2 * qd + 1] = __uint_as_float(r[1]);   // lower half: block 0's group 2 qd + 1; upper: block 1's
        }
        auto emit = [&](int seg, float m) {
          if (seg < 0) return;
          const float x = fmaxf(m * sc + bias, 0.f);
          omax = fmaxf(omax, x);
          if (nn < CoutF) atomicMax(outu + (size_t)seg * p.ldc + p.c_coff + nn, __float_as_uint(x));
        };
        int cur = sidv[c][0];
        float m = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) {
          const bool chg = sidv[c][i] != cur;
          if (chg) emit(cur, m);
          m = chg ? v[i] : fmaxf(m, v[i]);
          cur = sidv[c][i];
        }
        emit(cur, m);
        }
      }
      if (PL == 2 && p.out_amax) {
        const uint32_t wm = wave_max_u32(__float_as_uint(omax));
        if (lane == 0) amax_publish(p.out_amax, wm, tile * 4 + wave + strip, p0, p_hi, p.rps);
      }
    } else if (q.relu && p.K == 64) {
      // max over the 64 neighbours FIRST, on the raw accumulators (the scales are positive powers
      // of two, so max commutes with them exactly), then one scale + bias + ReLU per channel:
      // 1 instead of 4 vector instructions per accumulator element.  A wave's rows are NRB / 2 centroids.
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const float sc = __shfl(e_sc, cb * 32 + li);
        const float bias = __shfl(e_bias, cb * 32 + li);
        const int nn = n0 + cb * 32 + li;
#pragma unroll
        for (int c = 0; c < NRB / 2; ++c) {
          float m = acc[cb][2 * c][0];
#pragma unroll
          for (int rb = 2 * c; rb < 2 * c + 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[cb][rb][r]);
          m = fmaxf(m, __shfl_xor(m, 32));
          const float v = fmaxf(m * sc + bias, 0.f);
          omax = fmaxf(omax, v);
          const int row0 = p0 + wr * WROWS + c * 64;
          if (nn < CoutF && lh == 0 && row0 < p.P) p.out[(size_t)(row0 >> 6) * p.ldc + p.c_coff + nn] = v;
        }
      }
      if (PL == 2 && p.out_amax) {
        const uint32_t wm = wave_max_u32(__float_as_uint(omax));
        if (lane == 0) amax_publish(p.out_amax, wm, tile * 4 + wave + strip, p0, p_hi, p.rps);
      }
    } else if constexpr (NRB == 2) {
      // (no ReLU behind the last layer: the generic max epilogue, which takes [row block][channel block])
      f32x16 acct[2][2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const float sc = __shfl(e_sc, cb * 32 + li);
        const float bias = __shfl(e_bias, cb * 32 + li);
        float mx = -__builtin_inff(), mn = __builtin_inff();
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[cb][rb][r] * sc;
            acct[rb][cb][r] = v;
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
          }
        const float hi = mx + bias, lo = mn + bias;
        omax = fmaxf(omax, q.relu ? hi : fmaxf(fabsf(hi), fabsf(lo)));
      }
      if (p.out_amax) {
        const uint32_t wm = wave_max_u32(__float_as_uint(fmaxf(omax, 0.f)));
        if (lane == 0) amax_publish(p.out_amax, wm, tile * 4 + wave + strip, p0, p_hi, p.rps);
      }
      gemm_epilogue<EPI_MAX, 2>(q, acct, bg2, g, p0, n0, wave, wr, 0, li, lh, smemf);
    }
  }
  S4G_STAMP(11);
#undef S4G_F2_STRIP
#undef S4G_F2_TERM
}

template <int LOADER, int EPI2, int RW, int KC, int PL>
static int launch_mlp_chain(const GemmParams& p, int groups, hipStream_t st) {
  constexpr int CW = RW == 8 ? 8 : 4 / RW, RWN = RW == 8 ? 1 : RW;
  constexpr int NRB = PL == 1 ? S4G_CHAIN_BF16_NRB : 2;
  constexpr int BM = 32 * NRB * RWN, K = 64 * CW, NW = RWN * CW;
  constexpr size_t lds = sizeof(uint16_t) * PL * BM * (size_t)(K + 8) + sizeof(float) * (NW * 128 + 16);
  static_assert(lds <= (RW == 8 ? 160 : 80) * 1024, "two workgroups per CU (one for the 8-wave form)");
  static LdsAttrCache lds_cache;
  if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(&mlp_chain_kernel<LOADER, EPI2, RW, KC, PL>), lds, lds_cache)) return rc;
  const dim3 grid((unsigned)((p.P + BM - 1) / BM), (unsigned)groups);
  hipLaunchKernelGGL((mlp_chain_kernel<LOADER, EPI2, RW, KC, PL>), grid, dim3(64 * NW), lds, st, p);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

#ifdef S4G_VARIANTS
#define S4G_VARIANT_PART 2
#include "variants/gemm_resident.inc"
#undef S4G_VARIANT_PART
#endif

template <int LOADER, int EPI, int PL = 2>
static int launch_gemm_f16x2(const GemmParams& p, int groups, hipStream_t st) {
  constexpr int force = 0;   // (tile width by shape; forcing it was a tuning knob of round 1)
#ifdef S4G_VARIANTS
  // S4G_GEMM_RESIDENT=0 never / 1 whenever the shape qualifies / unset: only where it
  // measured faster than the tiled kernel (Cout >= 1024: +6 %; short strips lose to
  // the per-workgroup prologue)
  const char* rmode = s4g::knob("S4G_GEMM_RESIDENT");
  const bool no_resident = rmode && rmode[0] == '0';
  const bool any_resident = rmode && rmode[0] == '1';
  if constexpr (EPI != EPI_CHANNEL_FIRST && LOADER != LOAD_INTERP && LOADER != LOAD_INTERP_ADD) {
    // resident-A kernel: short contractions whose A panel fits LDS twice per CU
    const bool vec_ok = ((p.ldc | p.c_coff | p.c_gcol) & 3) == 0 &&
                        ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
    if (PL == 2 && !no_resident && !force && !p.out2 && p.Wfrag && (EPI != EPI_STORE || vec_ok) &&
        (EPI != EPI_MAX || p.K == 64) && (any_resident || p.Cout >= 1024)) {
      if (p.Kpad16 == 256 && p.Cout % 256 == 0)
        return launch_gemm_f16x2_resident<LOADER, EPI, 1, 256>(p, groups, st);
      if (p.Kpad16 == 128 && p.Cout % 256 == 0)
        return launch_gemm_f16x2_resident<LOADER, EPI, 1, 128>(p, groups, st);
      if (p.Kpad16 == 128 && p.Cout % 128 == 0)
        return launch_gemm_f16x2_resident<LOADER, EPI, 2, 128>(p, groups, st);
    }
  }
#endif
  // the INTERP / GATHER loaders hold too much per-row state for the wide tile's
  // 128 accumulator registers (they would spill)
  // ... and 256-wide tiles only pay when they still fill the chip twice over: measured on the
  // nine single-layer launches of a step, the 128-wide tile is 7-28 % faster below 512 workgroups
  // and for Cout <= 256, the 256-wide one 12-16 % faster above
  const int64_t wide_wgs = (int64_t)((p.P + GM_BM - 1) / GM_BM) * ((p.Cout + 255) / 256) * groups;
  const bool wide = force ? force == 4
                          : p.Cout > 256 && wide_wgs >= 512 &&
                                (LOADER == LOAD_PLAIN || LOADER == LOAD_GATHER_MLP1 || LOADER == LOAD_GATHER_ADD);
  if (wide) return launch_gemm_f16x2_cfg<LOADER, EPI, 4, PL>(p, groups, st);
  return launch_gemm_f16x2_cfg<LOADER, EPI, 2, PL>(p, groups, st);
}

template <int LOADER, int EPI, int NS, int WAVES>
static int launch_gemm_bf16x3_cfg(const GemmParams& p, int groups, hipStream_t st) {
  constexpr int BK = 32 * NS;
  size_t lds = sizeof(float) * GM_BM * (BK + 4) + sizeof(uint16_t) * 3 * GM_BN * (BK + 8);
  const size_t epi = sizeof(float) * 4 * 32 * 68;   // staged epilogue stores (either layout)
  if (lds < epi) lds = epi;
  // one-time, thread-safe (C++11 magic static): allow > 64 KB of dynamic LDS
  static LdsAttrCache lds_cache;
  if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(&mlp_gemm_bf16x3_kernel<LOADER, EPI, NS, WAVES>), lds, lds_cache)) return rc;
  const dim3 grid((unsigned)(p.mtiles * p.ntiles), (unsigned)groups);
  hipLaunchKernelGGL((mlp_gemm_bf16x3_kernel<LOADER, EPI, NS, WAVES>), grid, dim3(64 * WAVES), lds,
                     st, p);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

template <int LOADER, int EPI>
static int launch_gemm_bf16x3(const GemmParams& p, int groups, hipStream_t st) {
  // Measured on MI355X (tools/bench_gemm.py): 64-deep K tiles at one workgroup
  // per CU lose 20 %, eight-wave workgroups at 4 waves/SIMD change nothing
  // (-3 %); 128x128x32, four waves, two workgroups per CU is what ships.
  return launch_gemm_bf16x3_cfg<LOADER, EPI, 1, 4>(p, groups, st);
}

template <int LOADER, int EPI>
static int launch_gemm(const GemmParams& p, int groups, hipStream_t st) {
  const size_t lds = sizeof(float) * 2 * (GM_BM + GM_BN) * GM_LDS;
  static LdsAttrCache lds_cache;
  if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(&mlp_gemm_kernel<LOADER, EPI>), lds, lds_cache)) return rc;
  const dim3 grid((unsigned)(p.mtiles * p.ntiles), (unsigned)groups);
  hipLaunchKernelGGL((mlp_gemm_kernel<LOADER, EPI>), grid, dim3(GM_THREADS), lds, st, p);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

}  // namespace s4g

// Instantiations of the fused chain kernel: (first layer's loader, final epilogue, RW code
// (2: C = 128, 1: C = 256, 8: C = 512), panels the first layer is deep).  The dispatch and
// s4g_gemm_chain_supported are generated from this one list.
#define S4G_FUSED2_LIST(X)                                                                       \
  X(LOAD_GATHER_MLP1, EPI_MAX, 2, 1)                                                             \
  X(LOAD_GATHER_MLP1, EPI_MAX, 1, 1)                                                             \
  X(LOAD_GATHER_ADD, EPI_MAX, 2, 1)                                                              \
  X(LOAD_GATHER_ADD, EPI_MAX, 1, 1)                                                              \
  X(LOAD_GATHER_ADD, EPI_MAX, 8, 1)   /* 512-wide pairs: eight waves, one workgroup per CU */    \
  X(LOAD_PLAIN, EPI_MAX, 2, 1)                                                                   \
  X(LOAD_PLAIN, EPI_MAX, 1, 1)                                                                   \
  X(LOAD_PLAIN, EPI_MAX, 8, 1)                                                                   \
  X(LOAD_PLAIN, EPI_STORE, 2, 1)                                                                 \
  X(LOAD_PLAIN, EPI_STORE, 1, 1)                                                                 \
  X(LOAD_PLAIN, EPI_STORE, 1, 2)      /* first layer two panels deep (512 -> 256 -> ...) */      \
  X(LOAD_PLAIN, EPI_STORE, 1, 4)      /* ... or four (the 1 024-deep single-layer launches) */   \
  X(LOAD_PLAIN, EPI_STORE, 8, 1)                                                                 \
  X(LOAD_INTERP_ADD, EPI_STORE, 2, 1)                                                            \
  X(LOAD_INTERP_ADD, EPI_STORE, 1, 1)                                                            \
  X(LOAD_INTERP_ADD, EPI_STORE, 8, 1) /* FP level 1's last layer + the next level's linear first layer */

#ifdef S4G_CHAIN_STAMPS
extern "C" int s4g_debug_chain_stamps(unsigned long long* host_out_512x16) {
  return (int)hipMemcpyFromSymbol(host_out_512x16, HIP_SYMBOL(s4g::g_chain_stamps), sizeof(unsigned long long) * 512 * 16);
}
#endif

extern "C" int s4g_gemm_chain_supported(int loader, int epilogue, int C, int Kpad16) {
  using namespace s4g;
  const int r = C == 128 ? 2 : (C == 256 ? 1 : (C == 512 ? 8 : 0));
  if (r == 0 || Kpad16 <= 0 || Kpad16 % C) return 0;
  const int kc = Kpad16 / C;
#define S4G_FUSED2_Q(L, E, R, KCH) \
  if (loader == L && epilogue == (int)E && r == R && kc == KCH) return 1;
  S4G_FUSED2_LIST(S4G_FUSED2_Q)
#undef S4G_FUSED2_Q
  return 0;
}

extern "C" int s4g_mlp_gemm_f32(const s4g_gemm_desc_t* d, s4g_stream_t stream) {
  using namespace s4g;
  if (!d || d->P < 0 || d->Cout <= 0 || d->groups <= 0 || !d->bias) return S4G_EINVAL;
  const bool split = d->precision == S4G_GEMM_BF16X3 || d->precision == S4G_GEMM_BF16;
  const bool h2 = d->precision == S4G_GEMM_F16X2;
  if (h2) {
    if (!d->W_f16x2 || !d->w_inv_scale || d->Kpad16 <= 0 || (d->Kpad16 & 15) ||
        !(d->a_amax_floor >= 0.f) || (!d->a_amax && !d->a_amax2 && !(d->a_amax_floor > 0.f)))
      return S4G_EINVAL;
  } else if (split) {
    if (!d->W_bf16x3 || d->Kpad16 <= 0 || (d->Kpad16 & 15)) return S4G_EINVAL;
  } else if (d->precision == S4G_GEMM_FP32) {
    if (!d->W || d->Kpad <= 0 || (d->Kpad & 7)) return S4G_EINVAL;
  } else {
    return S4G_EINVAL;
  }
  if (d->P == 0) return S4G_OK;
  GemmParams p;
  p.P = d->P; p.Cin = d->Cin; p.Kpad = d->Kpad; p.Cout = d->Cout; p.relu = d->relu;
  p.W = d->W; p.bias = d->bias;
  p.A = d->A; p.lda = d->lda; p.a_coff = d->a_coff; p.a_gcol = d->a_gcol;
  p.gidx = d->gidx; p.feat = d->feat; p.xyz = d->xyz; p.ctr = d->ctr;
  p.Cf = d->Cf; p.N = d->N; p.M = d->M; p.K = d->K;
  p.mlp1 = (const float4*)d->mlp1_w;
  p.rel4 = (const float4*)d->rel_xyz4;
  p.seg4 = d->seg4;
  p.seg_rows = d->seg_rows;
  p.out2 = d->out2;
  p.ldc2 = d->ldc2;
  p.split_n = d->out2 ? d->split_n : 0;
  p.out_amax2 = (uint32_t*)d->out_amax2;
  p.nidx = d->nidx; p.nw = d->nw; p.sparse = d->sparse; p.dense = d->dense;
  p.C2 = d->C2; p.C1 = d->C1; p.N2 = d->N2; p.N1 = d->N1;
  p.out = d->out; p.ldc = d->ldc; p.c_coff = d->c_coff; p.c_gcol = d->c_gcol;
  p.w_gstride = d->w_gstride; p.b_gstride = d->b_gstride;
  for (int i = 0; i < 4; ++i) p.cf_ptr[i] = d->cf_ptr[i];
  for (int i = 0; i < 5; ++i) p.cf_start[i] = d->cf_start[i];
  p.cf_sigmoid_from = d->cf_sigmoid_from; p.cf_N = d->cf_N;
  p.W3 = (const uint16_t*)d->W_bf16x3;
  p.Kpad16 = d->Kpad16;
  p.bf16_single = d->precision == S4G_GEMM_BF16 ? 1 : 0;
  p.w3_plane = (size_t)d->groups * (size_t)d->Cout * (size_t)d->Kpad16;
  p.Wh2 = (const uint16_t*)d->W_f16x2;
  p.w_inv_scale = d->w_inv_scale;
  p.a_amax = d->a_amax; p.a_amax2 = d->a_amax2; p.a_amax_floor = d->a_amax_floor;
  p.out_amax = (uint32_t*)d->out_amax;
  p.rps = d->rows_per_scene > 0 ? d->rows_per_scene : 0;
  p.Wfrag = (const uint16_t*)d->W_f16x2_frag;
  p.lbias = d->loader_bias;
  p.Wfrag2 = (const uint16_t*)d->W2_f16x2_frag;
  p.w_inv_scale2 = d->w2_inv_scale;
  p.bias2 = d->bias2;
  p.Cout2 = d->Cout2;
  p.relu2 = d->relu2;
  p.Wfrag3 = (const uint16_t*)d->W3_f16x2_frag;
  p.w_inv_scale3 = d->w3_inv_scale;
  p.bias3 = d->bias3;
  p.Cout3 = d->Cout3;
  p.relu3 = d->relu3;
  p.mtiles = (d->P + GM_BM - 1) / GM_BM;
  p.ntiles = (d->Cout + GM_BN - 1) / GM_BN;
  hipStream_t st = (hipStream_t)stream;

  // loader-specific validation
  if (d->loader == S4G_GEMM_LOAD_PLAIN) {
    if (!d->A || (d->lda & 3) || (d->a_coff & 3) || (d->a_gcol & 3) || (d->Cin & 3)) return S4G_EINVAL;
  } else if (d->loader == S4G_GEMM_LOAD_GATHER) {
    if (!d->gidx || !d->xyz || !d->ctr || (d->Cf & 3) || (d->Cf > 0 && !d->feat) ||
        d->K <= 0 || d->M <= 0 || d->N <= 0 || d->groups != 1)
      return S4G_EINVAL;
  } else if (d->loader == S4G_GEMM_LOAD_GATHER_MLP1) {
    // (with rel_xyz4 the rows are pre-gathered: gidx / xyz / ctr are not read)
    if ((!d->rel_xyz4 && (!d->gidx || !d->xyz || !d->ctr)) || ((uintptr_t)d->rel_xyz4 & 15) ||
        !d->mlp1_w || (d->Cin & 3) || d->K <= 0 || d->M <= 0 || d->N <= 0 || d->groups != 1 ||
        ((uintptr_t)d->mlp1_w & 15))
      return S4G_EINVAL;
    // distinct-row form: only the fused chain with the max epilogue knows how to merge a centroid's pieces
    if ((d->seg4 != nullptr) != (d->seg_rows != nullptr)) return S4G_EINVAL;
    if (d->seg4 && (!d->rel_xyz4 || !d->W2_f16x2_frag || d->epilogue != S4G_GEMM_EPI_MAX || d->K != 64 ||
                    d->rows_per_scene <= 0 || (d->rows_per_scene & 255) || !d->relu2 || d->W3_f16x2_frag))
      return S4G_EINVAL;
  } else if (d->loader == S4G_GEMM_LOAD_GATHER_ADD) {
    if (!d->gidx || !d->xyz || !d->ctr || !d->mlp1_w || !d->feat || (d->Cin & 3) || d->Cf != d->Cin ||
        d->K <= 0 || d->M <= 0 || d->N <= 0 || d->groups != 1 || ((uintptr_t)d->mlp1_w & 15) ||
        ((uintptr_t)d->feat & 15))
      return S4G_EINVAL;
  } else if (d->loader == S4G_GEMM_LOAD_INTERP_ADD) {
    if (!d->nidx || !d->nw || !d->sparse || !d->loader_bias || (d->Cin & 3) || d->C2 != d->Cin ||
        d->N1 <= 0 || d->N2 <= 0 || d->groups != 1 ||
        (((uintptr_t)d->sparse | (uintptr_t)d->dense | (uintptr_t)d->loader_bias) & 15))
      return S4G_EINVAL;
  } else if (d->loader == S4G_GEMM_LOAD_INTERP) {
    if (!d->nidx || !d->nw || !d->sparse || (d->C2 & 3) || (d->C1 & 3) ||
        (d->C1 > 0 && !d->dense) || d->N1 <= 0 || d->N2 <= 0 || d->groups != 1)
      return S4G_EINVAL;
  } else {
    return S4G_EINVAL;
  }
  if (d->loader != S4G_GEMM_LOAD_GATHER_MLP1 && (d->seg4 || d->seg_rows)) return S4G_EINVAL;
  if (d->out2) {   // second output tensor: plain single layers of the f16x2 / bf16 kernels only
    const bool bf1_ = d->precision == S4G_GEMM_BF16;
    if (!(h2 || bf1_) || d->loader != S4G_GEMM_LOAD_PLAIN || d->epilogue != S4G_GEMM_EPI_STORE || d->W2_f16x2_frag ||
        d->groups != 1 || d->split_n <= 0 || (d->split_n & 255) || d->split_n >= d->Cout || (d->Cout & 255) ||
        (d->ldc2 & 3) || ((uintptr_t)d->out2 & 15) || d->c_coff != 0 || (d->ldc & 3) || ((uintptr_t)d->out & 15))
      return S4G_EINVAL;
  }
  if (d->epilogue == S4G_GEMM_EPI_MAX) {
    if (!(d->K == 16 || d->K == 32 || d->K == 64) || d->groups != 1 || !d->out) return S4G_EINVAL;
  } else if (d->epilogue == S4G_GEMM_EPI_STORE) {
    if (!d->out) return S4G_EINVAL;
  } else if (d->epilogue == S4G_GEMM_EPI_CHANNEL_FIRST) {
    if (d->cf_N <= 0 || d->groups != 1) return S4G_EINVAL;
  } else {
    return S4G_EINVAL;
  }

  if (d->W2_f16x2_frag) {
    // two fused layers: K = C -> C -> Cout2 with C = 128 (128 positions per workgroup) or
    // C = 256 (64 positions), epilogue MAX (K == 64 neighbours) or STORE
    // layer 1 may be one or two panels deep (Kpad16 = C or 2 C)
    const bool c128 = d->Cout == 128 && d->Kpad16 == 128,
               c256 = d->Cout == 256 && (d->Kpad16 == 256 || d->Kpad16 == 512),
               c512 = d->Cout == 512 && d->Kpad16 == 512;
    const bool store = d->epilogue == S4G_GEMM_EPI_STORE;
    const bool bf1 = d->precision == S4G_GEMM_BF16;   // one bf16 plane, one product, no scales
    if ((!h2 && !bf1) || (!store && (d->epilogue != S4G_GEMM_EPI_MAX || d->K != 64 || (d->P & 63))) ||
        (!c128 && !c256 && !c512) || d->Cout2 <= 0 || (d->Cout2 & 63) || (!store && d->groups != 1) || !d->W_f16x2_frag ||
        (d->W3_f16x2_frag && (d->Cout2 != d->Cout || d->Cout3 <= 0 || (d->Cout3 & 63) ||
                              (h2 && !d->w3_inv_scale) || !d->bias3)) ||
        (h2 && !d->w2_inv_scale) || !d->bias2 ||
        (store && (((d->ldc | d->c_coff | d->c_gcol) & 3) || ((uintptr_t)d->out & 15))))
      return S4G_EINVAL;
    // the xyz-only first layer of an SA level on pre-gathered records: as one MFMA step inside the chain kernel
    // (f16x2 form; S4G_MLP1_MFMA=0, read per launch: the vector-ALU loader, for tests and A/B runs)
    if (h2 && d->loader == S4G_GEMM_LOAD_GATHER_MLP1 && d->rel_xyz4 && d->epilogue == S4G_GEMM_EPI_MAX &&
        d->Kpad16 == d->Cout && d->Cin <= d->Cout && (c128 || c256)) {
      const char* m0 = s4g::knob("S4G_MLP1_MFMA");
      if (!(m0 && m0[0] == '0'))
        return c128 ? launch_mlp_chain<LOAD_REL_MLP1, EPI_MAX, 2, 1, 2>(p, d->groups, st)
                    : launch_mlp_chain<LOAD_REL_MLP1, EPI_MAX, 1, 1, 2>(p, d->groups, st);
    }
#define S4G_FUSED2_CASE(L, E, R, KCH)                                                        \
  if (d->loader == L && (int)d->epilogue == (int)E && (c128 ? 2 : (c256 ? 1 : 8)) == R &&     \
      d->Kpad16 / d->Cout == KCH)                                                             \
    return bf1 ? launch_mlp_chain<L, E, R, KCH, 1>(p, d->groups, st)                 \
               : launch_mlp_chain<L, E, R, KCH, 2>(p, d->groups, st);
    S4G_FUSED2_LIST(S4G_FUSED2_CASE)
#undef S4G_FUSED2_CASE
    return S4G_EUNSUPPORTED;
  }
  // plain single layers whose widths allow it run on the chain kernel's first-layer machinery (round 4):
  // 64 positions x 256 channels per workgroup, the 256-channel strips of a wider layer as groups that
  // share A (S4G_GEMM_SINGLE_CHAIN=0: the tiled kernel)
  {
    const char* sc_env = s4g::knob("S4G_GEMM_SINGLE_CHAIN");   // (read per launch: a test knob)
    const bool single_chain = !(sc_env && sc_env[0] == '0');
    const bool bf1 = d->precision == S4G_GEMM_BF16;
    const int kc = d->Kpad16 / 256;
    if (single_chain && (h2 || bf1) && d->loader == S4G_GEMM_LOAD_PLAIN && d->epilogue == S4G_GEMM_EPI_STORE &&
        d->W_f16x2_frag && d->groups == 1 && d->Cout % 256 == 0 && d->Kpad16 % 256 == 0 &&
        (kc == 1 || kc == 2 || kc == 4) && ((d->ldc | d->c_coff) & 3) == 0 && ((uintptr_t)d->out & 15) == 0 &&
        // measured per launch (profiles/r04_single_layer.md): it wins from four 256-channel strips or a
        // 1 024-deep contraction (fp0.0s / fp0.0d / fp0.1 / fp1.0s: -6 ... -25 %), the tiled kernel below that
        (d->Cout >= 1024 || d->Kpad16 >= 1024 || (sc_env && sc_env[0] == '1'))) {
      GemmParams q = p;
      q.Cout = 256;
      q.a_gcol = 0;
      q.c_gcol = 256;
      q.b_gstride = 256;
      const int strips = d->Cout / 256;
#define S4G_SINGLE_CASE(KCH)                                                                     \
  if (kc == KCH)                                                                                 \
    return bf1 ? launch_mlp_chain<LOAD_PLAIN, EPI_STORE, 1, KCH, 1>(q, strips, st)               \
               : launch_mlp_chain<LOAD_PLAIN, EPI_STORE, 1, KCH, 2>(q, strips, st);
      S4G_SINGLE_CASE(1)
      S4G_SINGLE_CASE(2)
      S4G_SINGLE_CASE(4)
#undef S4G_SINGLE_CASE
    }
  }
  // single-product bf16: the swizzled-LDS tile kernel with one plane
  constexpr bool bf16_tiled = true;
#define S4G_GEMM_CASE(L, E)                                            \
  if (d->loader == L && d->epilogue == E)                              \
    return h2 ? launch_gemm_f16x2<L, E>(p, d->groups, st)              \
           : (p.bf16_single && bf16_tiled) ? launch_gemm_f16x2<L, E, 1>(p, d->groups, st) \
           : split ? launch_gemm_bf16x3<L, E>(p, d->groups, st)        \
                   : launch_gemm<L, E>(p, d->groups, st);
  S4G_GEMM_CASE(LOAD_PLAIN, EPI_STORE)
  S4G_GEMM_CASE(LOAD_PLAIN, EPI_MAX)
  S4G_GEMM_CASE(LOAD_PLAIN, EPI_CHANNEL_FIRST)
  S4G_GEMM_CASE(LOAD_GATHER, EPI_STORE)
  S4G_GEMM_CASE(LOAD_GATHER, EPI_MAX)
  S4G_GEMM_CASE(LOAD_INTERP, EPI_STORE)
  S4G_GEMM_CASE(LOAD_GATHER_MLP1, EPI_STORE)
  S4G_GEMM_CASE(LOAD_GATHER_MLP1, EPI_MAX)
  S4G_GEMM_CASE(LOAD_GATHER_ADD, EPI_STORE)
  S4G_GEMM_CASE(LOAD_GATHER_ADD, EPI_MAX)
  S4G_GEMM_CASE(LOAD_INTERP_ADD, EPI_STORE)
#undef S4G_GEMM_CASE
  return S4G_EUNSUPPORTED;
}
