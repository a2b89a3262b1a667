/*
 * s4g_ops.h -- C ABI of libs4g_hip.so, the MI355X (gfx950) replacement for the
 * reference's pybind11 extension `pn2_ext`.
 *
 * Reference interface replaced (paths relative to
 * /root/reference/inference/grasp_proposal/network_models/models/pointnet2_utils/):
 *   csrc/main.cpp:6-14 registers seven at::Tensor functions; each entry point
 *   below cites the prototype it replaces.  The reference's Python wrappers
 *   (functions.py:42,72,99,105,127,163,170) are the only callers.
 *
 * Conventions (all entry points):
 *   - extern "C", plain device pointers and int64 sizes, no torch types.
 *   - return 0 on success, <0 = S4G_E* argument error (nothing launched),
 *     >0 = hipError_t from the launch.  No exceptions, no allocation, no
 *     host synchronisation, no global state; re-entrant.
 *   - the caller owns every buffer (outputs and workspace) and passes the HIP
 *     stream to launch on (hipStream_t as void*; NULL = default stream) --
 *     the reference launches on the legacy default stream with no guard.
 *   - clouds are channel-first fp32 exactly as the Python API hands them over,
 *     (B,3,N) contiguous; the reference's internal (B,N,3) transposed copies
 *     (sampling_kernel.cu:141, ball_query_kernel.cu:105-106,
 *     interpolate_kernel.cu:111-112) are not made.
 *   - indices are int64 as in the reference (AT kLong outputs).
 *   - `flags`: bit 0 (S4G_FLAG_FMAD) selects the distance arithmetic.
 *       0: every fp32 op rounded separately, d = ((dx*dx)+(dy*dy))+(dz*dz)
 *          (canonical; what the oracle and all parity tests use)
 *       1: emulate nvcc's default -fmad contraction,
 *          d = fma(dz,dz, fma(dy,dy, dx*dx)).
 */
#ifndef S4G_OPS_H_
#define S4G_OPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: operators; 2: f16x2 contraction fields, crop / voxel / outlier / grid 3-NN / _ws entry points;
 * 3: fused layer chains (W2 / W3 fields), GATHER_ADD / INTERP_ADD loaders, s4g_interp_add_cl_f32,
 *    s4g_group_points_ws_f32, device-side cell choice of s4g_three_nn_grid_f32 (cell < 0).
 * 4: per-scene activation maxima (rows_per_scene), bf16 chains, s4g_heads_chain_f32.
 * 5: s4g_group_rel_xyz_i32 and the rel_xyz4 field of s4g_gemm_desc_t.
 * 6: pre_* members of s4g_heads_desc_t (the last FP level's tail in front of the heads).
 * 7: s4g_fps_gather_ex_i32, s4g_fps_prefix_check_f32, s4g_fps_prepass_f32 (no layout change).
 * 8: s4g_group_rel_xyz_unique_i32 and the seg4 / seg_rows fields of s4g_gemm_desc_t (the first SA level
 *    contracts a centroid's distinct rows only); out2 / ldc2 / split_n / out_amax2 (two layers that read the
 *    same tensor as one launch); the operators in double (*_f64); s4g_build_variants.
 * 9: s4g_heads_desc_t.out_batch_stride (the four heads written as channel slices of one packed
 *    (B, 21, N) tensor: the all-gather payload of the multi-GPU path without a packing copy).
 * 10: s4g_group_points_backward_det_f32 / s4g_three_interpolate_backward_det_f32 / s4g_scatter_det_workspace_bytes
 *    (the backward scatters in a fixed order: run-to-run bit-identical, equal to the oracle's sequential sum).
 * 11: s4g_heads_desc_t.head_mask (a launch may evaluate a subset of the four heads).
 * 12: s4g_test_knobs_enabled (the A/B knobs below are ignored without S4G_TEST_KNOBS=1; no layout change),
 *     s4g_collision_counts_n_f32 (collision counts over padded best-first pose lists with device-side counts),
 *     s4g_sort_pairs_u32 / s4g_exclusive_scan_i32 (the library's own stable radix sort and scan). */
#define S4G_ABI_VERSION 12

/* ---------------------------------------------------------------------------
 * Environment variables.
 *
 * PRODUCTION SURFACE -- the only variables the shipped library and its Python host read on their own (7):
 *   S4G_TEST_KNOBS=1               master switch for everything in the second list (library and host side)
 *   S4G_HIP_LIB=path               load another build of this library (warns; tools/ab_libs.sh)
 *   S4G_GEMM_MODE=f16x2|bf16x3|fp32|bf16   default contraction arithmetic of FusedPointNet2 (constructor argument wins)
 *   S4G_DIST_MODE=fmad             distance arithmetic contract (S4G_FLAG_FMAD) instead of strict
 *   S4G_BACKWARD=atomic            group_points / three_interpolate backward through the atomicAdd kernels (the
 *                                  reference's scheme, order undefined) instead of the deterministic sorted-segment sums
 *   S4G_GEO_STREAMS=n, S4G_DENSE_STREAMS=n   geometry / contraction streams of the pipeline (2 / 1)
 *  (bench.py reads S4G_BENCH_FORCE_DIST, S4G_BENCH_TIMER_EVERY, NCCL_MAX_NCHANNELS, and for tests/test_bench_world.py
 *   S4G_BENCH_BACKEND=gloo; the oracle reads S4G_ORACLE_LIB / S4G_ORACLE_F64 -- test infrastructure, not the product.)
 *
 * A/B AND TEST KNOBS -- IGNORED unless the process sets S4G_TEST_KNOBS=1 (tests/conftest.py does; tools that set one set
 * it too) or the library is a -DS4G_VARIANTS measurement build (s4g_test_knobs_enabled() says which).  Round 6: a stray
 * variable in a launcher's environment can no longer change which kernel a production rank runs.  Every alternative is
 * exact (same results; the defaults are the measured-fastest paths), so the switch can change speed, never values.
 *  library side (s4g::knob in csrc/s4g_common.h; per call unless noted):
 *   S4G_FPS_MODE=dense|pruned      FPS kernel for N <= 25 600: full scan | group-pruned (default: pruned
 *                                  above 10 240 points, and above 5 120 when M >= 2 048).  =dense also turns the L2-resident pruned kernel
 *                                  off for 25 600 < N <= 65 535 (the streaming kernel runs instead)
 *   S4G_FPS_DENSE_STEPS=k          first k picks by the full-scan kernel in front of the pruned one (0)
 *   S4G_BQ_MODE=scan|grid          ball query path (default: grid from 8 192 points)
 *   S4G_GRID_BUILD=loop            streaming grid build for every size (read once)
 *   S4G_NN_SPLIT=0                 3-NN: never the split scan for 24 <= N2 <= 2 048
 *   S4G_NN_CELL_FACTOR=f           3-NN operator API: cell edge = f x measured 3rd-neighbour spacing (1.75; once)
 *   S4G_INTERP_MODE=lane           three_interpolate: lane-per-point kernel instead of the LDS tile
 *   S4G_GEMM_SINGLE_CHAIN=0|1      plain single layers never / wherever supported on mlp_chain_kernel's first-layer
 *                                  form (default: where it measured faster: Cout >= 1024 or K >= 1024)
 *   S4G_MLP1_MFMA=0                first SA level's 3 -> C layer on the vector ALU (the chain kernel's loader) instead of
 *                                  one MFMA step inside the chain kernel (f16x2 form with rel_xyz4 records)
 *   measurement builds only: S4G_FPS_MODE=cluster|hybrid, S4G_BQ_MODE=cell (+ S4G_BQ_CELL_WGS), S4G_GEMM_RESIDENT=0|1
 *  host side (_cabi.knob; read when a FusedPointNet2 is built / an operator is called):
 *   S4G_SA_UNIQUE=0                first SA level contracts all K rows, padding copies included
 *   S4G_REL_XYZ=0                  first SA level's loader follows the indices itself
 *   S4G_SA_LINEAR_FIRST=0, S4G_FP_LINEAR_FIRST=0   no linear-layer-before-grouping / -interpolation
 *   S4G_FP_LOADER_ADD=auto|none|levels, S4G_FP_CHAIN_NEXT=0   where the FP sums are formed
 *   S4G_GEMM_FUSE2=0, S4G_GEMM_FUSE3=0, S4G_GEMM_FUSE512=0    layer chains as separate launches
 *   S4G_MERGE_SHARED=0             sa{l}.0f and fp{f}.0d (same input tensor) as two launches instead of one
 *   S4G_HEADS_FUSED=0, S4G_HEADS_PRE=0   heads layer by layer / FP tail outside the heads launch
 *   S4G_FPS_PREFIX=0               always sample SA levels 2 and 3 (no prefix proof)
 *   S4G_NN_MODE=scan               3-NN: never the cell-grid search
 *   S4G_PACKED_OUT=0               FusedPointNet2 returns four head tensors of their own instead of channel slices of one
 *                                  packed (B, 21, N) tensor
 * ------------------------------------------------------------------------- */

#define S4G_OK 0
#define S4G_EINVAL (-1)     /* bad size / null pointer */
#define S4G_EWORKSPACE (-2) /* workspace too small */
#define S4G_EUNSUPPORTED (-3)

#define S4G_FLAG_FMAD 1

typedef void *s4g_stream_t; /* hipStream_t */

/* operator ids for s4g_workspace_bytes */
#define S4G_OP_FPS 1
#define S4G_OP_BALL_QUERY 2
#define S4G_OP_THREE_NN 3

int s4g_abi_version(void);
/* ABI >= 8.  1 if the library is a measurement build (make HIPFLAGS_EXTRA=-DS4G_VARIANTS) that also
 * carries the measured-slower kernel variants of csrc/variants/ (two-CU / hybrid FPS for 51 200 points,
 * the cell-centric ball query, the resident-A single-layer contraction); 0 for the shipped library,
 * where S4G_FPS_MODE=cluster|hybrid, S4G_BQ_MODE=cell and S4G_GEMM_RESIDENT select nothing. */
int s4g_build_variants(void);
/* 1 when the A/B / test knobs listed above are honoured in this process (S4G_TEST_KNOBS=1 in the environment, or a
 * measurement build); 0: the library ignores every one of them. */
int s4g_test_knobs_enabled(void);
const char *s4g_error_string(int code);

/* Bytes of device scratch an operator needs for the given problem
 * (0 if none).  dims: FPS (B,N,M,0); BALL_QUERY (B,N,M,K); THREE_NN (B,N1,N2,0). */
size_t s4g_workspace_bytes(int op, int64_t B, int64_t d0, int64_t d1, int64_t d2);

/* FarthestPointSample(points (B,3,N), num_centroids) -> index (B,M) int64
 * replaces csrc/sampling.h:7-9, csrc/sampling_kernel.cu:128-172.
 * Requires M > 0, N >= M (sampling_kernel.cu:137-139).  idx[b,0] = 0. */
int s4g_fps_f32(const float *xyz_b3n, int64_t B, int64_t N, int64_t M,
                int64_t *idx_bm, void *ws, size_t ws_bytes, int flags,
                s4g_stream_t stream);

/* BallQuery(points (B,3,N), centroids (B,3,M), radius, K)
 *   -> index (B,M,K) int64, count (B,M) int64
 * replaces csrc/ball_query.h:7-11, csrc/ball_query_kernel.cu:89-133.
 * Outputs are fully written (rows without any hit are zero, as the
 * reference's at::zeros leaves them). */
int s4g_ball_query_f32(const float *xyz_b3n, const float *ctr_b3m, int64_t B,
                       int64_t N, int64_t M, float radius, int64_t K,
                       int64_t *idx_bmk, int64_t *cnt_bm, void *ws,
                       size_t ws_bytes, int flags, s4g_stream_t stream);

/* GroupPointsForward(input (B,C,N), index (B,M,K)) -> (B,C,M,K)
 * replaces csrc/grouping.h:7-9, csrc/grouping_kernel.cu:32-54. */
int s4g_group_points_f32(const float *in_bcn, const int64_t *idx_bmk, int64_t B,
                         int64_t C, int64_t N, int64_t M, int64_t K,
                         float *out_bcmk, s4g_stream_t stream);

/* GroupPointsBackward(grad_output (B,C,M,K), index, N) -> grad_input (B,C,N)
 * replaces csrc/grouping.h:11-14, csrc/grouping_kernel.cu:106-152.
 * grad_in is zeroed by the call, then scatter-added (fp32 atomics). */
/* group_points for C == 3 (the xyz grouping of QueryGrouper, modules.py:42) through an
 * index-ordered (x, y, z, 0) copy in `ws` (B * N * 16 bytes): one 16-byte gather per
 * neighbour.  Same output as s4g_group_points_f32(C = 3); without a (large enough,
 * 16-byte aligned) workspace or with M*K % 4 != 0 it IS that call. */
int s4g_group_points_xyz_f32(const float *xyz_b3n, const int64_t *idx_bmk, int64_t B, int64_t N,
                             int64_t M, int64_t K, float *out_b3mk, void *ws, size_t ws_bytes,
                             s4g_stream_t stream);
/* s4g_group_points_f32 through a channels-last copy of the features in `ws` (B * N * C
 * floats): 64 channels of a neighbour are one 256-byte read and the channel-first rows
 * leave as 16-byte stores.  Bit-identical output; without a (large enough, 16-byte
 * aligned) workspace or with C % 4 != 0 it IS that call. */
int s4g_group_points_ws_f32(const float *feat_bcn, const int64_t *idx_bmk, int64_t B, int64_t C,
                            int64_t N, int64_t M, int64_t K, float *out_bcmk, void *ws,
                            size_t ws_bytes, s4g_stream_t stream);
int s4g_group_points_backward_f32(const float *gout_bcmk,
                                  const int64_t *idx_bmk, int64_t B, int64_t C,
                                  int64_t N, int64_t M, int64_t K,
                                  float *gin_bcn, s4g_stream_t stream);

/* gather_points(points (B,C,N), index (B,M)) -> (B,C,M)
 * replaces the torch.gather in functions.py:10-25. */
int s4g_gather_points_f32(const float *in_bcn, const int64_t *idx_bm, int64_t B,
                          int64_t C, int64_t N, int64_t M, float *out_bcm,
                          s4g_stream_t stream);

/* PointSearch(query (B,3,N1), key (B,3,N2), 3)
 *   -> index (B,N1,3) int64, SQUARED distance (B,N1,3) fp32
 * replaces csrc/interpolate.h:8-11, csrc/interpolate_kernel.cu:92-132.
 * Requires N2 >= 3 (interpolate_kernel.cu:106).
 * Out of contract, but contained: a query with a NaN / inf coordinate (or squared distances that all overflow) enters no
 * key at all; the reference then leaves its initialisers -- index -1, distance +inf in slot 0 -- and its consumers read
 * row -1.  Every 3-NN entry point of this library (this one, _i32, _grid, _weights*, _f64) writes index 0 for such a
 * slot and keeps the distance: the interpolation weight of that slot is 0 either way, and nothing downstream can be
 * driven out of bounds by a bad depth pixel (a GPU memory fault takes the process down). */
int s4g_three_nn_f32(const float *q_b3n1, const float *k_b3n2, int64_t B,
                     int64_t N1, int64_t N2, int64_t *idx_bn3, float *d2_bn3,
                     void *ws, size_t ws_bytes, int flags, s4g_stream_t stream);

/* InterpolateForward(input (B,C,N2), index (B,N1,3), weight (B,N1,3))
 *   -> (B,C,N1)
 * replaces csrc/interpolate.h:13-16, csrc/interpolate_kernel.cu:191-236. */
int s4g_three_interpolate_f32(const float *feat_bcn2, const int64_t *idx_bn3,
                              const float *w_bn3, int64_t B, int64_t C,
                              int64_t N2, int64_t N1, float *out_bcn1,
                              int flags, s4g_stream_t stream);

/* InterpolateBackward(grad_output (B,C,N1), index, weight, N2) -> (B,C,N2)
 * replaces csrc/interpolate.h:18-22, csrc/interpolate_kernel.cu:296-341. */
/* s4g_three_interpolate_f32 through a channels-last copy of the sparse features in `ws`
 * (B * N2 * C floats): a quad of channels is one 16-byte gather.  Bit-identical output;
 * without a (large enough, 16-byte aligned) workspace or with C % 4 != 0 it IS that call. */
int s4g_three_interpolate_ws_f32(const float *feat_bcn2, const int64_t *idx_bn3, const float *w_bn3,
                                 int64_t B, int64_t C, int64_t N2, int64_t N1, float *out_bcn1,
                                 void *ws, size_t ws_bytes, int flags, s4g_stream_t stream);
/* Fast path, FP levels: the first shared-MLP layer is linear, so it is applied to the sparse
 * features BEFORE the interpolation (and to the skip features separately); this call then
 * forms out[p][c] = act(y[p][c] + bias[c] + sum_k nw[p][k] * sparse[b*N2 + nidx[p][k]][c]) on
 * channels-last tensors (y may be NULL; C % 4 == 0, C <= 1024) and leaves max|out| in
 * out_amax64 (B rows of 64 uint32 slots, one row per scene, zeroed by the caller; may be NULL).  No reference counterpart:
 * modules.py:122-128 interpolates first; the two orders agree to fp32 round-off. */
int s4g_interp_add_cl_f32(const float *y_pc, const float *sparse_rc, const int32_t *nidx_p3,
                          const float *nw_p3, const float *bias_c, int64_t B, int64_t N1, int64_t N2,
                          int64_t C, int relu, float *out_pc, float *out_amax64, s4g_stream_t stream);
int s4g_three_interpolate_backward_f32(const float *gout_bcn1,
                                       const int64_t *idx_bn3,
                                       const float *w_bn3, int64_t B, int64_t C,
                                       int64_t N2, int64_t N1, float *gin_bcn2,
                                       s4g_stream_t stream);

/* ABI >= 10.  The two backward scatters DETERMINISTICALLY: same signatures plus a workspace of
 * s4g_scatter_det_workspace_bytes(B, N, T) bytes (256-byte aligned; T = M K for group_points, 3 N1 for
 * three_interpolate; 0 = the sizes are not supported: B T and B N must stay below 2^31).  The contributions are
 * ordered by a stable radix sort on (scene, target point) and every target is summed by one thread in ascending
 * position order -- the sum a sequential loop over the positions forms: run-to-run bit-identical and equal to
 * the CPU oracle bit for bit, where the reference's atomicAdd scatter (grouping_kernel.cu:94,
 * interpolate_kernel.cu:283; s4g_*_backward_f32 above) leaves the order to the hardware.  An index outside
 * [0, N) contributes nothing. */
size_t s4g_scatter_det_workspace_bytes(int64_t B, int64_t N, int64_t T);
/* ... plus room for a channels-last copy of the gradients (C >= 32; weighted = 0: group_points, T = M K; 1:
 * three_interpolate, T = 3 N1): given that much, the two entry points below sum a target from 256-byte rows instead of
 * 4-byte gathers (measured ~3 x faster on the feature tensors; bit-identical results).  Given only the size above they
 * take the gather form. */
size_t s4g_scatter_det_workspace_bytes_c(int64_t B, int64_t C, int64_t N, int64_t T, int weighted);
int s4g_group_points_backward_det_f32(const float *gout_bcmk, const int64_t *idx_bmk, int64_t B, int64_t C,
                                      int64_t N, int64_t M, int64_t K, float *gin_bcn, void *ws,
                                      size_t ws_bytes, s4g_stream_t stream);
int s4g_three_interpolate_backward_det_f32(const float *gout_bcn1, const int64_t *idx_bn3, const float *w_bn3,
                                           int64_t B, int64_t C, int64_t N2, int64_t N1, float *gin_bcn2,
                                           void *ws, size_t ws_bytes, s4g_stream_t stream);

/* Inverse-distance weights of FeatureInterpolator.forward
 * (modules.py:118-120): w = (1/max(d2,eps)) / sum_k (1/max(d2,eps)). */
int s4g_interp_weights_f32(const float *d2_bn3, int64_t B, int64_t N1,
                           float eps, float *w_bn3, s4g_stream_t stream);


/* ---------------------------------------------------------------------------
 * Inference fast path: the shared-MLP contraction on the fp32 matrix cores.
 *
 * One launch computes, for `groups` independent problems (blockIdx.y),
 *     Y[p][n] = act( sum_k A[p][k] * W[n][k] + bias[n] ),  p < P, n < Cout
 * with BatchNorm already folded into W / bias by the caller and every
 * activation stored channels-last ([position][channel], fp32).  It replaces
 * the per-layer Conv{1,2}d(k=1) -> BatchNorm -> ReLU of
 * network_models/nn_utils/conv.py:28-34,68-74 together with the tensor
 * plumbing around it in pointnet2_utils/modules.py (group + concat :42-50,
 * max over neighbours :242-243, interpolate + concat :118-127).
 *
 * loader  S4G_GEMM_LOAD_PLAIN   A row p = A + p*lda + a_coff + g*a_gcol
 *         S4G_GEMM_LOAD_GATHER  A row p = [ feat[b*N + gidx[p]][0..Cf) |
 *                                xyz[b,:,gidx[p]] - ctr[b,:,m] | 0.. ]
 *                                (K order [feat, xyz]; W permuted to match)
 *         S4G_GEMM_LOAD_INTERP  A row p = [ sum_k nw[p,k]*sparse[b*N2+nidx[p,k]] |
 *                                dense[p][0..C1) ]
 *         S4G_GEMM_LOAD_GATHER_MLP1  first xyz-only SA layer folded into the loader:
 *                                A row p = relu(W1 . (xyz[b,:,gidx[p]] - ctr[b,:,m]) + b1),
 *                                mlp1_w = Cin x (wx, wy, wz, bias) fp32
 *         S4G_GEMM_LOAD_GATHER_ADD  first SA layer of a level WITH features, applied to the
 *                                features before the grouping (it is linear; feat = F =
 *                                W_feat . features, (B*N, Cf = Cin) channels-last):
 *                                A row p = relu(F[b*N + gidx[p]] + W_xyz . (xyz - ctr) + b1),
 *                                mlp1_w = Cin x (wx, wy, wz, bias); a_amax bounds |F| and
 *                                a_amax_floor the xyz + bias part (the two are ADDED)
 *         S4G_GEMM_LOAD_INTERP_ADD  first FP layer applied before the interpolation (linear):
 *                                A row p = relu(dense[p] + loader_bias + sum_k nw[p,k] *
 *                                sparse[b*N2 + nidx[p,k]]), sparse = W_a . sparse features and
 *                                dense = W_b . skip features (or NULL), both (rows, C2 = Cin);
 *                                a_amax / a_amax2 / a_amax_floor bound the three terms (ADDED)
 * epilogue S4G_GEMM_EPI_STORE   out[p*ldc + c_coff + g*c_gcol + n]
 *          S4G_GEMM_EPI_MAX     out[(p/K)*ldc + c_coff + n] = max over the K
 *                               consecutive rows of a group (K in 16,32,64)
 *          S4G_GEMM_EPI_CHANNEL_FIRST  (B,C,N) tensors cf_ptr[h], channel
 *                               ranges cf_start[h]..cf_start[h+1], sigmoid on
 *                               channels >= cf_sigmoid_from
 * W is [groups][Cout][Kpad] with Kpad % 8 == 0 (zero padded), bias
 * [groups][Cout].  Cf, C2, lda, a_coff, C1 must be multiples of 4.
 * ------------------------------------------------------------------------- */
#define S4G_GEMM_LOAD_PLAIN 0
#define S4G_GEMM_LOAD_GATHER 1
#define S4G_GEMM_LOAD_INTERP 2
#define S4G_GEMM_LOAD_GATHER_MLP1 3
#define S4G_GEMM_LOAD_GATHER_ADD 4
#define S4G_GEMM_LOAD_INTERP_ADD 5
#define S4G_GEMM_EPI_STORE 0
#define S4G_GEMM_EPI_MAX 1
#define S4G_GEMM_EPI_CHANNEL_FIRST 2
#define S4G_GEMM_FP32 0
#define S4G_GEMM_BF16X3 1
#define S4G_GEMM_BF16 2 /* reduced precision: one bf16 product, fp32 accumulate */
#define S4G_GEMM_F16X2 3 /* fp32-class: two fp16 planes per operand, three products */

typedef struct s4g_gemm_desc {
  int32_t loader, epilogue, groups, relu;
  int32_t P, Cin, Kpad, Cout;
  const float *W;
  const float *bias;
  int32_t w_gstride, b_gstride; /* elements between groups */
  /* PLAIN */
  const float *A;
  int32_t lda, a_coff, a_gcol;
  /* GATHER */
  const int32_t *gidx; /* (B*M*K) neighbour index inside its scene */
  const float *feat;   /* (B*N, Cf) channels-last or NULL when Cf == 0 */
  const float *xyz;    /* (B,3,N) */
  const float *ctr;    /* (B,3,M) */
  int32_t Cf, N, M, K;
  /* INTERP */
  const int32_t *nidx; /* (B*N1, 3) */
  const float *nw;     /* (B*N1, 3) */
  const float *sparse; /* (B*N2, C2) */
  const float *dense;  /* (B*N1, C1) or NULL when C1 == 0 */
  int32_t C2, C1, N2, N1;
  /* output */
  float *out;
  int32_t ldc, c_coff, c_gcol;
  float *cf_ptr[4];
  int32_t cf_start[5];
  int32_t cf_sigmoid_from, cf_N;
  /* arithmetic: S4G_GEMM_FP32 = v_mfma_f32_32x32x2_f32 on W (exact fp32 fma
   * chain); S4G_GEMM_BF16X3 = each fp32 operand split exactly into three bf16
   * numbers, six v_mfma_f32_32x32x16_bf16 per step, fp32 accumulate (drops
   * only terms below 2^-24 |a||b|); S4G_GEMM_BF16 = the hi planes only (plain
   * bf16 inputs, fp32 accumulate: reduced precision, NOT within the 1e-4 bar,
   * for the bf16 roofline configuration).  W_bf16x3 is [3][groups][Cout][Kpad16]
   * bf16 (hi, mid, lo planes of W), Kpad16 % 16 == 0. */
  int32_t precision, Kpad16;
  const void *W_bf16x3;
  const float *mlp1_w; /* GATHER_MLP1: (Cin, 4) = wx, wy, wz, bias */
  /* S4G_GEMM_F16X2 (ABI >= 2): every fp32 operand x is scaled by a power of two
   * s into fp16's range and split x*s = x1 + x2 (two fp16, round-to-nearest:
   * 22 significand bits); three v_mfma_f32_32x32x16_f16 per step evaluate
   * a1*w1 + (a1*w2 + a2*w1) with fp32 accumulation -- the error of a plain
   * fp32 dot product, at half the MFMA count of BF16X3.
   *   W_f16x2      [2][groups][Cout][Kpad16] fp16 planes of W[n][:] / w_inv_scale[n]
   *   w_inv_scale  [groups][Cout] power-of-two 1/s per output channel
   *   a_amax(2)    NULL or 64 floats whose maximum bounds |A| (two pointers: the
   *                INTERP loader reads two tensors); a_amax_floor >= 0 is a host
   *                side bound joined with them (ball radius for the GATHER xyz
   *                columns, the MLP1 bound) -- at least one must be positive
   *   out_amax     NULL or 64 uint32 slots (zeroed by the caller before the
   *                launch): receives atomicMax(bits of max |out|), the a_amax of
   *                the next layer. */
  const void *W_f16x2;
  const float *w_inv_scale;
  const float *a_amax, *a_amax2;
  float a_amax_floor;
  float *out_amax;
  /* optional: the same two fp16 planes in MFMA-fragment order,
   * [groups][Cout/32][Kpad16/16][2 planes][64 lanes][8 halves] with lane = 32*(k/8 % 2) + n % 32
   * (needs Cout % 32 == 0).  When given, launches with Kpad16 % 64 == 0,
   * Cout % 128 == 0 and a short contraction (the A panel of 64 or 128 positions
   * fits LDS) use the resident-A kernel, which streams W fragments straight into
   * the matrix-core operand registers. */
  const void *W_f16x2_frag;
  /* ABI >= 4: with precision S4G_GEMM_BF16 the three *_f16x2_frag pointers hold ONE bf16 plane
   * each in the same fragment order ([groups][Cout/32][Kpad16/16][64 lanes][8 bf16]) and select the
   * single-product form of the fused chains below (no scales: w*_inv_scale, a_amax*, out_amax are
   * ignored) -- the reduced-precision roofline configuration.
   * ABI >= 3, optional: a SECOND layer fused behind this one (S4G_GEMM_F16X2, loader PLAIN
   * GATHER_MLP1, GATHER_ADD or INTERP_ADD, Kpad16 == Cout == C with C = 128, 256 or 512 -- for the plain
   * loader + STORE also Kpad16 == 2 C == 512: the first layer then runs through two panel loads
   * --, Cout2 % 64 == 0; epilogue MAX
   * with K == 64 and groups == 1, or STORE with any group count -- W2 / w2_inv_scale / bias2
   * then hold `groups` blocks like their first-layer counterparts): the launch computes
   *   out = epilogue(relu2(bias2 + W2 . relu(bias + W . A)))
   * with the C-channel intermediate kept in LDS (split with a per-tile power-of-two scale).
   * W2_f16x2_frag / w2_inv_scale / bias2 describe W2 (Cout2 x C) like W_f16x2_frag /
   * w_inv_scale / bias describe W; out, ldc, c_coff, out_amax refer to the final output. */
  const void *W2_f16x2_frag;
  const float *w2_inv_scale;
  const float *bias2;
  int32_t Cout2, relu2;
  /* optional THIRD layer (then Cout2 == C: layer 2's output stays in LDS as well and the
   * epilogue / out / ldc / out_amax describe layer 3, Cout3 % 64 == 0). */
  const void *W3_f16x2_frag;
  const float *w3_inv_scale;
  const float *bias3;
  int32_t Cout3, relu3;
  const float *loader_bias; /* S4G_GEMM_LOAD_INTERP_ADD: Cin floats */
  /* ABI >= 4: activation maxima PER SCENE.  rows_per_scene > 0: a_amax / a_amax2 / out_amax are
   * [P / rows_per_scene][64] slot rows and loader row p reads / feeds row p / rows_per_scene, so a
   * scene's power-of-two scales -- and its results -- do not depend on the other scenes of the
   * batch (a tile that straddles scenes joins their rows); 0: one 64-slot row for all rows. */
  int32_t rows_per_scene;
  /* ABI >= 5, optional, S4G_GEMM_LOAD_GATHER_MLP1: (P, 4) fp32 rows (xyz[b,:,gidx[p]] - ctr[b,:,m], 0)
   * as s4g_group_rel_xyz_i32 writes them.  The loader then reads one coalesced 16-byte record per
   * row instead of following gidx into the cloud (two dependent round trips at the head of every
   * workgroup); gidx / xyz / ctr are not read.  Same values, same results. */
  const float *rel_xyz4;
  /* ABI >= 8, optional, both or neither; only S4G_GEMM_LOAD_GATHER_MLP1 + rel_xyz4 + a fused second
   * layer (W2_f16x2_frag) + S4G_GEMM_EPI_MAX with K == 64 and relu2: the DISTINCT-row form.  rel_xyz4 then
   * holds what s4g_group_rel_xyz_unique_i32 wrote -- per centroid only the rows ball_query did not pad
   * (ball_query_kernel.cu:64-67 repeats the first hit; modules.py:243's max over the neighbours cannot
   * see the copies) -- scene b's rows at b * rows_per_scene .. + seg_rows[b] (a multiple of 256, the
   * tallest tile; rows_per_scene % 256 == 0), seg4[row / 4] the OUTPUT row (b M + m) of every group of
   * four rows, -1 for filler.  `out` must be zero-filled by the caller: a centroid's pieces are merged
   * with an unsigned atomicMax on the post-ReLU values.  Same maxima as the 64-row form; the hidden
   * layer's per-tile power-of-two scales see other rows, so outputs agree to fp32 round-off, not bitwise. */
  const int32_t *seg4;
  const int32_t *seg_rows;
  /* ABI >= 8, optional: a SECOND output tensor for a plain single layer (loader PLAIN, epilogue STORE, groups 1,
   * no fused layers; f16x2 / bf16 precision): two layers that read the same input run as one launch with W, bias
   * and scales concatenated along Cout -- output channels [0, split_n) go to `out` (row stride ldc, c_coff 0),
   * channels [split_n, Cout) to `out2` (row stride ldc2, column n - split_n) and their per-scene maxima to
   * out_amax2.  split_n and Cout multiples of 256.  (The network's first SA layer on features and the first FP
   * layer on the skip features read the same level: modules.py:242 / :505.) */
  float *out2;
  int32_t ldc2, split_n;
  float *out_amax2;
} s4g_gemm_desc_t;

int s4g_mlp_gemm_f32(const s4g_gemm_desc_t *desc, s4g_stream_t stream);

/* ---------------------------------------------------------------------------
 * The four per-point heads as ONE launch (ABI >= 4).  Replaces, for inference,
 * PointNet2_tcls.py:126-140: mlp_seg / mlp_R / mlp_t / mlp_movable (four SharedMLP stacks
 * C -> H0 -> H1 -> H2 -> H3 over the SAME (B, C, N) input, definitions :83-95) and their
 * Conv1d logit layers (+ Sigmoid on movable_logit), BatchNorm folded by the caller.  A workgroup
 * keeps the 64 x C input panel and every hidden activation in LDS; only X and the four
 * (B, c_h, N) outputs touch HBM.  Shipped widths only: C = 256, H = (512, 256, 256, 128).
 *   precision    S4G_GEMM_F16X2 (fp32-class) or S4G_GEMM_BF16 (one bf16 plane, reduced precision)
 *   W_frag[l]    layer l's planes in MFMA-fragment order ([Cout/32][K/16][planes][64][8]):
 *                l = 0: the four first layers stacked (4 H0 x C); l = 1..3: (4, H_l, H_{l-1});
 *                l = 4: the logit layers zero-padded to (4, 32, H3)
 *   bias[l], w_inv_scale[l]  per output channel, same stacking (w_inv_scale: F16X2 only)
 *   out[h], channels[h]      head h's (B, channels[h], N) fp32 tensor; sigmoid_head = index of the
 *                head whose logits pass through a sigmoid (-1: none)
 *   a_amax / a_amax_floor / rows_per_scene   bound |X| per scene (F16X2), as in s4g_gemm_desc_t
 * ------------------------------------------------------------------------- */
typedef struct s4g_heads_desc {
  int32_t precision, P, N, ldx;
  int32_t C, H0, H1, H2, H3;
  const float *X; /* (P, ldx >= C) channels-last */
  const void *W_frag[5];
  const float *bias[5];
  const float *w_inv_scale[5];
  float *out[4];
  int32_t channels[4];
  int32_t sigmoid_head;
  const float *a_amax;
  float a_amax_floor;
  int32_t rows_per_scene;
  /* ABI >= 6, optional (pre_W_frag[0] != NULL): the tail of the LAST feature-propagation level in
   * front of the heads, in the same launch -- PointnetFPModule.forward (pointnet2_utils/modules.py:
   * 498-507) of fp_modules[2] with its first layer already applied to the sparse features
   * (S4G_GEMM_LOAD_INTERP_ADD's algebra): the workgroup's input panel is formed as
   *     X0[p] = relu(sum_k pre_nw[p][k] * pre_sparse[b N2 + pre_nidx[p][k]] (+ pre_dense[p]) + pre_lbias)
   * and two C -> C layers (pre_W_frag / pre_bias / pre_w_inv_scale [0..1], ReLU each) run on it
   * inside LDS before the heads read it; X / ldx are then unused and a_amax / pre_a_amax2 bound
   * |pre_sparse| / |pre_dense| per scene (a_amax_floor: the bias bound, summed with them).  The
   * (P, C) feature tensor between fp_modules[2] and the heads never exists in HBM. */
  const void *pre_W_frag[2];
  const float *pre_bias[2];
  const float *pre_w_inv_scale[2];
  const int32_t *pre_nidx;   /* (P, 3) */
  const float *pre_nw;       /* (P, 3) */
  const float *pre_sparse;   /* (B N2, C) channels-last */
  const float *pre_dense;    /* (P, C) or NULL */
  const float *pre_lbias;    /* C */
  const float *pre_a_amax2;  /* per-scene maxima of pre_dense or NULL */
  int32_t pre_N2;
  /* ABI >= 9, optional: floats between two consecutive scenes' blocks of EVERY out[h].  0 = each out[h] is
   * its own contiguous (B, channels[h], N) tensor.  Non-zero (>= channels[h] N): the four heads are channel
   * slices of ONE packed (B, C_total, N) tensor -- out[h] = packed + first_channel_h * N, out_batch_stride =
   * C_total * N -- which is what the multi-GPU path all-gathers (dist.py), so no copy packs the outputs. */
  int64_t out_batch_stride;
  /* ABI >= 11, optional: bit h set = evaluate head h; 0 = all four.  Heads that are not evaluated leave their out[h]
   * untouched (it may be NULL).  A serving path that only decodes the best-scoring points runs the score head (bit 0) on
   * every point and the pose heads (bits 1..3) on the points it keeps: FusedPointNet2(..., topk=). */
  int32_t head_mask;
} s4g_heads_desc_t;

int s4g_heads_chain_f32(const s4g_heads_desc_t *desc, s4g_stream_t stream);

/* 1 when s4g_mlp_gemm_f32 has a fused-chain form (W2_f16x2_frag set) for this first-layer
 * loader, final epilogue, chain width C (= Cout = Cout2 of a three-layer chain) and first-layer
 * depth Kpad16; callers decide with it which layers to hand over as one launch. */
int s4g_gemm_chain_supported(int loader, int epilogue, int C, int Kpad16);

/* int32-index variants used by the fast path (same kernels and semantics as
 * s4g_ball_query_f32 / s4g_three_nn_f32; the int64 API tensors are an
 * interface requirement of the reference, not of the hardware).
 * s4g_three_nn_weights_i32 also applies the inverse-distance weights of
 * modules.py:118-120 and does not write the distances. */
int s4g_ball_query_i32(const float *xyz_b3n, const float *ctr_b3m, int64_t B,
                       int64_t N, int64_t M, float radius, int64_t K,
                       int32_t *idx_bmk, int32_t *cnt_bm, void *ws,
                       size_t ws_bytes, int flags, s4g_stream_t stream);
int s4g_three_nn_weights_i32(const float *q_b3n1, const float *k_b3n2, int64_t B,
                             int64_t N1, int64_t N2, float eps, int32_t *idx_bn3,
                             float *w_bn3, void *ws, size_t ws_bytes, int flags,
                             s4g_stream_t stream);
/* Grid-accelerated variant of s4g_three_nn_weights_i32 for the fast path: keys are
 * binned into cells of edge `cell` (use the set-abstraction radius of the level
 * the keys came from), the queries are binned into the same cells by the same launch
 * and walked in cell order (a wave's 64 queries share their key rows), each searches
 * 27 cells, unanswered queries fall back to the index-order scan in the same call --
 * identical results for every input.  cell < 0 (round 4, what the fast path passes): the edge is chosen on
 * the device, 1.75 x the median distance from a key to its third-nearest other key over 64 sample keys (one
 * wave each) -- the SA radius is the right edge on surface-like clouds only.
 * Workspace: s4g_three_nn_grid_workspace_bytes(B, N1, N2) (keys' grid + fail list +
 * the binned queries: ~128 bytes per query + 0.6 MB per scene); N2 <= 65536. */
size_t s4g_three_nn_grid_workspace_bytes(int64_t B, int64_t N1, int64_t N2);
/* Diagnostic (ABI >= 8): byte offset, inside that workspace, of the fail list's header of int32 words -- word 0:
 * the queries the 27 cells could not answer in the last call (they took the all-keys scan), words 4 / 5: the
 * device-chosen 1 / edge and acceptance bound (floats) when cell < 0. */
size_t s4g_three_nn_grid_header_offset(int64_t B, int64_t N2);
int s4g_three_nn_weights_grid_i32(const float *q_b3n1, const float *k_b3n2, int64_t B,
                                  int64_t N1, int64_t N2, float eps, float cell,
                                  int32_t *idx_bn3, float *w_bn3, void *ws,
                                  size_t ws_bytes, int flags, s4g_stream_t stream);

/* s4g_three_nn_f32's outputs (int64 indices, squared distances) through the same grid:
 * for operator-API callers.  cell > 0: the caller names the cell edge; cell < 0: the call
 * derives one on the device (1.75 x the median third-neighbour distance of 64 sample keys,
 * no host read).  Same workspace; identical results for any cell. */
int s4g_three_nn_grid_f32(const float *q_b3n1, const float *k_b3n2, int64_t B, int64_t N1,
                          int64_t N2, float cell, int64_t *idx_bn3, float *d2_bn3, void *ws,
                          size_t ws_bytes, int flags, s4g_stream_t stream);

/* QueryGrouper's operator pair in one pass (modules.py:39-42):
 * ball_query + group_points(xyz, index).  Same outputs as calling
 * s4g_ball_query_f32 then s4g_group_points_f32 with C = 3: index (B,M,K) int64,
 * count (B,M) int64, grouped (B,3,M,K) fp32 (NOT centroid-subtracted). */
int s4g_query_group_f32(const float *xyz_b3n, const float *ctr_b3m, int64_t B,
                        int64_t N, int64_t M, float radius, int64_t K,
                        int64_t *idx_bmk, int64_t *cnt_bm, float *grouped_b3mk,
                        void *ws, size_t ws_bytes, int flags, s4g_stream_t stream);

/* group_points(xyz, index) - centroid as one 16-byte record per neighbour (modules.py:42-44:
 * group_xyz = group_points(xyz, index); group_xyz -= new_xyz.unsqueeze(-1)), for the first SA
 * layer's loader (s4g_gemm_desc_t.rel_xyz4): out (B*M*K, 4) = (x - cx, y - cy, z - cz, 0), each
 * difference one rounded fp32 subtraction. */
int s4g_group_rel_xyz_i32(const float *xyz_b3n, const float *ctr_b3m, const int32_t *idx_bmk,
                          int64_t B, int64_t N, int64_t M, int64_t K, float *rel_pk4,
                          s4g_stream_t stream);

/* The same records without ball_query's padding copies (ABI >= 8; see s4g_gemm_desc_t.seg4).
 * cnt_bm = ball_query's count output (int32).  Centroid m of scene b contributes
 * c4 = round_up(max(cnt, 1), 4) rows (slots cnt .. c4-1 are copies of slot 0, so they are valid
 * padding; an empty ball keeps its K copies of point 0 as 4 rows), centroids back to back from
 * row b M K:
 *   rel_pk4       (B M K, 4) capacity, rows as above; rows between a scene's last centroid and
 *                 rows_b[b] are zero records
 *   seg4          (B M K / 4) int32: b M + m per group of 4 rows, -1 for the filler rows
 *   row_start_bm  (B, M) int32: first row of every centroid relative to its scene's base
 *   rows_b        (B) int32: rows of scene b, rounded up to 256
 * A scene whose rows would exceed 7/8 of M K keeps the PLAIN layout instead (centroid m at row m K, all K
 * slots, rows_b[b] == M K): the segmented epilogue would cost more than the few copies save; the
 * contraction takes its 64-row epilogue for such a scene.  The choice depends on the scene alone.
 * K % 4 == 0 and (M K) % 256 == 0, else S4G_EUNSUPPORTED. */
int s4g_group_rel_xyz_unique_i32(const float *xyz_b3n, const float *ctr_b3m, const int32_t *idx_bmk,
                                 const int32_t *cnt_bm, int64_t B, int64_t N, int64_t M, int64_t K,
                                 float *rel_pk4, int32_t *seg4, int32_t *row_start_bm, int32_t *rows_b,
                                 s4g_stream_t stream);

/* FPS + centroid gather in one call: idx (B,M) int32 and ctr (B,3,M) planar. */
int s4g_fps_gather_i32(const float *xyz_b3n, int64_t B, int64_t N, int64_t M,
                       int32_t *idx_bm, float *ctr_b3m, void *ws, size_t ws_bytes,
                       int flags, s4g_stream_t stream);

/* FPS of the NEXT set-abstraction level without running it (round 3).
 *
 * modules.py:80-83 samples each level from the previous level's centroids, which are the picks of
 * the previous FPS in pick order.  FPS over such a set, started at its element 0 (sampling_kernel.cu:
 * 66-68), re-picks the set's own prefix: element k was the farthest point of the WHOLE cloud from
 * {0..k-1}, the set contains it, so it is also the farthest point of the set -- as long as no other
 * element ties with it, which is the only case in which the tie rule of sampling_kernel.cu:87-105
 * matters.  s4g_fps_gather_ex_i32 = s4g_fps_gather_i32 with two optional extras:
 *   dist_bm (B,M): the min-distance every pick had when it was taken (+inf for pick 0);
 *                  S4G_EUNSUPPORTED (nothing launched) if this size's kernel cannot report it;
 *   run_b (B):     run_b[b] == 0 = "scene b's result is the identity prefix": idx = 0..M-1, ctr =
 *                  the first M input points, nothing sampled (only honoured by the kernels for
 *                  N <= 10 240; larger inputs are sampled regardless).
 * s4g_fps_prefix_check_f32 proves or refutes the prefix property per scene: ctr_b3m (B,3,M1) and
 * dist_bm (B,M1) from the previous level's call; run_b[b] = 0 iff, at every step k < M2, no element
 * other than k reaches element k's distance (same fp32 arithmetic as the sampler; FMAD flag as
 * usual).  A scene with run_b[b] = 1 is then sampled for real, so the indices are the reference's
 * in every case.  One check over M2 steps also covers deeper levels that sample a prefix of this
 * one (fewer steps over fewer elements).  5 120 -> 1 024: 20 us instead of 0.87 ms. */
int s4g_fps_gather_ex_i32(const float *xyz_b3n, int64_t B, int64_t N, int64_t M, int32_t *idx_bm,
                          float *ctr_b3m, float *dist_bm, const int32_t *run_b, void *ws,
                          size_t ws_bytes, int flags, s4g_stream_t stream);
int s4g_fps_prefix_check_f32(const float *ctr_b3m, const float *dist_bm, int64_t B, int64_t M1,
                             int64_t M2, int32_t *run_b, int flags, s4g_stream_t stream);

/* Diagnostic entry (tests): the pruned FPS kernels' pre-pass on its own -- ONE launch, one
 * workgroup per scene: bounding box, a 15-bit cell key per point (2-D Hilbert curve over the two
 * long axes for thin clouds, extent-dealt Morton bits otherwise), LDS counting sort.  Outputs
 * perm (B,N): a permutation of 0..N-1 per scene (cell order; the order INSIDE a cell is not
 * reproducible), and gbox (B,G,6): (min x,y,z, max x,y,z) of every run of 64 consecutive perm
 * entries; G >= ceil(N / 64) groups are written (groups past the end hold NaN).  N <= 65 535.
 * The FPS result never depends on the permutation (sampling_kernel.cu:49-119 has no such step);
 * it only decides how many groups a pick has to revisit. */
int s4g_fps_prepass_f32(const float *xyz_b3n, int64_t B, int64_t N, int64_t G, int32_t *perm_bn,
                        float *gbox_bg6, s4g_stream_t stream);

/* ---------------------------------------------------------------------------
 * Next row: pose decode right after the network.
 * s4g_expected_score_f32: softmax over the C score classes of (B,C,N) logits,
 *   score = sum_c values[c] * softmax[c]   (utils/file_logger_cls.py:34-36,66-68;
 *   grasp_detector.py:145-149 with its own `values`).
 * s4g_decode_poses_f32: for the selected point indices sel (B,K): row-major
 *   R from frame_R (B,9,N), tau = sum_c t_bins[c]*softmax(frame_t)[c],
 *   t = -tau*R[:,0] + p, Gram-Schmidt -> H (B,K,4,4) row-major
 *   (utils/file_logger_cls.py:38-47,203-218; grasp_detector.py:124-135,176-180).
 * ------------------------------------------------------------------------- */
int s4g_expected_score_f32(const float *logits_bcn, int64_t B, int64_t C, int64_t N,
                           const float *values_c, float *score_bn, s4g_stream_t stream);
int s4g_decode_poses_f32(const float *xyz_b3n, const float *frame_R_b9n,
                         const float *frame_t_btn, const int64_t *sel_bk, int64_t B,
                         int64_t N, int64_t K, int64_t TC, const float *t_bins,
                         float *H_bk44, s4g_stream_t stream);

/* Next row f2: batched gripper-vs-cloud collision counts, replaces the per-pose
 * loop over CloudCollisionChecker.view_non_collision
 * (cloud_processor/view_collision_checker.py:37-65, grasp_detector.py:216-234).
 * g2l = global->gripper 4x4 row-major per pose; gripper6 (HOST pointer) =
 * {FINGER_LENGTH, BOTTOM_LENGTH, HALF_HAND_THICKNESS, HALF_BOTTOM_WIDTH,
 *  HALF_BOTTOM_SPACE, BACK_COLLISION_MARGIN} (configs/gripper_config.py:10-21,
 * processing_config.py:39); counts (B,K,2) int32 = {behind the palm, inside
 * the finger volumes}. */
int s4g_collision_counts_f32(const float *xyz_b3n, const float *g2l_bk44, int64_t B,
                             int64_t N, int64_t K, const float *gripper6,
                             int32_t *counts_bk2, s4g_stream_t stream);
/* ABI 12: the same over best-first pose lists padded to K rows: only the first pose_count_b[b] (DEVICE int64, (B,); NULL =
 * all K) rows of scene b are poses -- the rest get zero counts without scanning the cloud (detector.GraspDetector: K =
 * 2 048 rows, a few dozen to a few hundred of them poses; the counts never visit the host) -- a workgroup walks several
 * poses, so the launch stays small.  invert_se3 = 1: the matrices are the POSES themselves (gripper -> global) and the
 * kernel forms their analytic SE(3) inverse [R^T | -R^T t] in fp32, the form grasp_detector.py:219 feeds the check
 * (torch_batch_transformation_inv, utils/math_utils.py:26-40), instead of the caller. */
int s4g_collision_counts_n_f32(const float *xyz_b3n, const float *g2l_bk44, int64_t B, int64_t N,
                               int64_t K, const float *gripper6, const int64_t *pose_count_b,
                               int invert_se3, int32_t *counts_bk2, s4g_stream_t stream);

/* ---- next row f3: cloud pre-processing on device -------------------------
 * Single-scene passes in front of the network (reference
 * grasp_proposal/cloud_processor/cloud_processor.py:12-42, constants
 * configs/processing_config.py:17-23, caller grasp_detector.py:94-105).  The
 * reference delegates voxelisation and outlier removal to open3d (>= 0.12,
 * absent here) and discards their results; the semantics below restate open3d's
 * published algorithms (oracle/preprocess.py; parity unpinned).
 *
 * s4g_crop_indices_f32: CloudPreProcessor.filter_work_space (:12-29).
 *   workspace6 = {lo_x, hi_x, lo_y, hi_y, lo_z, hi_z} (HOST pointer); index_n
 *   receives the ascending indices of the points strictly inside the box,
 *   *count (device) their number.
 * s4g_voxel_down_sample_f32: CloudPreProcessor.voxelize (:38-41) = open3d
 *   VoxelDownSample.  origin3 / dims3 are HOST pointers: origin = min(points) -
 *   voxel/2, dims = cells per axis (product < 2^32).  out_3n is (3, N) with row
 *   stride N: the first *count columns hold the cell means (double sum in point
 *   order, rounded once), cells in ascending (iz, iy, ix) order.
 * s4g_radius_outlier_mask_f32: CloudPreProcessor.remove_outliers (:31-36) = open3d
 *   RemoveRadiusOutliers.  keep_n[j] = 1 iff more than nb_points points (j itself
 *   included) lie at squared distance < radius^2 (canonical fp32 arithmetic).
 *   Workspace: s4g_radius_outlier_workspace_bytes(N) (0 for N > 65536: scan). */
/* ABI 12: the library's own device primitives behind the deterministic scatters and the voxel down-sample (round 6:
 * csrc/radix_sort.hip replaces rocPRIM's): a STABLE least-significant-digit radix sort of n (uint32 key, uint32 value)
 * pairs on the low `bits` key bits (8 bits per pass; equal keys keep their input order) -- result in keys_out / vals_out,
 * keys_in / vals_in are clobbered -- and an int32 exclusive scan (out must not alias in).  No counterpart in the
 * reference (which leaves order to atomicAdd / open3d's hash map). */
size_t s4g_sort_pairs_workspace_bytes(int64_t n);
int s4g_sort_pairs_u32(uint32_t *keys_in, uint32_t *vals_in, int64_t n, int bits, uint32_t *keys_out,
                       uint32_t *vals_out, void *ws, size_t ws_bytes, s4g_stream_t stream);
size_t s4g_exclusive_scan_workspace_bytes(int64_t n);
int s4g_exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, void *ws, size_t ws_bytes,
                           s4g_stream_t stream);

int s4g_crop_indices_f32(const float *xyz_3n, int64_t N, const float *workspace6,
                         int32_t *index_n, int32_t *count, s4g_stream_t stream);
size_t s4g_voxel_down_sample_workspace_bytes(int64_t N);
int s4g_voxel_down_sample_f32(const float *xyz_3n, int64_t N, float voxel,
                              const float *origin3, const int32_t *dims3, float *out_3n,
                              int32_t *count, void *ws, size_t ws_bytes, s4g_stream_t stream);
size_t s4g_radius_outlier_workspace_bytes(int64_t N);
int s4g_radius_outlier_mask_f32(const float *xyz_3n, int64_t N, float radius,
                                int32_t nb_points, uint8_t *keep_n, void *ws, size_t ws_bytes,
                                int flags, s4g_stream_t stream);

/* ---------------------------------------------------------------------------
 * The operators in DOUBLE (ABI >= 8).  The reference's extension dispatches every kernel over float and
 * double (AT_DISPATCH_FLOATING_TYPES: sampling_kernel.cu:148-167, ball_query_kernel.cu:116-128,
 * grouping_kernel.cu:48-51,136-150, interpolate_kernel.cu:114-126,212-232,317-338).  Same layouts, index
 * types, tie rules and padding as the *_f32 entry points; plain kernels (S4G never uses double).
 *   s4g_fps_f64: ws = (B, N) doubles (the reference's `temp`), ws_bytes >= 8 B N.
 *   s4g_ball_query_f64: `radius` is a C float as in ball_query.h and is cast to double before squaring.
 *   gather_points = group_points with K == 1.  Inverse-distance weights: torch ops in the caller.
 * ------------------------------------------------------------------------- */
int s4g_fps_f64(const double *xyz_b3n, int64_t B, int64_t N, int64_t M, int64_t *idx_bm, void *ws,
                size_t ws_bytes, int flags, s4g_stream_t stream);
int s4g_ball_query_f64(const double *xyz_b3n, const double *ctr_b3m, int64_t B, int64_t N, int64_t M,
                       float radius, int64_t K, int64_t *idx_bmk, int64_t *cnt_bm, int flags,
                       s4g_stream_t stream);
int s4g_three_nn_f64(const double *query_b3n1, const double *key_b3n2, int64_t B, int64_t N1, int64_t N2,
                     int64_t *idx_bn3, double *d2_bn3, int flags, s4g_stream_t stream);
int s4g_group_points_f64(const double *in_bcn, const int64_t *idx_bmk, int64_t B, int64_t C, int64_t N,
                         int64_t M, int64_t K, double *out_bcmk, s4g_stream_t stream);
int s4g_group_points_backward_f64(const double *gout_bcmk, const int64_t *idx_bmk, int64_t B, int64_t C,
                                  int64_t N, int64_t M, int64_t K, double *gin_bcn, s4g_stream_t stream);
int s4g_three_interpolate_f64(const double *feat_bcn2, const int64_t *idx_bn3, const double *w_bn3,
                              int64_t B, int64_t C, int64_t N2, int64_t N1, double *out_bcn1, int flags,
                              s4g_stream_t stream);
int s4g_three_interpolate_backward_f64(const double *gout_bcn1, const int64_t *idx_bn3,
                                       const double *w_bn3, int64_t B, int64_t C, int64_t N2, int64_t N1,
                                       double *gin_bcn2, s4g_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* S4G_OPS_H_ */
