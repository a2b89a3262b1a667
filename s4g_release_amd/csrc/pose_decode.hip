// Pose decode on the device: the step right after the network in the
// reference's callers.
//
// Restates, per point / per selected grasp,
//   utils/file_logger_cls.py:34-47   softmax over score classes, frame_R as
//                                     row-major 3x3, softmax over the 4 t bins
//                                     times (0.08, 0.06, 0.04, 0.02),
//                                     t = -tau * R[:,0] + p
//   utils/file_logger_cls.py:66-68   expected score  sum_c value[c] * softmax[c]
//                                     (value = linspace(0,1,C+1)[:-1]; the
//                                     detector uses [1:], grasp_detector.py:146-148)
//   utils/file_logger_cls.py:203-218 Gram-Schmidt: x = R[:,0]/|.|,
//   grasp_detector.py:124-135        y = R[:,1] - (x.y) x, normalised, z = x cross y,
//                                     H = [x y z | t; 0 0 0 1]
// Two tiny HBM-bound kernels around a top-K selection, so only K poses per
// scene (K * 18 floats) instead of 21 channels x N points leave the GPU / go
// into the all-gather.
#include "s4g_common.h"

namespace s4g {

// expected score per point; score logits are (B, C, N) channel-first, C <= 8
__global__ __launch_bounds__(256) void expected_score_kernel(
    const float* __restrict__ logits, int C, int N, const float* __restrict__ values,
    float* __restrict__ out) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* l = logits + (size_t)b * C * N + n;
  float mx = l[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, l[(size_t)c * N]);
  float den = 0.f, num = 0.f;
  for (int c = 0; c < C; ++c) {
    const float e = expf(l[(size_t)c * N] - mx);
    den += e;
    num += values[c] * e;
  }
  out[(size_t)b * N + n] = num / den;
}

// one thread per selected grasp: 4x4 row-major pose
__global__ __launch_bounds__(64) void decode_pose_kernel(
    const float* __restrict__ xyz, const float* __restrict__ frame_R,
    const float* __restrict__ frame_t, const int64_t* __restrict__ sel, int N, int K,
    int TC, const float* __restrict__ t_bins, float* __restrict__ H) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  const int n = (int)sel[(size_t)b * K + k];
  const float* R = frame_R + (size_t)b * 9 * N + n;     // R[i][j] = channel 3i + j
  float r[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) r[c] = R[(size_t)c * N];
  const float* tl = frame_t + (size_t)b * TC * N + n;
  float mx = tl[0];
  for (int c = 1; c < TC; ++c) mx = fmaxf(mx, tl[(size_t)c * N]);
  float den = 0.f, num = 0.f;
  for (int c = 0; c < TC; ++c) {
    const float e = expf(tl[(size_t)c * N] - mx);
    den += e;
    num += t_bins[c] * e;
  }
  const float tau = num / den;
  const float* p = xyz + (size_t)b * 3 * N + n;
  const float tx = -tau * r[0] + p[0], ty = -tau * r[3] + p[N], tz = -tau * r[6] + p[2 * (size_t)N];
  // Gram-Schmidt on the first two columns
  float x0 = r[0], x1 = r[3], x2 = r[6];
  const float xn = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
  x0 /= xn; x1 /= xn; x2 /= xn;
  float y0 = r[1], y1 = r[4], y2 = r[7];
  const float d = x0 * y0 + x1 * y1 + x2 * y2;
  y0 -= d * x0; y1 -= d * x1; y2 -= d * x2;
  const float yn = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
  y0 /= yn; y1 /= yn; y2 /= yn;
  const float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
  float* h = H + ((size_t)b * K + k) * 16;
  h[0] = x0; h[1] = y0; h[2] = z0; h[3] = tx;
  h[4] = x1; h[5] = y1; h[6] = z1; h[7] = ty;
  h[8] = x2; h[9] = y2; h[10] = z2; h[11] = tz;
  h[12] = 0.f; h[13] = 0.f; h[14] = 0.f; h[15] = 1.f;
}

}  // namespace s4g

extern "C" int s4g_expected_score_f32(const float* logits_bcn, int64_t B, int64_t C, int64_t N,
                                      const float* values_c, float* score_bn,
                                      s4g_stream_t stream) {
  if (B < 0 || C <= 0 || N < 0 || B > 65535 || N >= (1ll << 31)) return S4G_EINVAL;
  if (B == 0 || N == 0) return S4G_OK;
  if (!logits_bcn || !values_c || !score_bn) return S4G_EINVAL;
  hipLaunchKernelGGL(s4g::expected_score_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)B),
                     dim3(256), 0, (hipStream_t)stream, logits_bcn, (int)C, (int)N, values_c,
                     score_bn);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" int s4g_decode_poses_f32(const float* xyz_b3n, const float* frame_R_b9n,
                                    const float* frame_t_btn, const int64_t* sel_bk, int64_t B,
                                    int64_t N, int64_t K, int64_t TC, const float* t_bins,
                                    float* H_bk44, s4g_stream_t stream) {
  if (B < 0 || N <= 0 || K < 0 || TC <= 0 || B > 65535 || N >= (1ll << 31)) return S4G_EINVAL;
  if (B == 0 || K == 0) return S4G_OK;
  if (!xyz_b3n || !frame_R_b9n || !frame_t_btn || !sel_bk || !t_bins || !H_bk44) return S4G_EINVAL;
  hipLaunchKernelGGL(s4g::decode_pose_kernel, dim3((unsigned)((K + 63) / 64), (unsigned)B), dim3(64),
                     0, (hipStream_t)stream, xyz_b3n, frame_R_b9n, frame_t_btn, sel_bk, (int)N,
                     (int)K, (int)TC, t_bins, H_bk44);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

// ---------------------------------------------------------------------------
// Batched gripper-vs-cloud collision counts: restates
// CloudCollisionChecker.view_non_collision
// (cloud_processor/view_collision_checker.py:37-65), which the detector calls
// once per pose in a Python loop with a host sync each (grasp_detector.py:216-234).
// One workgroup per pose streams the cloud once (HBM/L2-bound, 12 B per point)
// and counts the points behind the palm and inside the two finger volumes.
// ---------------------------------------------------------------------------
namespace s4g {

struct GripperBox {
  float finger_length, bottom_length, half_hand_thickness, half_bottom_width, half_bottom_space,
      back_margin;
};

constexpr int COLL_CHUNKS = 8;     // point ranges per scene (a workgroup scans one range)
constexpr int COLL_GX = 16;        // workgroups that share a scene's pose list (pose k belongs to workgroup k mod 16)
constexpr int COLL_U = 4;          // points per lane held in registers while the workgroup's poses pass over them
constexpr int COLL_SLOTS = 32;     // poses per pass (their matrices and counters live in LDS)

// Points outer, poses inner: a workgroup keeps 1 024 points in registers and runs ALL its poses over them before it
// loads the next 1 024 -- the cloud is read once per workgroup instead of once per pose.  (Round 6: with one workgroup
// per pose every candidate re-read the whole 48 902-point cloud from L2: ~500 candidates per scene x 16 scenes x 587 KB
// = 4.8 GB per call, 0.5 ms of a pipelined step.)
__global__ __launch_bounds__(256) void collision_counts_kernel(
    const float* __restrict__ xyz, const float* __restrict__ g2l, int N, int K, GripperBox g,
    int* __restrict__ counts, const int64_t* __restrict__ pose_count, int invert_se3) {
  __shared__ float gl[COLL_SLOTS][12];
  __shared__ int cnt[COLL_SLOTS][2];
  const int b = blockIdx.z, chunk = blockIdx.y, t = threadIdx.x, lane = t & 63;
  const float* px = xyz + (size_t)b * 3 * N;
  const int nc = (N + COLL_CHUNKS - 1) / COLL_CHUNKS;
  const int i_lo = chunk * nc, i_hi = min(N, i_lo + nc);
  int kmax = K;
  if (pose_count) kmax = (int)min((int64_t)K, max((int64_t)0, pose_count[b]));   // padding rows: never scanned (counts pre-zeroed)
  // this workgroup's poses: k = blockIdx.x + COLL_GX * j, j = 0, 1, ... ; COLL_SLOTS of them per pass
  for (int j0 = 0; blockIdx.x + COLL_GX * j0 < kmax; j0 += COLL_SLOTS) {
    __syncthreads();                                  // (the previous pass's tables have been read)
    if (t < COLL_SLOTS) {
      const int k = blockIdx.x + COLL_GX * (j0 + t);
      cnt[t][0] = cnt[t][1] = 0;
      if (k < kmax) {
        const float* G = g2l + ((size_t)b * K + k) * 16;   // row-major 4x4
        float g00 = G[0], g01 = G[1], g02 = G[2], g03 = G[3];
        float g10 = G[4], g11 = G[5], g12 = G[6], g13 = G[7];
        float g20 = G[8], g21 = G[9], g22 = G[10], g23 = G[11];
        if (invert_se3) {
          // the matrix is the POSE (gripper -> global): its analytic SE(3) inverse [R^T | -R^T t] in fp32
          // (torch_batch_transformation_inv, utils/math_utils.py:26-40, as grasp_detector.py:219 calls it) -- formed
          // here instead of by a batched 3x3 library GEMM per call (0.26 ms for 16 x 2 048 poses)
          const float tx = g03, ty = g13, tz = g23;
          const float r01 = g01, r02 = g02, r12 = g12;
          g01 = g10; g02 = g20; g12 = g21;
          g10 = r01; g20 = r02; g21 = r12;
          g03 = -__fadd_rn(__fadd_rn(__fmul_rn(g00, tx), __fmul_rn(g01, ty)), __fmul_rn(g02, tz));
          g13 = -__fadd_rn(__fadd_rn(__fmul_rn(g10, tx), __fmul_rn(g11, ty)), __fmul_rn(g12, tz));
          g23 = -__fadd_rn(__fadd_rn(__fmul_rn(g20, tx), __fmul_rn(g21, ty)), __fmul_rn(g22, tz));
        }
        gl[t][0] = g00; gl[t][1] = g01; gl[t][2] = g02; gl[t][3] = g03;
        gl[t][4] = g10; gl[t][5] = g11; gl[t][6] = g12; gl[t][7] = g13;
        gl[t][8] = g20; gl[t][9] = g21; gl[t][10] = g22; gl[t][11] = g23;
      }
    }
    __syncthreads();
    const int left = (kmax - 1 - (int)blockIdx.x) / COLL_GX + 1 - j0;     // poses of this workgroup from j0 on
    const int nslot = left < COLL_SLOTS ? left : COLL_SLOTS;
    for (int i0 = i_lo + t; i0 < i_hi + 256 * (COLL_U - 1); i0 += 256 * COLL_U) {
      float x[COLL_U], y[COLL_U], z[COLL_U];
      bool in[COLL_U];
#pragma unroll
      for (int u = 0; u < COLL_U; ++u) {
        const int i = i0 + 256 * u;
        in[u] = i < i_hi;
        const int ii = in[u] ? i : i_lo;
        x[u] = px[ii];
        y[u] = px[N + ii];
        z[u] = px[2 * (size_t)N + ii];
      }
      for (int sl = 0; sl < nslot; ++sl) {
        const float g00 = gl[sl][0], g01 = gl[sl][1], g02 = gl[sl][2], g03 = gl[sl][3];
        const float g10 = gl[sl][4], g11 = gl[sl][5], g12 = gl[sl][6], g13 = gl[sl][7];
        const float g20 = gl[sl][8], g21 = gl[sl][9], g22 = gl[sl][10], g23 = gl[sl][11];
        int nback = 0, nfing = 0;
#pragma unroll
        for (int u = 0; u < COLL_U; ++u) {
          const float lx = g00 * x[u] + g01 * y[u] + g02 * z[u] + g03;
          const float ly = g10 * x[u] + g11 * y[u] + g12 * z[u] + g13;
          const float lz = g20 * x[u] + g21 * y[u] + g22 * z[u] + g23;
          const bool close = (lx < g.finger_length) && (lx > -g.bottom_length);            // :39-40
          const bool zin = (lz < g.half_hand_thickness) && (lz > -g.half_hand_thickness);  // :44-45
          const bool back = in[u] && close && zin && (ly < g.half_bottom_width) && (ly > -g.half_bottom_width) &&
                            (lx < -g.back_margin);                                          // :47-49
          const bool fl = (ly < g.half_bottom_width) && (ly > g.half_bottom_space);        // :54-55
          const bool fr = (ly > -g.half_bottom_width) && (ly < -g.half_bottom_space);      // :56-57
          const bool fing = in[u] && close && zin && (fl || fr);                            // :59-60
          nback += __popcll(__ballot(back));      // wave-uniform
          nfing += __popcll(__ballot(fing));
        }
        if (lane == 0) {
          if (nback) atomicAdd(&cnt[sl][0], nback);
          if (nfing) atomicAdd(&cnt[sl][1], nfing);
        }
      }
    }
    __syncthreads();
    if (t < 2 * nslot) {
      const int sl = t >> 1, w = t & 1;
      const int k = blockIdx.x + COLL_GX * (j0 + sl);
      if (cnt[sl][w]) atomicAdd(counts + ((size_t)b * K + k) * 2 + w, cnt[sl][w]);
    }
  }
}

static int launch_collision(const float* xyz, const float* g2l, int64_t B, int64_t N, int64_t K, const float* gripper6,
                            const int64_t* pose_count, int invert_se3, int32_t* counts, hipStream_t st) {
  GripperBox g = {gripper6[0], gripper6[1], gripper6[2], gripper6[3], gripper6[4], gripper6[5]};
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * (size_t)B * (size_t)K, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(collision_counts_kernel, dim3(COLL_GX, COLL_CHUNKS, (unsigned)B), dim3(256), 0, st, xyz, g2l, (int)N, (int)K,
                     g, counts, pose_count, invert_se3);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

}  // namespace s4g

extern "C" int s4g_collision_counts_f32(const float* xyz_b3n, const float* g2l_bk44, int64_t B,
                                        int64_t N, int64_t K, const float* gripper6,
                                        int32_t* counts_bk2, s4g_stream_t stream) {
  if (B < 0 || N <= 0 || K < 0 || B > 65535 || N >= (1ll << 31)) return S4G_EINVAL;
  if (B == 0 || K == 0) return S4G_OK;
  if (!xyz_b3n || !g2l_bk44 || !gripper6 || !counts_bk2) return S4G_EINVAL;
  return s4g::launch_collision(xyz_b3n, g2l_bk44, B, N, K, gripper6, nullptr, 0, counts_bk2, (hipStream_t)stream);
}

// The same for best-first pose lists of which only the first pose_count_b[b] rows of scene b are poses (device counts:
// the caller never reads them on the host): the padding rows get zero counts and are never scanned.
// invert_se3 = 1: the matrices are the POSES (gripper -> global) and the kernel forms their analytic SE(3) inverse itself.
extern "C" int s4g_collision_counts_n_f32(const float* xyz_b3n, const float* g2l_bk44, int64_t B,
                                          int64_t N, int64_t K, const float* gripper6,
                                          const int64_t* pose_count_b, int invert_se3, int32_t* counts_bk2,
                                          s4g_stream_t stream) {
  if (B < 0 || N <= 0 || K < 0 || B > 65535 || N >= (1ll << 31) || (invert_se3 & ~1)) return S4G_EINVAL;
  if (B == 0 || K == 0) return S4G_OK;
  if (!xyz_b3n || !g2l_bk44 || !gripper6 || !counts_bk2) return S4G_EINVAL;   // (pose_count_b may be NULL: every row is a pose)
  return s4g::launch_collision(xyz_b3n, g2l_bk44, B, N, K, gripper6, pose_count_b, invert_se3, counts_bk2,
                               (hipStream_t)stream);
}
