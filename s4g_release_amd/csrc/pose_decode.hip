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

__global__ __launch_bounds__(256) void collision_counts_kernel(
    const float* __restrict__ xyz, const float* __restrict__ g2l, int N, int K, GripperBox g,
    int* __restrict__ counts, const int64_t* __restrict__ pose_count) {
  __shared__ int sback[4], sfing[4];
  const int b = blockIdx.y, k = blockIdx.x, t = threadIdx.x;
  if (pose_count && k >= pose_count[b]) {   // a padding row of a best-first list: no pose there, nothing to scan
    if (t < 2) counts[((size_t)b * K + k) * 2 + t] = 0;
    return;
  }
  const float* G = g2l + ((size_t)b * K + k) * 16;   // row-major 4x4 global -> gripper frame
  const float g00 = G[0], g01 = G[1], g02 = G[2], g03 = G[3];
  const float g10 = G[4], g11 = G[5], g12 = G[6], g13 = G[7];
  const float g20 = G[8], g21 = G[9], g22 = G[10], g23 = G[11];
  const float* px = xyz + (size_t)b * 3 * N;
  int nback = 0, nfing = 0;
  for (int i = t; i < N; i += 256) {
    const float x = px[i], y = px[N + i], z = px[2 * (size_t)N + i];
    const float lx = g00 * x + g01 * y + g02 * z + g03;
    const float ly = g10 * x + g11 * y + g12 * z + g13;
    const float lz = g20 * x + g21 * y + g22 * z + g23;
    const bool close = (lx < g.finger_length) && (lx > -g.bottom_length);            // :39-40
    const bool zin = (lz < g.half_hand_thickness) && (lz > -g.half_hand_thickness);  // :44-45
    const bool back = close && zin && (ly < g.half_bottom_width) && (ly > -g.half_bottom_width) &&
                      (lx < -g.back_margin);                                          // :47-49
    const bool fl = (ly < g.half_bottom_width) && (ly > g.half_bottom_space);        // :54-55
    const bool fr = (ly > -g.half_bottom_width) && (ly < -g.half_bottom_space);      // :56-57
    const bool fing = close && zin && (fl || fr);                                     // :59-60
    nback += __popcll(__ballot(back)) ;
    nfing += __popcll(__ballot(fing));
  }
  // every lane of a wave holds the wave's totals (ballot counts are wave-uniform)
  if ((t & 63) == 0) {
    sback[t >> 6] = nback;
    sfing[t >> 6] = nfing;
  }
  __syncthreads();
  if (t == 0) {
    int* c = counts + ((size_t)b * K + k) * 2;
    c[0] = sback[0] + sback[1] + sback[2] + sback[3];
    c[1] = sfing[0] + sfing[1] + sfing[2] + sfing[3];
  }
}

}  // namespace s4g

extern "C" int s4g_collision_counts_f32(const float* xyz_b3n, const float* g2l_bk44, int64_t B,
                                        int64_t N, int64_t K, const float* gripper6,
                                        int32_t* counts_bk2, s4g_stream_t stream) {
  if (B < 0 || N <= 0 || K < 0 || B > 65535 || N >= (1ll << 31)) return S4G_EINVAL;
  if (B == 0 || K == 0) return S4G_OK;
  if (!xyz_b3n || !g2l_bk44 || !gripper6 || !counts_bk2) return S4G_EINVAL;
  s4g::GripperBox g = {gripper6[0], gripper6[1], gripper6[2], gripper6[3], gripper6[4], gripper6[5]};
  hipLaunchKernelGGL(s4g::collision_counts_kernel, dim3((unsigned)K, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, xyz_b3n, g2l_bk44, (int)N, (int)K, g, counts_bk2, (const int64_t*)nullptr);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

// The same for best-first pose lists of which only the first pose_count_b[b] rows of scene b are poses (device counts:
// the caller never reads them on the host): the padding rows get zero counts and cost one workgroup exit each.
extern "C" int s4g_collision_counts_n_f32(const float* xyz_b3n, const float* g2l_bk44, int64_t B,
                                          int64_t N, int64_t K, const float* gripper6,
                                          const int64_t* pose_count_b, int32_t* counts_bk2, s4g_stream_t stream) {
  if (B < 0 || N <= 0 || K < 0 || B > 65535 || N >= (1ll << 31)) return S4G_EINVAL;
  if (B == 0 || K == 0) return S4G_OK;
  if (!xyz_b3n || !g2l_bk44 || !gripper6 || !counts_bk2 || !pose_count_b) return S4G_EINVAL;
  s4g::GripperBox g = {gripper6[0], gripper6[1], gripper6[2], gripper6[3], gripper6[4], gripper6[5]};
  hipLaunchKernelGGL(s4g::collision_counts_kernel, dim3((unsigned)K, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, xyz_b3n, g2l_bk44, (int)N, (int)K, g, counts_bk2, pose_count_b);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}
