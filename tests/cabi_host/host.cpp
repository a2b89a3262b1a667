// Stand-alone C++ host of the C ABI (no Python, no torch): the INTEGRATION.md section 4
// path.  Allocates a synthetic cloud with hipMalloc, runs FPS -> gather -> ball query ->
// group_points through libs4g_hip.so and checks the results against the CPU oracle
// (liboracle is TEST INFRASTRUCTURE; this program lives under tests/).
// Build: see tests/test_cabi_host_gpu.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "s4g_ops.h"

// oracle entry points (oracle/s4g_oracle.c)
extern "C" {
int s4g_oracle_fps(const float* xyz, int64_t B, int64_t N, int64_t M, int64_t* idx, int fmad);
int s4g_oracle_ball_query(const float* xyz, const float* ctr, int64_t B, int64_t N, int64_t M,
                          float radius, int64_t K, int64_t* idx, int64_t* cnt, int fmad);
}

#define CHECK_HIP(x)                                                         \
  do {                                                                       \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
      return 2;                                                              \
    }                                                                        \
  } while (0)
#define CHECK_S4G(x)                                                         \
  do {                                                                       \
    int rc_ = (x);                                                           \
    if (rc_ != S4G_OK) {                                                     \
      fprintf(stderr, "%s: %s (%d)\n", #x, s4g_error_string(rc_), rc_);      \
      return 3;                                                              \
    }                                                                        \
  } while (0)

int main() {
  const int64_t B = 2, N = 9000, M = 700, K = 32;
  const float radius = 0.06f;
  if (s4g_abi_version() != S4G_ABI_VERSION) {
    fprintf(stderr, "ABI version mismatch\n");
    return 1;
  }
  // deterministic cloud: points on a wavy sheet (LCG noise)
  std::vector<float> xyz((size_t)B * 3 * N);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((s >> 40) & 0xFFFFFF) / 16777216.0f;
  };
  for (int64_t b = 0; b < B; ++b)
    for (int64_t j = 0; j < N; ++j) {
      const float u = rnd(), v = rnd();
      xyz[(b * 3 + 0) * N + j] = u * 0.6f;
      xyz[(b * 3 + 1) * N + j] = v * 0.6f;
      xyz[(b * 3 + 2) * N + j] = 0.05f * (u * u - v) + 0.002f * rnd();
    }

  float *d_xyz, *d_ctr, *d_grp;
  int64_t *d_fps, *d_idx, *d_cnt;
  CHECK_HIP(hipMalloc(&d_xyz, xyz.size() * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_ctr, (size_t)B * 3 * M * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_grp, (size_t)B * 3 * M * K * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_fps, (size_t)B * M * sizeof(int64_t)));
  CHECK_HIP(hipMalloc(&d_idx, (size_t)B * M * K * sizeof(int64_t)));
  CHECK_HIP(hipMalloc(&d_cnt, (size_t)B * M * sizeof(int64_t)));
  CHECK_HIP(hipMemcpy(d_xyz, xyz.data(), xyz.size() * sizeof(float), hipMemcpyHostToDevice));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));

  size_t ws_fps = s4g_workspace_bytes(S4G_OP_FPS, B, N, M, 0);
  size_t ws_bq = s4g_workspace_bytes(S4G_OP_BALL_QUERY, B, N, M, K);
  void *w_fps = nullptr, *w_bq = nullptr;
  if (ws_fps) CHECK_HIP(hipMalloc(&w_fps, ws_fps));
  if (ws_bq) CHECK_HIP(hipMalloc(&w_bq, ws_bq));

  CHECK_S4G(s4g_fps_f32(d_xyz, B, N, M, d_fps, w_fps, ws_fps, 0, st));
  CHECK_S4G(s4g_gather_points_f32(d_xyz, d_fps, B, 3, N, M, d_ctr, st));
  CHECK_S4G(s4g_ball_query_f32(d_xyz, d_ctr, B, N, M, radius, K, d_idx, d_cnt, w_bq, ws_bq, 0, st));
  CHECK_S4G(s4g_group_points_f32(d_xyz, d_idx, B, 3, N, M, K, d_grp, st));
  CHECK_HIP(hipStreamSynchronize(st));

  std::vector<int64_t> fps((size_t)B * M), idx((size_t)B * M * K), cnt((size_t)B * M);
  std::vector<float> ctr((size_t)B * 3 * M), grp((size_t)B * 3 * M * K);
  CHECK_HIP(hipMemcpy(fps.data(), d_fps, fps.size() * 8, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(idx.data(), d_idx, idx.size() * 8, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(cnt.data(), d_cnt, cnt.size() * 8, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(ctr.data(), d_ctr, ctr.size() * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(grp.data(), d_grp, grp.size() * 4, hipMemcpyDeviceToHost));

  std::vector<int64_t> rfps(fps.size()), ridx(idx.size()), rcnt(cnt.size());
  s4g_oracle_fps(xyz.data(), B, N, M, rfps.data(), 0);
  if (memcmp(fps.data(), rfps.data(), fps.size() * 8)) {
    fprintf(stderr, "FPS indices differ from the oracle\n");
    return 10;
  }
  for (int64_t b = 0; b < B; ++b)
    for (int c = 0; c < 3; ++c)
      for (int64_t m = 0; m < M; ++m)
        if (ctr[(b * 3 + c) * M + m] != xyz[(b * 3 + c) * N + rfps[b * M + m]]) {
          fprintf(stderr, "gather_points differs\n");
          return 11;
        }
  s4g_oracle_ball_query(xyz.data(), ctr.data(), B, N, M, radius, K, ridx.data(), rcnt.data(), 0);
  if (memcmp(idx.data(), ridx.data(), idx.size() * 8) || memcmp(cnt.data(), rcnt.data(), cnt.size() * 8)) {
    fprintf(stderr, "ball query differs from the oracle\n");
    return 12;
  }
  for (int64_t b = 0; b < B; ++b)
    for (int c = 0; c < 3; ++c)
      for (int64_t mk = 0; mk < M * K; ++mk)
        if (grp[(b * 3 + c) * M * K + mk] != xyz[(b * 3 + c) * N + ridx[b * M * K + mk]]) {
          fprintf(stderr, "group_points differs\n");
          return 13;
        }
  // argument errors are return codes, never exceptions or aborts
  if (s4g_fps_f32(d_xyz, B, 10, 20, d_fps, nullptr, 0, 0, st) != S4G_EINVAL) return 14;
  printf("cabi_host OK: B=%lld N=%lld M=%lld K=%lld\n", (long long)B, (long long)N, (long long)M, (long long)K);
  return 0;
}
