// Version, error strings and workspace sizing of libs4g_hip.so.
#include "s4g_common.h"

namespace s4g {
size_t fps_workspace_bytes(int64_t B, int64_t N);
size_t ball_query_workspace_bytes(int64_t B, int64_t N, int64_t M, int64_t K);
}  // namespace s4g

extern "C" int s4g_abi_version(void) { return S4G_ABI_VERSION; }

// 1: a measurement build (-DS4G_VARIANTS) that also carries the measured-slower kernel variants
extern "C" int s4g_build_variants(void) {
#ifdef S4G_VARIANTS
  return 1;
#else
  return 0;
#endif
}

// 1: the A/B knobs of include/s4g_ops.h are being honoured in this process (S4G_TEST_KNOBS=1 or a measurement build)
extern "C" int s4g_test_knobs_enabled(void) { return s4g::test_knobs_enabled() ? 1 : 0; }

extern "C" const char* s4g_error_string(int code) {
  if (code == S4G_OK) return "ok";
  if (code == S4G_EINVAL) return "invalid argument (size, null pointer or range)";
  if (code == S4G_EWORKSPACE) return "workspace missing or too small";
  if (code == S4G_EUNSUPPORTED) return "unsupported configuration";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown s4g error";
}

extern "C" size_t s4g_workspace_bytes(int op, int64_t B, int64_t d0, int64_t d1,
                                      int64_t d2) {
  switch (op) {
    case S4G_OP_FPS:
      return s4g::fps_workspace_bytes(B, d0);
    case S4G_OP_BALL_QUERY:
      return s4g::ball_query_workspace_bytes(B, d0, d1, d2);
    default:
      return 0;
  }
}
