// Cloud pre-processing on device ("next" row f3): workspace crop, voxel
// down-sample, radius-outlier removal.
//
// Replaces CloudPreProcessor.filter_work_space / voxelize / remove_outliers
// (reference grasp_proposal/cloud_processor/cloud_processor.py:12-42, constants
// grasp_proposal/configs/processing_config.py:17-23, caller
// grasp_detector.py:94-105).  The reference delegates the last two to open3d
// (>= 0.12, requirements.txt:2; absent here) and DISCARDS their results
// (cloud_processor.py:34,40 do not assign the returned clouds), so the behaviour
// is defined by the config constants and open3d's published algorithms, restated
// in oracle/preprocess.py:
//   crop      keep p iff lo < p < hi on every axis (strict, cloud_processor.py:16-19),
//             order preserved;
//   voxel     open3d VoxelDownSample: origin = min(points) - voxel/2, cell index =
//             floor((p - origin) / voxel), output = mean of the cell's points.
//             open3d emits cells in unordered_map order (unspecified); here cells
//             come out in ascending (iz, iy, ix) order and the mean is a double
//             sum in point-index order, rounded once to fp32;
//   outliers  open3d RemoveRadiusOutliers: keep p iff the number of points q
//             (p itself included) with |p - q|^2 < r^2 exceeds nb_points.
//             Distances use the library's canonical fp32 arithmetic (s4g_ops.h).
// All three are HBM-light single-scene passes; the neighbour count reuses the
// ball query's cell grid (grid.h) with cell edge = radius.
#include <string.h>

#include "radix_sort.h"

#include "grid.h"
#include "s4g_common.h"

namespace s4g {

constexpr int PP_THREADS = 1024;

// Ordered compaction of the indices whose point lies strictly inside the box:
// one workgroup walks the cloud 1024 points at a time (ballot + wave prefix +
// 16-entry LDS prefix), so the output keeps the input order without a scan pass.
__global__ __launch_bounds__(PP_THREADS) void crop_compact_kernel(
    const float* __restrict__ xyz, int N, float lox, float hix, float loy, float hiy, float loz,
    float hiz, int* __restrict__ index, int* __restrict__ count) {
  __shared__ int wave_cnt[PP_THREADS / 64];
  __shared__ int base_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* px = xyz;
  const float* py = xyz + N;
  const float* pz = xyz + 2 * (size_t)N;
  if (t == 0) base_s = 0;
  __syncthreads();
  for (int j0 = 0; j0 < N; j0 += PP_THREADS) {
    const int j = j0 + t;
    bool keep = false;
    if (j < N) {
      const float x = px[j], y = py[j], z = pz[j];
      keep = x > lox && y > loy && z > loz && x < hix && y < hiy && z < hiz;
    }
    const uint64_t m = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < PP_THREADS / 64; ++w) {
      const int c = wave_cnt[w];
      before += w < wave ? c : 0;
      total += c;
    }
    const int base = base_s;
    if (keep) index[base + before + mask_rank(m)] = j;
    __syncthreads();
    if (t == 0) base_s = base + total;
  }
  __syncthreads();
  if (t == 0) *count = base_s;
}

__global__ void voxel_key_kernel(const float* __restrict__ xyz, int N, float ox, float oy, float oz,
                                 float voxel, int dx, int dy, int dz, uint32_t* __restrict__ key,
                                 int* __restrict__ idx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  // floor((p - origin) / voxel), each operation rounded once (fp32)
  int ix = (int)floorf(__fdiv_rn(__fsub_rn(xyz[j], ox), voxel));
  int iy = (int)floorf(__fdiv_rn(__fsub_rn(xyz[N + j], oy), voxel));
  int iz = (int)floorf(__fdiv_rn(__fsub_rn(xyz[2 * (size_t)N + j], oz), voxel));
  ix = min(max(ix, 0), dx - 1);
  iy = min(max(iy, 0), dy - 1);
  iz = min(max(iz, 0), dz - 1);
  key[j] = ((uint32_t)iz * (uint32_t)dy + (uint32_t)iy) * (uint32_t)dx + (uint32_t)ix;
  idx[j] = j;
}

__global__ void voxel_head_kernel(const uint32_t* __restrict__ key, int N, int* __restrict__ head) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  head[j] = (j == 0 || key[j] != key[j - 1]) ? 1 : 0;
}

// One thread per voxel: walk its (index-ordered) segment, double accumulation.
__global__ void voxel_mean_kernel(const float* __restrict__ xyz, int N,
                                  const uint32_t* __restrict__ key, const int* __restrict__ idx,
                                  const int* __restrict__ head, const int* __restrict__ rank,
                                  float* __restrict__ out, int out_stride, int* __restrict__ count) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  if (j == N - 1) *count = rank[j] + head[j];
  if (!head[j]) return;
  const uint32_t k = key[j];
  double sx = 0.0, sy = 0.0, sz = 0.0;
  int n = 0;
  for (int i = j; i < N && key[i] == k; ++i) {
    const int p = idx[i];
    sx += (double)xyz[p];
    sy += (double)xyz[N + p];
    sz += (double)xyz[2 * (size_t)N + p];
    ++n;
  }
  const int v = rank[j];
  out[v] = (float)(sx / (double)n);
  out[out_stride + v] = (float)(sy / (double)n);
  out[2 * (size_t)out_stride + v] = (float)(sz / (double)n);
}

// Thread per point: count the points of the 27 surrounding cells within the
// radius (the point itself included), stop once the count exceeds nb.
template <bool FMAD>
__global__ void radius_count_kernel(const float* __restrict__ xyz, int N, float r2, float inv_h,
                                    int nb, GridWs ws, uint8_t* __restrict__ keep) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const float* px = xyz;
  const float* py = xyz + N;
  const float* pz = xyz + 2 * (size_t)N;
  const float ox = px[0], oy = py[0], oz = pz[0];
  const float x = px[j], y = py[j], z = pz[j];
  int cnt = 0;
  const bool exact = ws.flags[0] == 0 && grid_coord_ok(x, ox, inv_h) && grid_coord_ok(y, oy, inv_h) &&
                     grid_coord_ok(z, oz, inv_h);
  if (exact) {
    const int icx = grid_coord(x, ox, inv_h), icy = grid_coord(y, oy, inv_h),
              icz = grid_coord(z, oz, inv_h);
    const float4* __restrict__ rec = ws.sorted;
    for (int dz = -1; dz <= 1 && cnt <= nb; ++dz)
      for (int dy = -1; dy <= 1 && cnt <= nb; ++dy) {
        const int zz = (icz + dz) & 31, yy = (icy + dy) & 31;
        const int* __restrict__ st = ws.starts + grid_range(yy, zz) * GR_START_STRIDE +
                                     grid_local_row(yy, zz);
        const int x0 = (icx - 1) & 31;
        int b0 = st[x0], e0, b1 = 0, e1 = 0;
        if (x0 <= GR_DIM - 3) {
          e0 = st[x0 + 3];
        } else {
          e0 = st[GR_DIM];
          b1 = st[0];
          e1 = st[(x0 + 3) & 31];
        }
        for (int i = b0; i < e0 && cnt <= nb; ++i) {
          const float4 q = rec[i];
          cnt += dist2<FMAD>(x, y, z, q.x, q.y, q.z) < r2 ? 1 : 0;
        }
        for (int i = b1; i < e1 && cnt <= nb; ++i) {
          const float4 q = rec[i];
          cnt += dist2<FMAD>(x, y, z, q.x, q.y, q.z) < r2 ? 1 : 0;
        }
      }
  } else {
    // toroidal aliasing is only ruled out inside the grid's exactness range: scan
    for (int i = 0; i < N && cnt <= nb; ++i)
      cnt += dist2<FMAD>(x, y, z, px[i], py[i], pz[i]) < r2 ? 1 : 0;
  }
  keep[j] = cnt > nb ? 1 : 0;
}

template <bool FMAD>
__global__ void radius_count_scan_kernel(const float* __restrict__ xyz, int N, float r2, int nb,
                                         uint8_t* __restrict__ keep) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const float* px = xyz;
  const float* py = xyz + N;
  const float* pz = xyz + 2 * (size_t)N;
  const float x = px[j], y = py[j], z = pz[j];
  int cnt = 0;
  for (int i = 0; i < N && cnt <= nb; ++i)
    cnt += dist2<FMAD>(x, y, z, px[i], py[i], pz[i]) < r2 ? 1 : 0;
  keep[j] = cnt > nb ? 1 : 0;
}

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct VoxelWs {
  uint32_t *key_in, *key_out;
  int *idx_in, *idx_out, *head, *rank;
  void* tmp;
  size_t tmp_bytes, total;
};

static VoxelWs voxel_ws(void* base, int64_t N) {
  VoxelWs w;
  const size_t sort_bytes = radix_sort_ws_bytes((size_t)N), scan_bytes = scan_ws_bytes((size_t)N);
  w.tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(bytes);
    return r;
  };
  w.key_in = (uint32_t*)take(sizeof(uint32_t) * N);
  w.key_out = (uint32_t*)take(sizeof(uint32_t) * N);
  w.idx_in = (int*)take(sizeof(int) * N);
  w.idx_out = (int*)take(sizeof(int) * N);
  w.head = (int*)take(sizeof(int) * N);
  w.rank = (int*)take(sizeof(int) * N);
  w.tmp = take(w.tmp_bytes);
  w.total = off;
  return w;
}

}  // namespace s4g

extern "C" int s4g_crop_indices_f32(const float* xyz_3n, int64_t N, const float* workspace6,
                                    int32_t* index_n, int32_t* count, s4g_stream_t stream) {
  using namespace s4g;
  if (N < 0 || N >= (1ll << 31) || !workspace6 || !count) return S4G_EINVAL;
  if (N > 0 && (!xyz_3n || !index_n)) return S4G_EINVAL;
  hipLaunchKernelGGL(crop_compact_kernel, dim3(1), dim3(PP_THREADS), 0, (hipStream_t)stream, xyz_3n,
                     (int)N, workspace6[0], workspace6[1], workspace6[2], workspace6[3],
                     workspace6[4], workspace6[5], index_n, count);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" size_t s4g_voxel_down_sample_workspace_bytes(int64_t N) {
  if (N <= 0) return 0;
  return s4g::voxel_ws(nullptr, N).total;
}

extern "C" int s4g_voxel_down_sample_f32(const float* xyz_3n, int64_t N, float voxel,
                                         const float* origin3, const int32_t* dims3,
                                         float* out_3n, int32_t* count, void* ws, size_t ws_bytes,
                                         s4g_stream_t stream) {
  using namespace s4g;
  if (N < 0 || N >= (1ll << 31) || !(voxel > 0.f) || !origin3 || !dims3 || !count) return S4G_EINVAL;
  if (dims3[0] <= 0 || dims3[1] <= 0 || dims3[2] <= 0 ||
      (uint64_t)dims3[0] * (uint64_t)dims3[1] * (uint64_t)dims3[2] >= (1ull << 32))
    return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) return (int)hipMemsetAsync(count, 0, sizeof(int32_t), st);
  if (!xyz_3n || !out_3n) return S4G_EINVAL;
  VoxelWs w = voxel_ws(ws, N);
  if (!ws || ws_bytes < w.total) return S4G_EWORKSPACE;
  const int threads = 256, blocks = (int)((N + threads - 1) / threads);
  hipLaunchKernelGGL(voxel_key_kernel, dim3(blocks), dim3(threads), 0, st, xyz_3n, (int)N, origin3[0],
                     origin3[1], origin3[2], voxel, dims3[0], dims3[1], dims3[2], w.key_in, w.idx_in);
  S4G_LAUNCH_CHECK();
  const uint64_t cells = (uint64_t)dims3[0] * (uint64_t)dims3[1] * (uint64_t)dims3[2];
  unsigned bits = 1;
  while (bits < 32 && (1ull << bits) < cells) ++bits;      // the cell keys run 0 .. cells - 1
  int e = radix_sort_pairs(w.tmp, w.tmp_bytes, w.key_in, w.key_out, (uint32_t*)w.idx_in, (uint32_t*)w.idx_out,
                           (size_t)N, bits, st);
  if (e != (int)hipSuccess) return e;
  hipLaunchKernelGGL(voxel_head_kernel, dim3(blocks), dim3(threads), 0, st, w.key_out, (int)N, w.head);
  S4G_LAUNCH_CHECK();
  e = exclusive_scan_i32(w.tmp, w.tmp_bytes, w.head, w.rank, (size_t)N, st);
  if (e != (int)hipSuccess) return e;
  hipLaunchKernelGGL(voxel_mean_kernel, dim3(blocks), dim3(threads), 0, st, xyz_3n, (int)N, w.key_out,
                     w.idx_out, w.head, w.rank, out_3n, (int)N, count);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}

extern "C" size_t s4g_radius_outlier_workspace_bytes(int64_t N) {
  if (N <= 0 || N > s4g::GR_MAX_POINTS) return 0;
  return s4g::grid_ws_bytes(1, N);
}

extern "C" int s4g_radius_outlier_mask_f32(const float* xyz_3n, int64_t N, float radius,
                                           int32_t nb_points, uint8_t* keep_n, void* ws,
                                           size_t ws_bytes, int flags, s4g_stream_t stream) {
  using namespace s4g;
  if (N < 0 || N >= (1ll << 31) || !(radius > 0.f) || nb_points < 0) return S4G_EINVAL;
  if (N == 0) return S4G_OK;
  if (!xyz_3n || !keep_n) return S4G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float r2 = radius * radius;   // fp32 product, as for the ball query
  const bool fmad = (flags & S4G_FLAG_FMAD) != 0;
  const int threads = 256, blocks = (int)((N + threads - 1) / threads);
  if (N > GR_MAX_POINTS || !(radius < 1e18f)) {   // no grid for huge clouds: index-order scan (slow, exact)
    if (fmad) hipLaunchKernelGGL(radius_count_scan_kernel<true>, dim3(blocks), dim3(threads), 0, st, xyz_3n, (int)N, r2, nb_points, keep_n);
    else hipLaunchKernelGGL(radius_count_scan_kernel<false>, dim3(blocks), dim3(threads), 0, st, xyz_3n, (int)N, r2, nb_points, keep_n);
    S4G_LAUNCH_CHECK();
    return S4G_OK;
  }
  if (!ws || ws_bytes < grid_ws_bytes(1, N)) return S4G_EWORKSPACE;
  GridWs g = grid_ws_carve(ws, 1, N);
  // cell edge slightly above the radius, as for the ball query (grid.h)
  const float h = radius * (1.0f + 1.0f / 256.0f);
  const float inv_h = 1.0f / h;
  int rc = launch_grid_build(xyz_3n, 1, N, inv_h, g, st, false);
  if (rc != S4G_OK) return rc;
  if (fmad) hipLaunchKernelGGL(radius_count_kernel<true>, dim3(blocks), dim3(threads), 0, st, xyz_3n, (int)N, r2, inv_h, nb_points, g, keep_n);
  else hipLaunchKernelGGL(radius_count_kernel<false>, dim3(blocks), dim3(threads), 0, st, xyz_3n, (int)N, r2, inv_h, nb_points, g, keep_n);
  S4G_LAUNCH_CHECK();
  return S4G_OK;
}
