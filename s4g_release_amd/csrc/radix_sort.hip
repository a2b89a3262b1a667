// Stable LSD radix sort of (uint32, uint32) pairs + int32 exclusive scan for gfx950: see radix_sort.h.
#include "radix_sort.h"

#include "s4g_common.h"

namespace s4g {

namespace {

constexpr int RS_THREADS = 256, RS_WAVES = RS_THREADS / 64;
constexpr int RS_CHUNKS = 8;                          // 64-element chunks a wave walks, in order
constexpr int RS_WAVE_ELEMS = 64 * RS_CHUNKS;         // 512 consecutive elements per wave
constexpr int RS_TILE = RS_WAVE_ELEMS * RS_WAVES;     // 2 048 per workgroup
constexpr int RS_DIGITS = 256;

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(v, off);
    if (lane >= off) v += o;
  }
  return v;
}

// pass 1 of a digit: every tile's histogram, stored digit-major (hist[d][tile]) so that a digit's row is contiguous
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint32_t* __restrict__ keys, size_t n, int shift,
                                                             uint32_t dmask, uint32_t* __restrict__ hist, int nb) {
  __shared__ uint32_t h[RS_DIGITS];
  const int t = threadIdx.x;
  h[t] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_TILE / RS_THREADS; ++i) {
    const size_t j = base + (size_t)i * RS_THREADS + t;
    if (j < n) atomicAdd(&h[(keys[j] >> shift) & dmask], 1u);
  }
  __syncthreads();
  hist[(size_t)t * nb + blockIdx.x] = h[t];
}

// pass 2: one workgroup per digit turns its row into exclusive prefixes over the tiles and leaves the digit's total
__global__ __launch_bounds__(1024) void rs_row_scan_kernel(uint32_t* __restrict__ hist, int nb, uint32_t* __restrict__ total) {
  __shared__ uint32_t wsum[16], woff[16], chunk_total;
  uint32_t* __restrict__ row = hist + (size_t)blockIdx.x * nb;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t carry = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + t;
    const uint32_t v = i < nb ? row[i] : 0u;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
      const uint32_t s = lane < 16 ? wsum[lane] : 0u;
      const uint32_t si = wave_incl_scan_u32(s, lane);
      if (lane < 16) woff[lane] = si - s;
      if (lane == 15) chunk_total = si;
    }
    __syncthreads();
    if (i < nb) row[i] = incl - v + woff[wave] + carry;
    carry += chunk_total;
    __syncthreads();
  }
  if (t == 0) total[blockIdx.x] = carry;
}

// pass 3: stable placement.  Element order inside a tile = (wave, chunk, lane); a wave's running offset per digit
// starts at digit start + this tile's prefix + what the lower waves of the tile hold of that digit.
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, size_t n, int shift,
                                                                uint32_t dmask, const uint32_t* __restrict__ prefix,
                                                                const uint32_t* __restrict__ total, int nb) {
  __shared__ uint32_t whist[RS_WAVES][RS_DIGITS];
  __shared__ volatile uint32_t wbase[RS_WAVES][RS_DIGITS];
  __shared__ uint32_t wsum[RS_WAVES];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // exclusive scan of the 256 digit totals (thread t = digit t)
  const uint32_t tv = total[t];
  const uint32_t ti = wave_incl_scan_u32(tv, lane);
  if (lane == 63) wsum[wave] = ti;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) whist[w][t] = 0;
  __syncthreads();
  uint32_t dstart = ti - tv;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w)
    if (w < wave) dstart += wsum[w];

  const size_t base = (size_t)blockIdx.x * RS_TILE + (size_t)wave * RS_WAVE_ELEMS;
  uint32_t k[RS_CHUNKS], v[RS_CHUNKS];
  bool ok[RS_CHUNKS];
#pragma unroll
  for (int c = 0; c < RS_CHUNKS; ++c) {
    const size_t j = base + (size_t)c * 64 + lane;
    ok[c] = j < n;
    k[c] = ok[c] ? keys_in[j] : 0u;
    v[c] = ok[c] ? vals_in[j] : 0u;
    if (ok[c]) atomicAdd(&whist[wave][(k[c] >> shift) & dmask], 1u);
  }
  __syncthreads();
  {
    uint32_t run = dstart + prefix[(size_t)t * nb + blockIdx.x];
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      wbase[w][t] = run;
      run += whist[w][t];
    }
  }
  __syncthreads();
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int c = 0; c < RS_CHUNKS; ++c) {
    const uint32_t d = (k[c] >> shift) & dmask;
    uint64_t same = __ballot(ok[c]);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1u;
      const uint64_t b = __ballot(one);
      same &= one ? b : ~b;
    }
    const uint32_t rank = (uint32_t)__popcll(same & below);
    const uint32_t off = wbase[wave][d];
    if (ok[c]) {
      keys_out[(size_t)off + rank] = k[c];
      vals_out[(size_t)off + rank] = v[c];
      if (rank == 0) wbase[wave][d] = off + (uint32_t)__popcll(same);   // the run's first lane moves the digit's cursor
    }
  }
}

// ---- int32 exclusive scan: tile sums, one workgroup scans them, every tile scans itself on top of its offset
constexpr int SCN_ITEMS = 8, SCN_TILE = RS_THREADS * SCN_ITEMS;

__global__ __launch_bounds__(RS_THREADS) void scan_tile_sums_kernel(const int* __restrict__ in, size_t n, int* __restrict__ sums) {
  __shared__ int ws[RS_WAVES];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const size_t base = (size_t)blockIdx.x * SCN_TILE + (size_t)t * SCN_ITEMS;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCN_ITEMS; ++i) s += base + i < n ? in[base + i] : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) ws[wave] = s;
  __syncthreads();
  if (t == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(int* __restrict__ sums, int nb) {
  __shared__ uint32_t wsum[16], woff[16], chunk_total;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t carry = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + t;
    const uint32_t v = i < nb ? (uint32_t)sums[i] : 0u;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
      const uint32_t s = lane < 16 ? wsum[lane] : 0u;
      const uint32_t si = wave_incl_scan_u32(s, lane);
      if (lane < 16) woff[lane] = si - s;
      if (lane == 15) chunk_total = si;
    }
    __syncthreads();
    if (i < nb) sums[i] = (int)(incl - v + woff[wave] + carry);
    carry += chunk_total;
    __syncthreads();
  }
}

__global__ __launch_bounds__(RS_THREADS) void scan_tiles_kernel(const int* __restrict__ in, size_t n, const int* __restrict__ sums,
                                                                int* __restrict__ out) {
  __shared__ uint32_t ws[RS_WAVES];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const size_t base = (size_t)blockIdx.x * SCN_TILE + (size_t)t * SCN_ITEMS;
  int x[SCN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCN_ITEMS; ++i) {
    x[i] = base + i < n ? in[base + i] : 0;
    s += (uint32_t)x[i];
  }
  const uint32_t incl = wave_incl_scan_u32(s, lane);
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  uint32_t run = incl - s + (uint32_t)sums[blockIdx.x];
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w)
    if (w < wave) run += ws[w];
#pragma unroll
  for (int i = 0; i < SCN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = (int)run;
    run += (uint32_t)x[i];
  }
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

size_t radix_sort_ws_bytes(size_t n) {
  const size_t nb = (n + RS_TILE - 1) / RS_TILE;
  return up256(sizeof(uint32_t) * RS_DIGITS * (nb ? nb : 1)) + up256(sizeof(uint32_t) * RS_DIGITS);
}

int radix_sort_pairs(void* ws, size_t ws_bytes, uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in,
                     uint32_t* vals_out, size_t n, unsigned bits, hipStream_t st) {
  if (n == 0) return (int)hipSuccess;
  if (!ws || ws_bytes < radix_sort_ws_bytes(n) || !keys_in || !keys_out || !vals_in || !vals_out || bits > 32 ||
      n >= ((size_t)1 << 32))
    return (int)hipErrorInvalidValue;
  const size_t nb = (n + RS_TILE - 1) / RS_TILE;
  if (nb > 0x7fffffffu) return (int)hipErrorInvalidValue;
  uint32_t* hist = (uint32_t*)ws;
  uint32_t* total = (uint32_t*)((char*)ws + up256(sizeof(uint32_t) * RS_DIGITS * nb));
  uint32_t *ks = keys_in, *kd = keys_out, *vs = vals_in, *vd = vals_out;
  const int passes = (int)((bits + 7) / 8);
  for (int p = 0; p < passes; ++p) {
    const int shift = 8 * p;
    const int width = (int)bits - shift < 8 ? (int)bits - shift : 8;      // the last pass may be narrower: bits above `bits` never count
    const uint32_t dmask = (1u << width) - 1u;
    hipLaunchKernelGGL(rs_hist_kernel, dim3((unsigned)nb), dim3(RS_THREADS), 0, st, ks, n, shift, dmask, hist, (int)nb);
    hipLaunchKernelGGL(rs_row_scan_kernel, dim3(RS_DIGITS), dim3(1024), 0, st, hist, (int)nb, total);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3((unsigned)nb), dim3(RS_THREADS), 0, st, ks, vs, kd, vd, n, shift, dmask, hist,
                       total, (int)nb);
    uint32_t* tk = ks; ks = kd; kd = tk;
    uint32_t* tv = vs; vs = vd; vd = tv;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (ks != keys_out) {   // an even number of passes (or none) ends in the input buffers
    e = hipMemcpyAsync(keys_out, ks, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(vals_out, vs, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st);
  }
  return (int)e;
}

size_t scan_ws_bytes(size_t n) {
  const size_t nb = (n + SCN_TILE - 1) / SCN_TILE;
  return up256(sizeof(int) * (nb ? nb : 1));
}

int exclusive_scan_i32(void* ws, size_t ws_bytes, const int* in, int* out, size_t n, hipStream_t st) {
  if (n == 0) return (int)hipSuccess;
  if (!ws || ws_bytes < scan_ws_bytes(n) || !in || !out || n >= ((size_t)1 << 31)) return (int)hipErrorInvalidValue;
  const size_t nb = (n + SCN_TILE - 1) / SCN_TILE;
  int* sums = (int*)ws;
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nb), dim3(RS_THREADS), 0, st, in, n, sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, (int)nb);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)nb), dim3(RS_THREADS), 0, st, in, n, sums, out);
  return (int)hipGetLastError();
}

}  // namespace s4g

// C-ABI access for tests and other hosts (include/s4g_ops.h, ABI 12)
extern "C" size_t s4g_sort_pairs_workspace_bytes(int64_t n) { return n > 0 ? s4g::radix_sort_ws_bytes((size_t)n) : 256; }

extern "C" int s4g_sort_pairs_u32(uint32_t* keys_in, uint32_t* vals_in, int64_t n, int bits, uint32_t* keys_out,
                                  uint32_t* vals_out, void* ws, size_t ws_bytes, s4g_stream_t stream) {
  if (n < 0 || bits < 0 || bits > 32 || n >= ((int64_t)1 << 32)) return S4G_EINVAL;
  if (n == 0) return S4G_OK;
  if (!ws || ws_bytes < s4g::radix_sort_ws_bytes((size_t)n)) return S4G_EWORKSPACE;
  if (!keys_in || !vals_in || !keys_out || !vals_out) return S4G_EINVAL;
  const int e = s4g::radix_sort_pairs(ws, ws_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, (unsigned)bits,
                                      (hipStream_t)stream);
  return e == (int)hipSuccess ? S4G_OK : e;
}

extern "C" size_t s4g_exclusive_scan_workspace_bytes(int64_t n) { return n > 0 ? s4g::scan_ws_bytes((size_t)n) : 256; }

extern "C" int s4g_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes,
                                      s4g_stream_t stream) {
  if (n < 0 || n >= ((int64_t)1 << 31)) return S4G_EINVAL;
  if (n == 0) return S4G_OK;
  if (!ws || ws_bytes < s4g::scan_ws_bytes((size_t)n)) return S4G_EWORKSPACE;
  if (!in || !out || in == out) return S4G_EINVAL;
  const int e = s4g::exclusive_scan_i32(ws, ws_bytes, in, out, (size_t)n, (hipStream_t)stream);
  return e == (int)hipSuccess ? S4G_OK : e;
}
