// Toroidal 32^3 cell grid shared by the ball query and the 3-NN search
// (definitions; the build kernel lives in ball_query.hip).
#pragma once
#include "s4g_common.h"

namespace s4g {

constexpr int GR_DIM = 32;            // cells per axis (toroidal)
constexpr int GR_RANGES = 8;          // z-slabs = build workgroups per scene
constexpr int GR_RANGE_SLOTS = 4096;  // slots per slab
constexpr int GR_START_STRIDE = GR_RANGE_SLOTS + 4;  // +1 end entry, padded to 16 B
constexpr int GR_BUILD_THREADS = 1024;
constexpr int GR_BUILD_U = 8;         // independent point loads in flight per lane
constexpr int GR_COORD_LIMIT = 4096;  // |cell coordinate| bound of the exactness argument
constexpr int GR_MAX_POINTS = 65536;  // bitmap of N bits per wave must fit LDS

struct GridWs {
  float4* sorted;  // [B][GR_RANGES * N] records (x, y, z, index bits)
  int* starts;     // [B][GR_RANGES][GR_START_STRIDE], absolute record offsets
  int* flags;      // [B] 1 = scene out of the exactness range -> SCAN path
  float4* xyz4;    // [B][N] index-ordered (x, y, z, 0) copy: one 16-byte gather per neighbour
};

inline size_t grid_ws_bytes(int64_t B, int64_t N) {
  return (size_t)B * ((size_t)GR_RANGES * N * sizeof(float4) +
                      (size_t)GR_RANGES * GR_START_STRIDE * sizeof(int) + 64 +
                      (size_t)N * sizeof(float4));
}

inline GridWs grid_ws_carve(void* ws, int64_t B, int64_t N) {
  GridWs g;
  char* p = (char*)ws;
  g.sorted = (float4*)p;
  p += (size_t)B * GR_RANGES * N * sizeof(float4);
  g.starts = (int*)p;
  p += (size_t)B * GR_RANGES * GR_START_STRIDE * sizeof(int);
  g.flags = (int*)p;
  p += (size_t)B * 64;
  g.xyz4 = (float4*)p;
  return g;
}

// Queries (ball centres) binned into the same cells, for the cell-centric ball query:
// a compact list of the non-empty x-quads (4 x-adjacent cells) per stripe, each with its
// contiguous run of query records.
struct CellWs {
  float4* sorted;  // [B][GR_RANGES * M] records (x, y, z, query index bits)
  int4* cells;     // [B][GR_RANGES][GR_RANGE_SLOTS] (first slot, first record, count, 0), compacted
  int* ncell;      // [B][GR_RANGES] entries used in `cells`, then [B][GR_RANGES] query records per stripe
};

inline size_t cell_ws_bytes(int64_t B, int64_t M) {
  return (size_t)B * ((size_t)GR_RANGES * M * sizeof(float4) +
                      (size_t)GR_RANGES * GR_RANGE_SLOTS * sizeof(int4) + 64);
}

inline CellWs cell_ws_carve(void* ws, int64_t B, int64_t M) {
  CellWs c;
  char* p = (char*)ws;
  c.sorted = (float4*)p;
  p += (size_t)B * GR_RANGES * M * sizeof(float4);
  c.cells = (int4*)p;
  p += (size_t)B * GR_RANGES * GR_RANGE_SLOTS * sizeof(int4);
  c.ncell = (int*)p;
  return c;
}

__device__ __forceinline__ int grid_coord(float v, float o, float inv_h) {
  return (int)floorf(__fmul_rn(__fsub_rn(v, o), inv_h));
}
// false for NaN / inf / anything outside the exactness range (checked in float:
// the int conversion saturates and abs(INT_MIN) would slip through).
__device__ __forceinline__ bool grid_coord_ok(float v, float o, float inv_h) {
  return fabsf(__fmul_rn(__fsub_rn(v, o), inv_h)) < (float)(GR_COORD_LIMIT - 1);
}
// slot = (range, local): range = y mod 8 (interleaved stripes balance thin,
// table-top shaped clouds over the 8 build workgroups), local = z5 | y-high2 | x5.
// Any function of (y, z) keeps the 32 x-cells of a row contiguous.
__device__ __forceinline__ int grid_range(int yy, int zz) { (void)zz; return yy & 7; }
__device__ __forceinline__ int grid_local_row(int yy, int zz) { return (zz << 7) | ((yy >> 3) << 5); }
__device__ __forceinline__ int grid_slot(int cx, int cy, int cz) {
  const int yy = cy & 31, zz = cz & 31;
  return (grid_range(yy, zz) << 12) | grid_local_row(yy, zz) | (cx & 31);
}


// Builds the grid of `xyz` (B,3,N) with cell edge 1/inv_h into `ws` (one launch);
// write_aos also fills ws.xyz4 (only the fused query+group epilogue reads it).
// inv_h_dev (optional): the cell edge's reciprocal lives in device memory (replaces inv_h).
int launch_grid_build(const float* xyz, int64_t B, int64_t N, float inv_h, GridWs ws,
                      hipStream_t st, bool write_aos = false, const float* inv_h_dev = nullptr);
// Same launch, plus 8 more workgroups per scene that bin the M queries `ctr` (B,3,M)
// into `cw` (cells relative to the same origin as the points').
int launch_grid_build_queries(const float* xyz, const float* ctr, int64_t B, int64_t N,
                              int64_t M, float inv_h, GridWs ws, CellWs cw, hipStream_t st,
                              bool write_aos, const float* inv_h_dev = nullptr);

}  // namespace s4g
