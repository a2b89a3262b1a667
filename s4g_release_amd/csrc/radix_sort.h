// Stable LSD radix sort of (uint32 key, uint32 value) pairs and an int32 exclusive scan, hand-written for gfx950
// (round 6: they replace rocprim::radix_sort_pairs / rocprim::exclusive_scan in scatter.hip and preprocess.hip --
// both off the inference hot path: the deterministic backward scatters and the voxel down-sample).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace s4g {

// Bytes of scratch `radix_sort_pairs` needs for n pairs (a multiple of 256).
size_t radix_sort_ws_bytes(size_t n);

// Sorts n pairs by the low `bits` bits of the key, ascending, STABLE (equal keys keep their input order: the
// deterministic scatters rely on it -- the values are positions, and a target's contributions must come out in
// ascending position order).  8 bits per pass: per 2 048-element tile a digit histogram, per digit an exclusive scan
// over the tiles, then every tile places its elements -- a wave walks its 512 consecutive elements 64 at a time,
// ranks equal digits inside the 64 with eight ballots and keeps the running per-digit offsets in LDS.
// The result is in keys_out / vals_out; keys_in / vals_in are clobbered (ping-pong partner).  Returns a hipError_t.
int radix_sort_pairs(void* ws, size_t ws_bytes, uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in,
                     uint32_t* vals_out, size_t n, unsigned bits, hipStream_t st);

// out[i] = in[0] + ... + in[i - 1] for i < n (out may not alias in).  ws: scan_ws_bytes(n) bytes.
size_t scan_ws_bytes(size_t n);
int exclusive_scan_i32(void* ws, size_t ws_bytes, const int* in, int* out, size_t n, hipStream_t st);

}  // namespace s4g
