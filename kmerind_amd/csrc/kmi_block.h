// kmi_block.h -- workgroup-level primitives for 64-wide wavefronts (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmi {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

// Workgroup barrier that orders LDS traffic only: waits for this wave's outstanding LDS operations
// (lgkmcnt) and then s_barrier. Unlike __syncthreads() it does not drain vmcnt, so global loads
// issued before it (software prefetch of the next tile) stay in flight across the barrier. Every
// barrier in these kernels guards LDS data only; HBM results are consumed by later launches.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive scan across the 64 lanes of a wavefront
template <typename T> __device__ __forceinline__ T wave_inclusive_scan(T v) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    T o = __shfl_up(v, d, kWave);
    if ((int)lane_id() >= d) v += o;
  }
  return v;
}

// inclusive scans across the 64 lanes on the DPP path (row_shr 1, 2, 4, 8 inside the rows of 16, row_bcast 15 / 31 across
// them): six VALU instructions, no LDS crossbar
__device__ __forceinline__ uint32_t wave_inclusive_max_dpp(uint32_t v) {   // identity 0
  uint32_t t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;
  return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_sum_dpp(uint32_t v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);
  return v;
}

template <typename T> __device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
  return v;
}

// Exclusive scan of one value per thread over the whole workgroup.
// `scratch` must hold (blockDim.x/64 + 1) elements of T in LDS. Returns the exclusive
// prefix of this thread; *total receives the workgroup sum. Contains two barriers and
// may be called repeatedly with the same scratch (a trailing barrier protects reuse).
template <typename T, bool TRAILING_SYNC = true> __device__ __forceinline__ T block_exclusive_scan(T v, T *scratch, T *total) {
  const uint32_t nw = blockDim.x >> 6;
  T inc = wave_inclusive_scan(v);
  if (lane_id() == kWave - 1) scratch[wave_id()] = inc;
  lds_barrier();
  if (wave_id() == 0) {
    T w = (lane_id() < nw) ? scratch[lane_id()] : T(0);
    T winc = wave_inclusive_scan(w);
    if (lane_id() < nw) scratch[lane_id()] = winc - w;   // exclusive prefix of each wave
    if (lane_id() == nw - 1) scratch[nw] = winc;          // total
  }
  lds_barrier();
  T res = scratch[wave_id()] + inc - v;
  if (total) *total = scratch[nw];
  if (TRAILING_SYNC) lds_barrier();   // without it the caller must not reuse `scratch` before another barrier
  return res;
}

template <typename T> __device__ __forceinline__ T wave_inclusive_max(T v) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    T o = __shfl_up(v, d, kWave);
    if ((int)lane_id() >= d) v = (o > v) ? o : v;
  }
  return v;
}

// Exclusive running maximum over the workgroup (identity 0). scratch: blockDim.x/64 + 1 elements.
template <typename T> __device__ __forceinline__ T block_exclusive_max(T v, T *scratch, T *total) {
  const uint32_t nw = blockDim.x >> 6;
  T inc = wave_inclusive_max(v);
  if (lane_id() == kWave - 1) scratch[wave_id()] = inc;
  lds_barrier();
  if (wave_id() == 0) {
    T w = (lane_id() < nw) ? scratch[lane_id()] : T(0);
    T winc = wave_inclusive_max(w);
    T wexc = __shfl_up(winc, 1, kWave);
    if (lane_id() == 0) wexc = T(0);
    if (lane_id() < nw) scratch[lane_id()] = wexc;
    if (lane_id() == nw - 1) scratch[nw] = winc;
  }
  lds_barrier();
  T prev = __shfl_up(inc, 1, kWave);          // inclusive max of the lanes before this one
  if (lane_id() == 0) prev = T(0);
  T base = scratch[wave_id()];
  T res = (base > prev) ? base : prev;
  if (total) *total = scratch[nw];
  lds_barrier();
  return res;
}

}  // namespace kmi
