// What does the alignment of a partition pass's write runs cost? Workgroups write 16-byte tuples into 256 private streams each, a
// round of 2048 tuples at a time, as runs of consecutive tuples per stream (what tuple_scatter's staged rounds produce):
//   mode 0: runs of 8 tuples starting on 128-byte lines     mode 1: runs of 8 starting anywhere (every run straddles two lines)
//   mode 2: runs of 4..12 tuples (mean 8), starting wherever the previous one ended (what the kernel does today)
//   mode 11 / 12: runs of 32 tuples (512 bytes) starting anywhere / on lines
//   mode 3: runs of 16 aligned    mode 4: runs of 8 on 64-byte boundaries that are not 128-byte ones    mode 5: runs of 4 / 8 / 12 on 64-byte boundaries
// hipcc -O3 --offload-arch=gfx950 tools/write_runs.hip -o /tmp/write_runs && /tmp/write_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NB = 256, RT = 2048, NT = 512;
__global__ __launch_bounds__(NT) void k(ulonglong2 *out, uint64_t per_stream, uint32_t rounds, int mode) {
  __shared__ uint32_t s_lofs[NB + 1];
  __shared__ uint32_t s_len[NB];
  __shared__ uint64_t s_cur[NB];
  __shared__ uint16_t s_bkt[RT + 1024];
  const uint64_t wg_base = (uint64_t)blockIdx.x * NB * per_stream;
  if (threadIdx.x < NB) {
    uint32_t off = 0;
    if (mode == 1) off = 1u + (threadIdx.x * 2654435761u >> 7) % 7u;
    if (mode == 4 || mode == 5) off = 4u;
    if (mode == 11) off = 1u + (threadIdx.x * 2654435761u >> 7) % 7u;
    s_cur[threadIdx.x] = wg_base + (uint64_t)threadIdx.x * per_stream + off;
  }
  __syncthreads();
  for (uint32_t r = 0; r < rounds; ++r) {
    // run lengths of this round (bucket b = thread b; scan over 256 threads = 4 wavefronts)
    if (threadIdx.x < NB) {
      const uint32_t b = threadIdx.x;
      uint32_t l = 8;
      if (mode == 2) l = 4u + ((b * 40503u + r * 2654435761u) >> 13) % 9u;
      if (mode == 3) l = (b & 1) ? 0 : 16;
      if (mode == 11 || mode == 12) l = 32;   // (with 64 streams of the 256: what a fine scatter of 16-byte records writes per tile)
      if ((mode == 11 || mode == 12) && (b & 3u)) l = 0;
      if (mode == 6) l = 4;
      if (mode == 7) l = (b & 1) ? 4 : 12;
      if (mode == 8) l = 8u * (((b * 40503u + r * 2654435761u) >> 13) % 3u);   // 0, 8 or 16: whole lines
      if (mode == 9) l = 16u * (((b * 40503u + r * 2654435761u) >> 13) % 2u);   // 0 or 16
      if (mode == 5) l = 4u * (1u + ((b * 40503u + r * 2654435761u) >> 13) % 3u);   // 4, 8 or 12: whole 64-byte pieces
      s_len[b] = l;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      uint32_t a0 = s_len[4 * threadIdx.x], a1 = s_len[4 * threadIdx.x + 1], a2 = s_len[4 * threadIdx.x + 2], a3 = s_len[4 * threadIdx.x + 3];
      uint32_t sum = a0 + a1 + a2 + a3, inc = sum;
      for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d); if ((int)threadIdx.x >= d) inc += t; }
      uint32_t ex = inc - sum;
      s_lofs[4 * threadIdx.x] = ex; s_lofs[4 * threadIdx.x + 1] = ex + a0; s_lofs[4 * threadIdx.x + 2] = ex + a0 + a1; s_lofs[4 * threadIdx.x + 3] = ex + a0 + a1 + a2;
      if (threadIdx.x == 63) s_lofs[NB] = inc;
    }
    __syncthreads();
    if (threadIdx.x < NB) for (uint32_t i = s_lofs[threadIdx.x]; i < s_lofs[threadIdx.x + 1]; ++i) s_bkt[i] = (uint16_t)threadIdx.x;
    __syncthreads();
    const uint32_t total = s_lofs[NB];
    for (uint32_t s0 = threadIdx.x; s0 < total; s0 += NT) {
      // mode 10: whole aligned lines in memory, but the lanes are shifted by four against them -- every wave instruction
      // ends in the middle of a line that the next one completes
      const uint32_t s = mode == 10 ? (s0 + 4u < total ? s0 + 4u : s0 + 4u - total) : s0;
      const uint32_t b = s_bkt[s];
      out[s_cur[b] + (s - s_lofs[b])] = make_ulonglong2(s, r);
    }
    __syncthreads();
    if (threadIdx.x < NB) s_cur[threadIdx.x] += s_lofs[threadIdx.x + 1] - s_lofs[threadIdx.x];
    __syncthreads();
  }
}
int main() {
  const uint64_t total = 1200000000ull;   // tuples
  const int G = 2048;
  const uint32_t rounds = (uint32_t)(total / G / RT);
  const uint64_t per_stream = ((uint64_t)rounds * 16 + 64) / 8 * 8;
  ulonglong2 *out;
  CK(hipMalloc(&out, (uint64_t)G * NB * per_stream * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    float best_fill = 1e9f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      CK(hipMemsetAsync(out, it, total * 16, 0));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best_fill) best_fill = ms;
    }
    printf("fill of the same bytes: %.2f ms, %.2f TB/s\n", best_fill, (double)total * 16 / best_fill / 1e9);
    for (int mode = 0; mode < 13; ++mode) {
      float best = 1e9f;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(G), dim3(NT), 0, 0, out, per_stream, rounds, mode);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("mode %d: %.2f ms, %.2f TB/s written\n", mode, best, (double)G * rounds * RT * 16 / best / 1e9);
    }
  }
  return 0;
}
