// LDS instruction cost vs number of active lanes (gfx950). 512-thread workgroups, 2 per CU, 80 KB LDS each.
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_op_cost.hip -o tools/exp/lds_op_cost
// (round 4, idle MI355X, all 64 lanes active, beside 8.9 CU-clocks of index arithmetic per wave instruction: read b64 +1.0, add u32 +0,
//  write b64 +3.3, read b128 +4.4, compare-and-swap b64 with return +12.7 CU-clocks -- the LDS pipe is not what limits sk_reduce)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int CAP = 6656;
constexpr int U = 8;

// MODE 0 read b64, 1 add u32, 2 cas b64 (returning), 3 read b128, 4 read b32, 5 write b64, 6 none
template <int MODE> __global__ __launch_bounds__(512) void k(uint64_t *out, int iters, uint32_t lane_mod) {
  __shared__ __attribute__((aligned(16))) uint64_t s_k[CAP];
  __shared__ uint32_t s_v[CAP];
  for (int i = threadIdx.x; i < CAP; i += 512) { s_k[i] = ~0ull; s_v[i] = 0; }
  __syncthreads();
  uint32_t x = blockIdx.x * 512 + threadIdx.x + 12345u;
  const bool active = (threadIdx.x % lane_mod) == 0;
  uint64_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t slot[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x = x * 1664525u + 1013904223u; slot[u] = (uint32_t)(((uint64_t)(x >> 8) * CAP) >> 24); }
    if (active) {
      if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += __atomic_load_n(&s_k[slot[u]], __ATOMIC_RELAXED);
      } else if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) atomicAdd(&s_v[slot[u]], 1u);
      } else if (MODE == 2) {
        unsigned long long o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) o[u] = atomicCAS((unsigned long long *)&s_k[slot[u]], ~0ull, (unsigned long long)x + u);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += o[u];
      } else if (MODE == 3) {
#pragma unroll
        for (int u = 0; u < U; ++u) { typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); const u32x4 v = *(const u32x4 *)&s_k[slot[u] & ~1u]; acc += v.x + v.z; }
      } else if (MODE == 4) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += __atomic_load_n(&s_v[slot[u]], __ATOMIC_RELAXED);
      } else if (MODE == 5) {
#pragma unroll
        for (int u = 0; u < U; ++u) __atomic_store_n(&s_k[slot[u]], (uint64_t)x, __ATOMIC_RELAXED);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += slot[u];
      }
    }
  }
  __syncthreads();
  if (acc == 0x1234567) out[0] = acc;
  if (threadIdx.x == 0) out[1 + blockIdx.x] = s_k[blockIdx.x % CAP] + s_v[3];
}

template <int MODE> int run(const char *name, uint64_t *out, uint32_t lane_mod) {
  const int grid = 512, iters = 400;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, out, 10, lane_mod);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, out, iters, lane_mod);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double winstr = (double)grid * 8 * iters * U;          // wave-level instructions
  const double clk_per = (ms * 1e-3 * 2.4e9) / (winstr / 256);  // CU clocks per wave instruction
  printf("%-12s active 1/%-2u lanes: %7.3f ms  %6.1f CU-clk per wave instruction  (%5.2f active lanes/clk/CU)\n", name, lane_mod, ms, clk_per, 64.0 / lane_mod / clk_per);
  return 0;
}

int main() {
  uint64_t *out; CK(hipMalloc(&out, 8 * 4096));
  for (uint32_t m : {1u, 2u, 4u, 8u, 16u, 64u}) {
    run<6>("none", out, m);
    run<4>("read b32", out, m);
    run<0>("read b64", out, m);
    run<3>("read b128", out, m);
    run<5>("write b64", out, m);
    run<1>("add u32", out, m);
    run<2>("cas b64 rtn", out, m);
  }
  return 0;
}
