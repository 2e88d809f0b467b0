// LDS random-access microbenchmarks (gfx950): what do one-word table operations cost per CU?
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench3.hip -o /tmp/mb3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int CAP = 6656;
constexpr int U = 8;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t slot_of(uint32_t h) { return (uint32_t)(((uint64_t)(h & 0xffffffu) * CAP) >> 24); }

template <int MODE> __global__ __launch_bounds__(512) void lds_ops(uint64_t *out, int iters, uint32_t distinct) {
  __shared__ uint64_t s_k[CAP];
  __shared__ uint32_t s_v[CAP];
  for (int i = threadIdx.x; i < CAP; i += blockDim.x) { s_k[i] = ~0ull; s_v[i] = 0; }
  __syncthreads();
  uint32_t x = blockIdx.x * 512 + threadIdx.x;
  uint64_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t slot[U]; uint64_t key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x = x * 1664525u + 1013904223u; uint32_t id = mix(x) % distinct; key[u] = (uint64_t)mix(id) << 7 | 1; slot[u] = slot_of(mix(id ^ 0x9e3779b9u)); }
    if (MODE == 0) {          // random 64-bit reads
#pragma unroll
      for (int u = 0; u < U; ++u) acc += __atomic_load_n(&s_k[slot[u]], __ATOMIC_RELAXED);
    } else if (MODE == 1) {   // random 32-bit adds, no return
#pragma unroll
      for (int u = 0; u < U; ++u) atomicAdd(&s_v[slot[u]], 1u);
    } else if (MODE == 2) {   // random 64-bit CAS returning
      unsigned long long old[U];
#pragma unroll
      for (int u = 0; u < U; ++u) old[u] = atomicCAS((unsigned long long *)&s_k[slot[u]], ~0ull, (unsigned long long)key[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += old[u];
    } else if (MODE == 3) {   // CAS + add (first-probe fast path)
      unsigned long long old[U];
#pragma unroll
      for (int u = 0; u < U; ++u) old[u] = atomicCAS((unsigned long long *)&s_k[slot[u]], ~0ull, (unsigned long long)key[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) if (old[u] == ~0ull || old[u] == key[u]) atomicAdd(&s_v[slot[u]], 1u);
    } else if (MODE == 4) {   // 32-bit add returning
      uint32_t r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = atomicAdd(&s_v[slot[u]], 1u);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += r[u];
    } else if (MODE == 5) {   // read + add
      uint64_t c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) c[u] = __atomic_load_n(&s_k[slot[u]], __ATOMIC_RELAXED);
#pragma unroll
      for (int u = 0; u < U; ++u) if (c[u] != key[u]) atomicAdd(&s_v[slot[u]], 1u);
    } else if (MODE == 6) {   // random 32-bit reads
#pragma unroll
      for (int u = 0; u < U; ++u) acc += __atomic_load_n(&s_v[slot[u]], __ATOMIC_RELAXED);
    } else if (MODE == 7) {   // random 32-bit plain stores
#pragma unroll
      for (int u = 0; u < U; ++u) s_v[slot[u]] = (uint32_t)key[u];
    } else if (MODE == 8) {   // no LDS op: generation cost only
#pragma unroll
      for (int u = 0; u < U; ++u) acc += key[u] + slot[u];
    } else if (MODE == 9) {   // 32-bit CAS returning
      uint32_t r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = atomicCAS(&s_v[slot[u]], 0u, (uint32_t)key[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += r[u];
    }
  }
  __syncthreads();
  if (acc == 0x1234567) out[0] = acc;
  if (threadIdx.x == 0) out[1 + blockIdx.x] = s_k[blockIdx.x % CAP] + s_v[3];
}

template <int MODE> int run(const char *name, uint64_t *out, uint32_t distinct) {
  const int grid = 512, iters = 200;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(lds_ops<MODE>, dim3(grid), dim3(512), 0, 0, out, iters, distinct);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(lds_ops<MODE>, dim3(grid), dim3(512), 0, 0, out, iters, distinct);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double ops = (double)grid * 512 * iters * U;
  printf("%-34s distinct=%6u %7.3f ms  %7.2f Gops/s  %5.2f lanes/clk/CU (2.4 GHz, 256 CU)\n", name, distinct, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 2.4e9 / 256);
  return 0;
}

int main() {
  uint64_t *out; CK(hipMalloc(&out, 8 * 4096));
  for (uint32_t d : {3052u, 64u}) {
    run<8>("generation only", out, d);
    run<0>("read b64", out, d);
    run<6>("read b32", out, d);
    run<7>("store b32", out, d);
    run<1>("add u32 noret", out, d);
    run<4>("add u32 ret", out, d);
    run<9>("cas b32 ret", out, d);
    run<2>("cas b64 ret", out, d);
    run<3>("cas b64 + add", out, d);
    run<5>("read b64 + add", out, d);
  }
  return 0;
}
