// LDS counting-table insertion variants (gfx950), no HBM traffic: keys come from a counter hash.
// One workgroup = 512 threads, 64 "buckets" of 36608 keys with ~3052 distinct each, table of 6656 slots.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench4.hip -o tools/exp/mb4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int C> struct Cap { static constexpr int v = C; };
#ifndef CAPV
#define CAPV 6656
#endif
constexpr int CAP = CAPV;
constexpr uint64_t EMPTY = ~0ull;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t slot_of(uint32_t h) { return (uint32_t)(((uint64_t)(h & 0xffffffu) * CAP) >> 24); }
__device__ __forceinline__ uint32_t nxt(uint32_t s) { return s + 1 == CAP ? 0u : s + 1; }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void gen(uint32_t seq, uint32_t bucket, uint32_t distinct, uint64_t &key, uint32_t &slot) {
  const uint32_t id = (uint32_t)(((uint64_t)mix(seq * 2654435761u + bucket) * distinct) >> 32) + bucket * 65536u;
  key = ((uint64_t)mix(id) << 20) ^ id;
  slot = slot_of(mix(id ^ 0x9e3779b9u));
}

// V0: batched first compare-and-swap, then per-key serial probe loops
template <int U> __device__ __forceinline__ void insert_v0(uint64_t *s_k, uint32_t *s_v, const uint64_t (&k)[U], uint32_t (&slot)[U]) {
  unsigned long long old[U];
#pragma unroll
  for (int u = 0; u < U; ++u) old[u] = atomicCAS((unsigned long long *)&s_k[slot[u]], EMPTY, (unsigned long long)k[u]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint32_t s = slot[u];
    if (!(old[u] == EMPTY || old[u] == k[u])) {
      for (int p = 1; p < CAP; ++p) {
        s = nxt(s);
        unsigned long long o = atomicCAS((unsigned long long *)&s_k[s], EMPTY, (unsigned long long)k[u]);
        if (o == EMPTY || o == k[u]) break;
      }
    }
    atomicAdd(&s_v[s], 1u);
  }
}

// V1: rounds; all pending keys of a thread read their slot, then CAS where empty, then add where settled
template <int U> __device__ __forceinline__ void insert_v1(uint64_t *s_k, uint32_t *s_v, const uint64_t (&k)[U], uint32_t (&slot)[U]) {
  uint32_t pend = (1u << U) - 1u;
  while (pend) {
    uint64_t cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (pend & (1u << u)) cur[u] = __atomic_load_n(&s_k[slot[u]], __ATOMIC_RELAXED);
    unsigned long long old[U];
    uint32_t want = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) if ((pend & (1u << u)) && cur[u] == EMPTY) want |= 1u << u;
    if (want) {
#pragma unroll
      for (int u = 0; u < U; ++u) old[u] = atomicCAS((unsigned long long *)&s_k[slot[u]], EMPTY, (want & (1u << u)) ? (unsigned long long)k[u] : EMPTY);
#pragma unroll
      for (int u = 0; u < U; ++u) if (want & (1u << u)) cur[u] = (old[u] == EMPTY) ? k[u] : (uint64_t)old[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (pend & (1u << u)) {
        if (cur[u] == k[u]) { atomicAdd(&s_v[slot[u]], 1u); pend &= ~(1u << u); }
        else slot[u] = nxt(slot[u]);
      }
  }
}

// V2: like V0 but the continue loops probe with plain reads (CAS only on empty)
template <int U> __device__ __forceinline__ void insert_v2(uint64_t *s_k, uint32_t *s_v, const uint64_t (&k)[U], uint32_t (&slot)[U]) {
  uint64_t cur[U];
#pragma unroll
  for (int u = 0; u < U; ++u) cur[u] = __atomic_load_n(&s_k[slot[u]], __ATOMIC_RELAXED);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint32_t s = slot[u];
    uint64_t c = cur[u];
    while (c != k[u]) {
      if (c == EMPTY) {
        unsigned long long o = atomicCAS((unsigned long long *)&s_k[s], EMPTY, (unsigned long long)k[u]);
        if (o == EMPTY || o == k[u]) break;
      }
      s = nxt(s);
      c = __atomic_load_n(&s_k[s], __ATOMIC_RELAXED);
    }
    atomicAdd(&s_v[s], 1u);
  }
}

// V3: lower bound, first probe only (wrong results on collisions)
template <int U> __device__ __forceinline__ void insert_v3(uint64_t *s_k, uint32_t *s_v, const uint64_t (&k)[U], uint32_t (&slot)[U]) {
  unsigned long long old[U];
#pragma unroll
  for (int u = 0; u < U; ++u) old[u] = atomicCAS((unsigned long long *)&s_k[slot[u]], EMPTY, (unsigned long long)k[u]);
#pragma unroll
  for (int u = 0; u < U; ++u) if (old[u] == EMPTY || old[u] == k[u]) atomicAdd(&s_v[slot[u]], 1u);
}

// V4: two-slot window per probe step: read slot and slot+1 together (128-bit) -> half the rounds
template <int U> __device__ __forceinline__ void insert_v4(uint64_t *s_k, uint32_t *s_v, const uint64_t (&k)[U], uint32_t (&slot)[U]) {
  uint32_t pend = (1u << U) - 1u;
  while (pend) {
    uint64_t c0[U], c1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (pend & (1u << u)) { c0[u] = __atomic_load_n(&s_k[slot[u]], __ATOMIC_RELAXED); c1[u] = __atomic_load_n(&s_k[nxt(slot[u])], __ATOMIC_RELAXED); }
    uint32_t want = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) if (pend & (1u << u)) {
      if (c0[u] != k[u] && c0[u] != EMPTY) { slot[u] = nxt(slot[u]); c0[u] = c1[u]; }
      if (c0[u] == EMPTY) want |= 1u << u;
    }
    if (want) {
      unsigned long long old[U];
#pragma unroll
      for (int u = 0; u < U; ++u) old[u] = atomicCAS((unsigned long long *)&s_k[slot[u]], EMPTY, (want & (1u << u)) ? (unsigned long long)k[u] : EMPTY);
#pragma unroll
      for (int u = 0; u < U; ++u) if (want & (1u << u)) c0[u] = (old[u] == EMPTY) ? k[u] : (uint64_t)old[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (pend & (1u << u)) {
        if (c0[u] == k[u]) { atomicAdd(&s_v[slot[u]], 1u); pend &= ~(1u << u); }
        else slot[u] = nxt(slot[u]);
      }
  }
}


// V7: groups of 4 slots read with two 128-bit loads; first round batched over the U keys of a thread
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t lo64(u32x4 v) { return (uint64_t)v.x | ((uint64_t)v.y << 32); }
__device__ __forceinline__ uint64_t hi64(u32x4 v) { return (uint64_t)v.z | ((uint64_t)v.w << 32); }
template <int U> __device__ __forceinline__ void insert_v7(uint64_t *s_k, uint32_t *s_v, const uint64_t (&k)[U], uint32_t (&slot)[U]) {
  constexpr uint32_t NG = CAP / 4;
  u32x4 a[U], b[U];
  uint32_t g[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    g[u] = slot[u] >> 2;
    a[u] = *(const volatile u32x4 *)&s_k[g[u] * 4];
    b[u] = *(const volatile u32x4 *)&s_k[g[u] * 4 + 2];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint64_t c0 = lo64(a[u]), c1 = hi64(a[u]), c2 = lo64(b[u]), c3 = hi64(b[u]);
    uint32_t gg = g[u];
    while (true) {
      int pos = (c0 == k[u]) ? 0 : (c1 == k[u]) ? 1 : (c2 == k[u]) ? 2 : (c3 == k[u]) ? 3 : -1;
      if (pos >= 0) { atomicAdd(&s_v[gg * 4 + pos], 1u); break; }
      int e = (c0 == EMPTY) ? 0 : (c1 == EMPTY) ? 1 : (c2 == EMPTY) ? 2 : (c3 == EMPTY) ? 3 : -1;
      if (e >= 0) {
        unsigned long long o = atomicCAS((unsigned long long *)&s_k[gg * 4 + e], EMPTY, (unsigned long long)k[u]);
        if (o == EMPTY || o == k[u]) { atomicAdd(&s_v[gg * 4 + e], 1u); break; }
      } else {
        gg = (gg + 1 == NG) ? 0u : gg + 1;
      }
      const u32x4 x = *(const volatile u32x4 *)&s_k[gg * 4];
      const u32x4 y = *(const volatile u32x4 *)&s_k[gg * 4 + 2];
      c0 = lo64(x); c1 = hi64(x); c2 = lo64(y); c3 = hi64(y);
    }
  }
}

// V6: lane refill. Every lane walks its own U keys; a lane that settles a key moves on to its next key
// at once, so a wavefront iterates max-over-lanes of the SUM of probe lengths instead of the sum of maxima.
template <int U, int S> __device__ __forceinline__ void insert_v6(uint64_t *s_k, uint32_t *s_v, const uint64_t (&k)[U], uint32_t (&slot)[U]) {
  constexpr int PER = U / S;     // S independent streams of PER keys each
  int u[S]; uint64_t key[S]; uint32_t sl[S];
#pragma unroll
  for (int q = 0; q < S; ++q) { u[q] = 0; key[q] = k[q * PER]; sl[q] = slot[q * PER]; }
  bool any = true;
  while (any) {
    uint64_t c[S];
#pragma unroll
    for (int q = 0; q < S; ++q) c[q] = __atomic_load_n(&s_k[sl[q]], __ATOMIC_RELAXED);
    any = false;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      if (u[q] < PER) {
        bool settle = c[q] == key[q];
        if (c[q] == EMPTY) {
          unsigned long long o = atomicCAS((unsigned long long *)&s_k[sl[q]], EMPTY, (unsigned long long)key[q]);
          settle = (o == EMPTY) || (o == key[q]);
        }
        if (settle) {
          atomicAdd(&s_v[sl[q]], 1u);
          const int nu = ++u[q];
          uint64_t nk = k[q * PER + PER - 1]; uint32_t ns = slot[q * PER + PER - 1];
#pragma unroll
          for (int j = PER - 2; j >= 0; --j) { nk = (nu == j) ? k[q * PER + j] : nk; ns = (nu == j) ? slot[q * PER + j] : ns; }
          key[q] = nk; sl[q] = ns;
        } else {
          sl[q] = nxt(sl[q]);
        }
        any |= u[q] < PER;
      }
    }
  }
}

template <int V, int U, int NT> __global__ __launch_bounds__(NT) void tab_kernel(uint64_t *out, int buckets, int per_bucket, uint32_t distinct) {
  __shared__ __attribute__((aligned(16))) uint64_t s_k[CAP];
  __shared__ uint32_t s_v[CAP];
  uint64_t acc = 0;
  for (int b = 0; b < buckets; ++b) {
    for (int i = threadIdx.x; i < CAP; i += NT) { s_k[i] = EMPTY; s_v[i] = 0; }
    lds_barrier();
    const uint32_t bucket = blockIdx.x * buckets + b;
    for (int i0 = 0; i0 < per_bucket; i0 += NT * U) {
      uint64_t k[U]; uint32_t slot[U];
#pragma unroll
      for (int u = 0; u < U; ++u) gen(i0 + u * NT + threadIdx.x, bucket, distinct, k[u], slot[u]);
      if (V == 0) insert_v0<U>(s_k, s_v, k, slot);
      else if (V == 1) insert_v1<U>(s_k, s_v, k, slot);
      else if (V == 2) insert_v2<U>(s_k, s_v, k, slot);
      else if (V == 3) insert_v3<U>(s_k, s_v, k, slot);
      else if (V == 4) insert_v4<U>(s_k, s_v, k, slot);
      else if (V == 10) insert_v7<U>(s_k, s_v, k, slot);
      else if (V == 6) insert_v6<U, 1>(s_k, s_v, k, slot);
      else if (V == 7) insert_v6<U, 2>(s_k, s_v, k, slot);
      else if (V == 8) insert_v6<U, 4>(s_k, s_v, k, slot);
      else { for (int u = 0; u < U; ++u) acc += k[u] + slot[u]; }
    }
    lds_barrier();
    // checksum: total count and number of used slots
    uint32_t cnt = 0, used = 0;
    for (int i = threadIdx.x; i < CAP; i += NT) { cnt += s_v[i]; used += s_k[i] != EMPTY; }
    acc += ((uint64_t)used << 32) + cnt;
    lds_barrier();
  }
  atomicAdd((unsigned long long *)&out[0], (unsigned long long)acc);
}

template <int V, int U, int NT> int run(const char *name, uint64_t *out, uint32_t distinct) {
  const int grid = (CAPV == 6656) ? 512 : 256, buckets = (CAPV == 6656) ? 64 : 128, per_bucket = 36864;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((tab_kernel<V, U, NT>), dim3(grid), dim3(NT), 0, 0, out, 2, per_bucket, distinct);
  CK(hipMemset(out, 0, 8));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((tab_kernel<V, U, NT>), dim3(grid), dim3(NT), 0, 0, out, buckets, per_bucket, distinct);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  uint64_t h; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
  const double ops = (double)grid * buckets * per_bucket;
  printf("%-44s U=%2d NT=%4d distinct=%5u %7.3f ms  %6.2f keys/clk/CU  sum=%llu used=%llu\n", name, U, NT, distinct, ms, ops / (ms * 1e-3) / 2.4e9 / 256,
         (unsigned long long)(h & 0xffffffffu), (unsigned long long)(h >> 32));
  return 0;
}

int main() {
  uint64_t *out; CK(hipMalloc(&out, 64));
#if CAPV == 6656
  for (uint32_t d : {3052u, 1526u}) {
    run<9, 8, 512>("generation only", out, d);
    run<3, 8, 512>("V3 first probe only (lower bound)", out, d);
    run<0, 8, 512>("V0 batched CAS + serial continue", out, d);
    run<2, 8, 512>("V2 batched read + serial read-probe", out, d);
    run<10, 8, 512>("V7 groups of 4", out, d);
    run<10, 4, 512>("V7 groups of 4", out, d);
  }
#else
  for (uint32_t d : {3052u}) {
    run<9, 8, 1024>("generation only", out, d);
    run<3, 8, 1024>("V3 first probe only (lower bound)", out, d);
    run<0, 8, 1024>("V0 batched CAS + serial continue", out, d);
    run<2, 8, 1024>("V2 batched read + serial read-probe", out, d);
    run<2, 4, 1024>("V2", out, d);
    run<10, 8, 1024>("V7 groups of 4", out, d);
    run<10, 4, 1024>("V7 groups of 4", out, d);
  }
#endif
  return 0;
}
