// Scattered-run write/copy cost vs run length with cheap index math (powers of two, shifts only).
// 512 workgroups x 1024 threads; workgroup w writes runs of `run` keys round-robin to 256 bucket streams.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench6.hip -o tools/exp/mb6
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool READ>
__global__ __launch_bounds__(1024) void scatter_runs(const uint64_t *in, uint64_t *out, uint32_t chunk_log2, uint32_t run_log2, uint32_t misalign) {
  // chunk = keys per workgroup (power of two); nb = 256 buckets; per-bucket region of a workgroup = chunk / 256
  const uint32_t chunk = 1u << chunk_log2;
  const uint64_t b0 = (uint64_t)blockIdx.x << chunk_log2;
  const uint32_t per_wg_bucket = chunk >> 8;
  const uint64_t per_bucket = (uint64_t)per_wg_bucket * gridDim.x;
  for (uint32_t i = threadIdx.x; i < chunk; i += 1024) {
    const uint32_t r = i >> run_log2;            // run index inside this workgroup
    const uint32_t b = r & 255u;                 // bucket of the run
    const uint32_t k = r >> 8;                   // k-th run of bucket b from this workgroup
    const uint64_t dst = (uint64_t)b * per_bucket + (uint64_t)blockIdx.x * per_wg_bucket + ((uint64_t)k << run_log2) + (i & ((1u << run_log2) - 1u)) + misalign;
    out[dst] = READ ? in[b0 + i] : (uint64_t)i;
  }
}

int main() {
  const uint32_t chunk_log2 = 21;                // 2 M keys per workgroup, 512 workgroups -> 2^30 keys = 8.6 GB
  const uint64_t n = 512ull << chunk_log2;
  uint64_t *a, *b;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8 + 4096));
  CK(hipMemset(a, 1, n * 8)); CK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, double bytes, auto fn) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 3; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-52s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  for (uint32_t rl : {4u, 5u, 6u, 9u}) {
    for (uint32_t mis : {0u, 5u, 8u, 4u}) {
      char name[96];
      snprintf(name, sizeof(name), "write-only  256 streams run=%5u keys misalign=%u", 1u << rl, mis);
      timeit(name, 1.0 * n * 8, [&] { hipLaunchKernelGGL(scatter_runs<false>, dim3(512), dim3(1024), 0, 0, a, b, chunk_log2, rl, mis); });
      snprintf(name, sizeof(name), "read+write  256 streams run=%5u keys misalign=%u", 1u << rl, mis);
      timeit(name, 2.0 * n * 8, [&] { hipLaunchKernelGGL(scatter_runs<true>, dim3(512), dim3(1024), 0, 0, a, b, chunk_log2, rl, mis); });
    }
  }
  return 0;
}
