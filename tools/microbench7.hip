// Unaligned runs written (a) lane i -> element i (16-lane groups straddle two 128-byte lines) and
// (b) by destination line: a 16-lane group owns one 128-byte line of the destination and masks the lanes outside the run.
// 512 workgroups x 1024 threads; runs of `run` keys round-robin to 256 bucket streams, every run shifted by `mis` keys.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench7.hip -o tools/exp/mb7
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t run_base(uint32_t r, uint32_t run_log2, uint32_t per_wg_bucket, uint64_t per_bucket) {
  const uint32_t b = r & 255u, k = r >> 8;
  return (uint64_t)b * per_bucket + (uint64_t)blockIdx.x * per_wg_bucket + ((uint64_t)k << run_log2);
}

template <bool BYLINE>
__global__ __launch_bounds__(1024) void scatter_runs(const uint64_t *in, uint64_t *out, uint32_t chunk_log2, uint32_t run_log2, uint32_t mis) {
  const uint32_t chunk = 1u << chunk_log2, run = 1u << run_log2;
  const uint64_t b0 = (uint64_t)blockIdx.x << chunk_log2;
  const uint32_t per_wg_bucket = chunk >> 8;
  const uint64_t per_bucket = (uint64_t)per_wg_bucket * gridDim.x;
  if (!BYLINE) {
    for (uint32_t i = threadIdx.x; i < chunk; i += 1024) {
      const uint32_t r = i >> run_log2;
      out[run_base(r, run_log2, per_wg_bucket, per_bucket) + (i & (run - 1u)) + mis] = in[b0 + i];
    }
  } else {
    const uint32_t lpr = (run >> 4) + (mis ? 1u : 0u);          // destination lines per run
    const uint32_t n_tasks = (chunk >> run_log2) * lpr;
    for (uint32_t t = threadIdx.x >> 4; t < n_tasks; t += 64) {
      const uint32_t r = t / lpr, l = t - r * lpr;               // lpr is 3, 5 ... : a real division, as a table look-up would cost
      const int32_t e = (int32_t)(l * 16u + (threadIdx.x & 15u)) - (int32_t)mis;
      if (e >= 0 && e < (int32_t)run)
        out[run_base(r, run_log2, per_wg_bucket, per_bucket) + l * 16u + (threadIdx.x & 15u)] = in[b0 + ((uint64_t)r << run_log2) + e];
    }
  }
}

int main() {
  const uint32_t chunk_log2 = 21;
  const uint64_t n = 512ull << chunk_log2;
  uint64_t *a, *b;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8 + 4096));
  CK(hipMemset(a, 1, n * 8)); CK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, double bytes, auto fn) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 3; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-64s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  for (uint32_t rl : {5u, 6u}) {
    for (uint32_t mis : {0u, 5u, 11u}) {
      char name[96];
      snprintf(name, sizeof(name), "lane=element   run=%3u keys shift=%2u", 1u << rl, mis);
      timeit(name, 2.0 * n * 8, [&] { hipLaunchKernelGGL(scatter_runs<false>, dim3(512), dim3(1024), 0, 0, a, b, chunk_log2, rl, mis); });
      snprintf(name, sizeof(name), "by dest line   run=%3u keys shift=%2u", 1u << rl, mis);
      timeit(name, 2.0 * n * 8, [&] { hipLaunchKernelGGL(scatter_runs<true>, dim3(512), dim3(1024), 0, 0, a, b, chunk_log2, rl, mis); });
    }
  }
  return 0;
}
