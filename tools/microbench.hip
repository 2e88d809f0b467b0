// microbench.hip -- calibration of memory patterns used by the partition kernels (not product code)
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void copy8(const uint64_t *in, uint64_t *out, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy16(const uint4 *in, uint4 *out, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void read8(const uint64_t *in, uint64_t *out, uint64_t n) {
  uint64_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc ^= in[i];
  if (acc == 0x1234567) out[0] = acc;
}
__global__ void write8(uint64_t *out, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = i;
}
// each workgroup owns a contiguous chunk of the input; it writes runs of `run` keys to `nb` bucket streams.
// out layout: bucket b region = [b * per_bucket, ...), workgroup w region inside bucket = w * per_wg_bucket
__global__ void scatter_runs(const uint64_t *in, uint64_t *out, uint64_t n, uint32_t nb, uint32_t run, uint32_t misalign) {
  const uint64_t chunk = n / gridDim.x;
  const uint64_t b0 = (uint64_t)blockIdx.x * chunk;
  const uint64_t per_bucket = n / nb, per_wg_bucket = per_bucket / gridDim.x;
  for (uint64_t i = threadIdx.x; i < chunk; i += blockDim.x) {
    const uint64_t r = i / run;          // run index inside this workgroup
    const uint32_t b = (uint32_t)(r % nb);
    const uint64_t k = r / nb;           // k-th run of bucket b from this workgroup
    uint64_t dst = (uint64_t)b * per_bucket + (uint64_t)blockIdx.x * per_wg_bucket + k * run + (i % run) + misalign;
    if (dst < n) out[dst] = in[b0 + i];
  }
}

int main() {
  const uint64_t n = 1200000000ull;
  uint64_t *a, *b;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8 + 4096));
  CK(hipMemset(a, 1, n * 8)); CK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, double bytes, auto fn) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 3; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-40s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  for (int blocks : {512, 2048}) {
    printf("-- grid %d x 1024\n", blocks);
    timeit("copy8 (r+w bytes)", 2.0 * n * 8, [&] { hipLaunchKernelGGL(copy8, dim3(blocks), dim3(1024), 0, 0, a, b, n); });
    timeit("copy16 (r+w bytes)", 2.0 * n * 8, [&] { hipLaunchKernelGGL(copy16, dim3(blocks), dim3(1024), 0, 0, (const uint4 *)a, (uint4 *)b, n / 2); });
    timeit("read8", 1.0 * n * 8, [&] { hipLaunchKernelGGL(read8, dim3(blocks), dim3(1024), 0, 0, a, b, n); });
    timeit("write8", 1.0 * n * 8, [&] { hipLaunchKernelGGL(write8, dim3(blocks), dim3(1024), 0, 0, b, n); });
  }
  for (uint32_t run : {16u, 32u, 64u, 128u, 512u}) {
    for (uint32_t mis : {0u, 5u}) {
      char name[96]; snprintf(name, sizeof(name), "scatter 256 buckets run=%u keys misalign=%u", run, mis);
      timeit(name, 2.0 * n * 8, [&] { hipLaunchKernelGGL(scatter_runs, dim3(512), dim3(1024), 0, 0, a, b, n, 256u, run, mis); });
    }
  }
  return 0;
}
