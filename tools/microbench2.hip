// microbench2.hip -- returning global atomics on few addresses (cursor reservation pattern)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void cursor_atomics(unsigned long long *cur, int nbuckets, int iters, unsigned long long *sink) {
  // thread b < nbuckets of every workgroup reserves space in bucket b, `iters` times (one per "tile")
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    if ((int)threadIdx.x < nbuckets) acc += atomicAdd(&cur[threadIdx.x * 16], 32ull);   // 128-B apart
    __syncthreads();
  }
  if (acc == 12345) sink[0] = acc;
}
int main() {
  unsigned long long *cur, *sink;
  hipMalloc(&cur, 256 * 16 * 8); hipMalloc(&sink, 8); hipMemset(cur, 0, 256 * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nb : {256, 128}) for (int wgs : {512, 1024}) {
    int iters = 300;
    hipLaunchKernelGGL(cursor_atomics, dim3(wgs), dim3(512), 0, 0, cur, nb, 10, sink); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(cursor_atomics, dim3(wgs), dim3(512), 0, 0, cur, nb, iters, sink); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double total = (double)wgs * iters * nb;
    printf("buckets=%d wgs=%d tiles/wg=%d: %.3f ms, %.1f M atomics/s total, %.2f us per tile-round\n", nb, wgs, iters, ms, total / ms / 1e3, ms * 1e3 / iters);
  }
  return 0;
}
