// mall_probe.hip — what does the 256 MiB Infinity Cache (memory-side, shared by the 8 XCDs) do for a streaming pipeline?
// EXPERIMENT TOOL (not product). For buffer sizes S: bandwidth of a second pass over data that a first pass just READ or just WROTE,
// in the same and in reverse block order. Decides whether ordering the compaction kernel's re-read of the cloud after the min-z
// kernel's read can be served on-die.      hipcc --offload-arch=gfx950 -O3 tools/probes/mall_probe.hip -o variants/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) read_kernel(const float4* __restrict__ p, long n4, int reverse, float* sink) {
  // one workgroup = 2048 consecutive float4 (32 KB), like the min-z kernel's chunk
  long blk = reverse ? (long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const float4* q = p + blk * 2048;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 8; k++) { long i = blk * 2048 + k * 256 + threadIdx.x; if (i < n4) { float4 v = q[k * 256 + threadIdx.x]; acc += v.x + v.y + v.z + v.w; } }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) write_kernel(float4* __restrict__ p, long n4, float v) {
  long blk = blockIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) { long i = blk * 2048 + k * 256 + threadIdx.x; if (i < n4) p[i] = make_float4(v, v, v, v); }
}
__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ s, float4* __restrict__ d, long n4, int reverse) {
  long blk = reverse ? (long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) { long i = blk * 2048 + k * 256 + threadIdx.x; if (i < n4) d[i] = s[i]; }
}

int main() {
  const long maxb = 2048l << 20;
  float4 *a, *b; float* sink;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int sizes[] = {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 768, 1024, 2048};
  printf("%8s %14s %14s %14s %14s %14s %14s   (GB/s of the SECOND kernel)\n", "MiB", "cold read", "read>read", "read>read rev", "write>read", "write>read rev", "copy (r+w)");
  for (int s : sizes) {
    const long bytes = (long)s << 20, n4 = bytes / 16; const int grid = (int)((n4 + 2047) / 2048);
    float ms[6] = {0, 0, 0, 0, 0, 0};
    const int reps = 5;
    for (int r = 0; r < reps; r++) {
      float t;
      // cold: flush the cache with the other buffer first
      hipLaunchKernelGGL(read_kernel, dim3((int)(maxb / 16 / 2048)), dim3(256), 0, 0, b, maxb / 16, 0, sink);
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n4, 0, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t, e0, e1)); ms[0] += t;
      // read > read (same order): `a` was just read
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n4, 0, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t, e0, e1)); ms[1] += t;
      // read > read reversed
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n4, 1, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t, e0, e1)); ms[2] += t;
      // write > read
      hipLaunchKernelGGL(read_kernel, dim3((int)(maxb / 16 / 2048)), dim3(256), 0, 0, b, maxb / 16, 0, sink);
      hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, a, n4, 1.0f);
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n4, 0, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t, e0, e1)); ms[3] += t;
      hipLaunchKernelGGL(read_kernel, dim3((int)(maxb / 16 / 2048)), dim3(256), 0, 0, b, maxb / 16, 0, sink);
      hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, a, n4, 2.0f);
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n4, 1, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t, e0, e1)); ms[4] += t;
      // copy a -> b (bytes counted twice)
      if (2 * bytes <= 2 * maxb) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n4, 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1)); ms[5] += t;
      }
    }
    printf("%8d", s);
    for (int k = 0; k < 6; k++) printf(" %14.0f", (k == 5 ? 2.0 : 1.0) * bytes / (ms[k] / reps * 1e-3) / 1e9);
    printf("\n");
  }
  return 0;
}
