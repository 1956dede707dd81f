// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the pipeline uses (round-5 review, item 3:
// "calibrate FETCH_SIZE on 12-byte loads with a copy kernel first — the guide says that width is uncalibrated"). EXPERIMENT TOOLING, not product code.
// Every kernel streams a buffer once with a known byte count; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE in a pass of its own)
// and compare the counter (KB) with the bytes printed here:   tools/probes/fetch_calib  ->  profiles/r06_fetch_calibration.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct P12 { float x, y, z; };
__global__ void __launch_bounds__(256) read16_kernel(const float4* __restrict__ in, float* __restrict__ out, long n) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float4 q = in[i]; acc += q.x + q.y + q.z + q.w; }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) read12_kernel(const P12* __restrict__ in, float* __restrict__ out, long n) {   // the label kernel's load of packed points
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const P12 q = in[i]; acc += q.x + q.y + q.z; }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) read2_kernel(const unsigned short* __restrict__ in, float* __restrict__ out, long n) {   // the 2-byte cells
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += in[i];
  if (acc == 0xdeadbeefu) out[0] = (float)acc;
}
__global__ void __launch_bounds__(256) read1_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, long n) {   // the 1-byte channels
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += in[i];
  if (acc == 0xdeadbeefu) out[0] = (float)acc;
}
__global__ void __launch_bounds__(256) write4_kernel(int* __restrict__ out, long n) {   // the pixels
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (int)i;
}
__global__ void __launch_bounds__(256) write12_kernel(P12* __restrict__ out, long n) {   // the compaction kernel's packed elevated points
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { P12 q; q.x = (float)i; q.y = 1.f; q.z = 2.f; out[i] = q; }
}
__global__ void __launch_bounds__(256) write16_kernel(float4* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
int main() {
  const long n = 32l << 20;   // 32 Mi elements: 512 MiB as float4 — beyond the 256 MiB Infinity Cache
  void *a, *b; float* o;
  if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(a, 1, n * 16); hipMemset(b, 1, n * 16);
  const int grid = 256 * 16;
  for (int rep = 0; rep < 4; rep++) {
    hipLaunchKernelGGL(read16_kernel, dim3(grid), dim3(256), 0, 0, (const float4*)a, o, n);
    hipLaunchKernelGGL(read12_kernel, dim3(grid), dim3(256), 0, 0, (const P12*)b, o, n);
    hipLaunchKernelGGL(read2_kernel, dim3(grid), dim3(256), 0, 0, (const unsigned short*)a, o, n * 8);
    hipLaunchKernelGGL(read1_kernel, dim3(grid), dim3(256), 0, 0, (const unsigned char*)b, o, n * 16);
    hipLaunchKernelGGL(write4_kernel, dim3(grid), dim3(256), 0, 0, (int*)a, n * 4);
    hipLaunchKernelGGL(write12_kernel, dim3(grid), dim3(256), 0, 0, (P12*)b, n);
    hipLaunchKernelGGL(write16_kernel, dim3(grid), dim3(256), 0, 0, (float4*)a, n);
  }
  hipDeviceSynchronize();
  printf("bytes per launch (KB): read16 %ld  read12 %ld  read2 %ld  read1 %ld  write4 %ld  write12 %ld  write16 %ld\n", n * 16 / 1024, n * 12 / 1024, n * 16 / 1024, n * 16 / 1024,
         n * 16 / 1024, n * 12 / 1024, n * 16 / 1024);
  return 0;
}
