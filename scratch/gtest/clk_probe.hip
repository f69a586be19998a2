// How fast does ONE wavefront run on an otherwise idle chip?  A dependent chain of v_fma_f64 (and of s_add_u32) of known length,
// timed with HIP events, alone and beside a kernel that fills the GPU; s_memtime / s_memrealtime ticks of the same chain.
// build: hipcc -O3 --offload-arch=gfx950 clk_probe.hip -o clk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void chain(double* out, long long* ticks, int n) {
  double a = out[0], b = 1.0000001, c = 1e-9;
  const long long t0 = clock64(), r0 = wall_clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int j = 0; j < 64; j++) a = __builtin_fma(a, b, c);
  }
  const long long t1 = clock64(), r1 = wall_clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) ticks[0] = t1 - t0, ticks[1] = r1 - r0;
}
__global__ void fill(double* out, int n) {
  double a = out[blockIdx.x * blockDim.x + threadIdx.x], b = 1.0000001, c = 1e-9;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int j = 0; j < 64; j++) a = __builtin_fma(a, b, c);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
  double *d, *f; long long *t, ht[2];
  CK(hipMalloc(&d, 64 * 8)); CK(hipMalloc(&f, 1024 * 256 * 8)); CK(hipMalloc(&t, 16));
  CK(hipMemset(d, 0, 64 * 8)); CK(hipMemset(f, 0, 1024 * 256 * 8));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 2000;  // 128 000 dependent FMAs
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, n); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost));
    printf("alone        : %.1f us for %d dependent v_fma_f64 = %.2f ns each; s_memtime ticks %lld (%.3f per ns), s_memrealtime ticks %lld (%.3f per ns)\n",
           ms * 1e3, n * 64, ms * 1e6 / (n * 64.0), ht[0], ht[0] / (ms * 1e6), ht[1], ht[1] / (ms * 1e6));
  }
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(fill, dim3(1024 * 4), dim3(64), 0, s2, f, 40000);  // ~ tens of ms of full-chip fp64 work
    CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, n); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost));
    printf("beside a fill: %.1f us = %.2f ns each; s_memtime %.3f per ns\n", ms * 1e3, ms * 1e6 / (n * 64.0), ht[0] / (ms * 1e6));
    CK(hipDeviceSynchronize());
  }
  // after a pause (clocks may have dropped)
  for (int rep = 0; rep < 3; rep++) {
    hipDeviceSynchronize();
    struct timespec ts = {0, 200000000}; nanosleep(&ts, nullptr);
    CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, n); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("after 200 ms idle: %.1f us = %.2f ns each\n", ms * 1e3, ms * 1e6 / (n * 64.0));
  }
  return 0;
}
