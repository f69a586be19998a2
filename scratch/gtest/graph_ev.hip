// Does hipEventRecord inside a captured graph give usable timestamps (hipEventElapsedTime) on this ROCm?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(double* p, int n) { double s = p[threadIdx.x]; for (int i = 0; i < n; i++) s = s * 1.0000001 + 1e-9; p[threadIdx.x] = s; }
struct Big { double a[190]; };
__global__ void spinb(Big b, double* p, int n) { double s = p[threadIdx.x] + b.a[n & 127]; for (int i = 0; i < n; i++) s = s * 1.0000001 + 1e-9; p[threadIdx.x] = s; }
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  double* d; CK(hipMalloc(&d, 64 * 8)); CK(hipMemset(d, 0, 64 * 8));
  hipEvent_t evs[5]; for (auto& x : evs) CK(hipEventCreate(&x));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 4; i++) { CK(hipEventRecord(evs[i], st)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 20000 * (i + 1)); }
  CK(hipEventRecord(evs[4], st));
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    auto t1 = std::chrono::steady_clock::now();
    printf("rep %d wall %.1f us:", rep, std::chrono::duration<double, std::micro>(t1 - t0).count());
    for (int i = 0; i < 4; i++) { float ms = -1; hipError_t er = hipEventElapsedTime(&ms, evs[i], evs[i + 1]); printf(" k%d %.1f us (%s)", i, ms * 1e3, er == hipSuccess ? "ok" : hipGetErrorString(er)); }
    printf("\n");
  }
  // launch-gap comparison: 32 tiny kernels direct vs as a graph
  hipGraph_t g2; hipGraphExec_t ge2;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 32; i++) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 100);
  CK(hipStreamEndCapture(st, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 32; i++) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 100);
    CK(hipStreamSynchronize(st));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge2, st)); CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    printf("32 tiny kernels: direct %.1f us, graph %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count());
  }
  Big big; for (auto& x : big.a) x = 1.0;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 32; i++) hipLaunchKernelGGL(spinb, dim3(1), dim3(64), 0, st, big, d, 100);
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    printf("32 tiny kernels with a 1.5 kB by-value argument: enqueue %.1f us, until done %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t0).count());
  }
  int* hp; CK(hipHostMalloc(&hp, 4));
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    CK(hipMemcpyAsync(hp, d, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    auto t1 = std::chrono::steady_clock::now();
    printf("4-byte read-back + sync on an idle stream: %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  return 0;
}
