// Cross-stream dependency latency: per round  A (s1) -> [B (s2) || C (s1)] -> D (s1, after B)  against  A -> C -> D on one stream.
// Kernels spin for a given number of shader clocks.  usage: xstream [rounds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(long long cycles, int* sink) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (cycles < 0) *sink = 1;
}
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e1, e2;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  int* sink;
  CK(hipMalloc(&sink, 4));
  const long long us = 100;  // clock64 ticks per microsecond (s_memtime: 100 MHz)
  for (int variant = 0; variant < 3; variant++) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < rounds; r++) {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, 20 * us, sink);  // A: 20 us
        if (variant == 0) {  // one stream: A, B, C, D
          hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, 30 * us, sink);
          hipLaunchKernelGGL(spin, dim3(64), dim3(320), 0, s1, 35 * us, sink);
          hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, 5 * us, sink);
        } else if (variant == 1) {  // B beside C on a second stream
          CK(hipEventRecord(e1, s1));
          CK(hipStreamWaitEvent(s2, e1, 0));
          hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s2, 30 * us, sink);
          CK(hipEventRecord(e2, s2));
          hipLaunchKernelGGL(spin, dim3(64), dim3(320), 0, s1, 35 * us, sink);
          CK(hipStreamWaitEvent(s1, e2, 0));
          hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, 5 * us, sink);
        } else {  // lower bound: A, C, D only
          hipLaunchKernelGGL(spin, dim3(64), dim3(320), 0, s1, 35 * us, sink);
          hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, 5 * us, sink);
        }
      }
      CK(hipDeviceSynchronize());
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("%s: %.1f us per round\n", variant == 0 ? "one stream  A(20) B(30) C(35) D(5)" : variant == 1 ? "two streams A(20) [B(30) || C(35)] D(5)" : "without B   A(20) C(35) D(5)", dt / rounds * 1e6);
    }
  }
  return 0;
}
