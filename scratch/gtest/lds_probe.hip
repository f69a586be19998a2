// Latencies that bound a latency-chain kernel on gfx950, one workgroup on an idle chip (s_memtime = shader cycles):
//   (1) dependent ds_read_b64 chain (pointer chase), (2) s_barrier round with 4 wavefronts, (3) LDS write -> barrier -> read by
//   another wavefront -> dependent 8-term fp64 dot product -> write (one "phase" of k_riccati1q), (4) v_rcp_f64 / division chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define WG_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__global__ void __launch_bounds__(256) probe(double* out, long long* ticks, int n) {
  __shared__ int chase[256];
  __shared__ double buf[2][64 * 8];
  const int t = threadIdx.x;
  chase[t] = (t * 37 + 11) & 255;
  for (int j = 0; j < 8; j++) buf[0][(t & 63) * 8 + j] = 1.0 + 1e-3 * j + 1e-5 * t, buf[1][(t & 63) * 8 + j] = 0.5;
  __syncthreads();
  long long t0 = clock64();
  int p = t;
  for (int i = 0; i < n; i++) p = chase[p];
  long long t1 = clock64();
  if (t == 0) ticks[0] = t1 - t0;
  __syncthreads();
  t0 = clock64();
  for (int i = 0; i < n; i++) __builtin_amdgcn_s_barrier();
  t1 = clock64();
  if (t == 0) ticks[1] = t1 - t0;
  // phase: every thread reads 8 values another wavefront wrote, dots them with 8 own values, writes one value; barrier
  double acc = 0.0;
  const int src = ((t + 64) & 255) & 63;  // the slot written by the "next" wavefront's lane
  t0 = clock64();
  for (int i = 0; i < n; i++) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = buf[i & 1][src * 8 + j];
    double s = acc;
#pragma unroll
    for (int j = 0; j < 8; j++) s += v[j] * (1.0 + 1e-9 * j);
    acc = s * 1e-3;
    buf[(i & 1) ^ 1][(t & 63) * 8 + (i & 7)] = acc;
    WG_SYNC_LDS();
  }
  t1 = clock64();
  if (t == 0) ticks[2] = t1 - t0;
  t0 = clock64();
  double d = 1.0 + acc;
  for (int i = 0; i < n; i++) d = 1.0 / (d + 0.5);
  t1 = clock64();
  if (t == 0) ticks[3] = t1 - t0;
  out[t] = acc + p + d;
}
int main() {
  double* d; long long *t, ht[4];
  CK(hipMalloc(&d, 256 * 8)); CK(hipMalloc(&t, 32));
  const int n = 4000;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, t, n); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ht, t, 32, hipMemcpyDeviceToHost));
    printf("cycles per: dependent ds_read %.1f | s_barrier (4 wavefronts) %.1f | phase (8 LDS reads, 8-term dot, LDS write, barrier) %.1f | fp64 division %.1f\n",
           ht[0] / (double)n, ht[1] / (double)n, ht[2] / (double)n, ht[3] / (double)n);
  }
  return 0;
}
