// global_load_lds_dwordx4 (gfx950): where does lane L's 16 bytes land in LDS?  Expected: base + imm + L * 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void* gp;
typedef __attribute__((address_space(3))) void* lp;
__global__ void k(const double* __restrict__ src, double* __restrict__ dst) {
  __shared__ double buf[512];
  for (int i = threadIdx.x; i < 512; i += 64) buf[i] = -1.0;
  __syncthreads();
  // two chunks of 1 KB: lane L fetches src[2L], src[2L+1] (chunk 0) and src[128 + 2L], src[128 + 2L + 1] (chunk 1)
  __builtin_amdgcn_global_load_lds((gp)(src + threadIdx.x * 2), (lp)buf, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gp)(src + threadIdx.x * 2), (lp)(buf + 128), 16, 1024, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) dst[i] = buf[i];
}
int main() {
  std::vector<double> h(512), o(512);
  for (int i = 0; i < 512; i++) h[i] = i;
  double *d, *e;
  hipMalloc(&d, 512 * 8), hipMalloc(&e, 512 * 8);
  hipMemcpy(d, h.data(), 512 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
  hipMemcpy(o.data(), e, 512 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; i++) bad += o[i] != (double)i;
  for (int i = 256; i < 512; i++) bad += o[i] != -1.0;
  printf("mismatches %d; o[0..3] %g %g %g %g, o[126..131] %g %g %g %g %g %g, o[254..257] %g %g %g %g\n", bad, o[0], o[1], o[2], o[3], o[126], o[127], o[128],
         o[129], o[130], o[131], o[254], o[255], o[256], o[257]);
  return bad != 0;
}
