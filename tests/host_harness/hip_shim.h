// hip_shim.h — TEST INFRASTRUCTURE (tests/host_harness): just enough of the HIP device vocabulary for the solver's device
// functions and thread-per-slot kernels to compile as plain host C++ (g++), so that they can run under AddressSanitizer /
// UndefinedBehaviorSanitizer on the CPU (the pool's GPUs run no sanitizer).  Not a product path: nothing in the package
// includes this file, and the kernels that need a real wavefront (k_riccati8 / k_riccati1 / k_eval8 / k_step1 / packing) are
// compiled against inert placeholders and never called by the harness.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
  unsigned x = 1, y = 1, z = 1;
};
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

template <class T, class V>
inline T atomicAdd(T* p, V v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
inline long long clock64() { return 0; }
inline int min(int a, int b) { return a < b ? a : b; }
inline void __syncthreads() {}
inline void __threadfence() {}
// placeholders for the wave-level primitives (only reached from kernels the harness does not run)
inline int __any(int p) { return p; }
template <class T> inline T __shfl(T v, int) { return v; }
template <class T> inline T __shfl_xor(T v, int) { return v; }
#define __builtin_amdgcn_fence(a, b) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

// The 8 lanes that own one instance in k_pick exchange values with xor-shuffles over the lane bits 3..5 (layout.h grp_*):
// here 8 OS threads and a barrier, same exchange pattern, same order of additions.
struct LtLaneGroup {
  pthread_barrier_t bar;
  double slot[8];
};
inline thread_local LtLaneGroup* lt_group = nullptr;
inline thread_local int lt_lane = 0;
inline double lt_xchg(double v, int x) {
  if (!lt_group) return v;
  lt_group->slot[lt_lane] = v;
  pthread_barrier_wait(&lt_group->bar);
  const double r = lt_group->slot[lt_lane ^ x];
  pthread_barrier_wait(&lt_group->bar);
  return r;
}
inline double grp_max(double v) { v = fmax(v, lt_xchg(v, 1)), v = fmax(v, lt_xchg(v, 2)), v = fmax(v, lt_xchg(v, 4)); return v; }
inline double grp_sum(double v) { v += lt_xchg(v, 1), v += lt_xchg(v, 2), v += lt_xchg(v, 4); return v; }
inline double grp_min(double v) { v = fmin(v, lt_xchg(v, 1)), v = fmin(v, lt_xchg(v, 2)), v = fmin(v, lt_xchg(v, 4)); return v; }
