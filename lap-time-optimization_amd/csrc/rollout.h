// rollout.h — closed-loop rollout with free-running instances (ltompc_rollout_dev).
//
// make_step solves one tick for the whole batch and returns when its slowest instance has converged: 4 % of the instances
// need 2 - 10 times the iterations of the rest, and the launches that serve them are 40 % of a tick.  The
// reference's loop (src/mpc.py:140-153: make_step -> plant -> next x0) has no coupling BETWEEN instances, so in a rollout an
// instance that has converged takes its plant step and starts its next tick inside the running batch: the launches stay full,
// a slow instance delays only itself.  Per instance the arithmetic is the synchronous loop's, tick by tick (an instance's
// results do not depend on the batch it is solved in), so the controls are bit-identical to make_step + plant_step in a loop.
//
// Cycle of an instance (SI_PHASE): SOLVING -> (converged: k_roll_finish logs u0 / status, u_prev := u0, appends the instance to
// the pass's plant list) PLANT -> (k_roll_plant over that list, on another stream: RK4 plant step, ticks left -= 1) READY or
// FINAL -> (k_roll_mark: new x0, status bookkeeping as k_load_x0) INIT -> (k_roll_init: warm start of every slot as k_init)
// -> (k_roll_finish of that iteration) SOLVING.
// Each transition is made by ONE thread per instance in a kernel of its own, so that the thread-per-(interval, instance)
// kernels see one state for all their threads; the solver kernels only look at SI_DONE, which stays 1 outside SOLVING.
#pragma once
#include "aux_kernels.h"

namespace ltompc {

enum : int { PH_SOLVING = 0, PH_PLANT = 1, PH_READY = 2, PH_INIT = 3, PH_FINAL = 4 };

// start of a rollout: every instance READY (its first solve starts from the handle's current guess / warm start)
__global__ void k_roll_begin(Work W, int n_ticks) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
  W.si[(size_t)SI_PHASE * W.Bp + b] = PH_READY, W.si[(size_t)SI_TICKS * W.Bp + b] = n_ticks, W.si[(size_t)SI_FINAL * W.Bp + b] = 0;
  W.si[(size_t)SI_DONE * W.Bp + b] = 1;
}
// READY -> INIT: the new measured state (row b of x_rm, written by the plant kernel) becomes x0; status of the solve before
// and the sticky-restoration counter as in k_load_x0
__global__ void k_roll_mark(Work W, const double* x_rm, int sticky, int cold) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
  volatile int* ph = &W.si[(size_t)SI_PHASE * W.Bp + b];
  if (*ph != PH_READY) return;
  __threadfence();  // (x_rm[b] was written before READY)
  d_load_x0(W, x_rm, b, (size_t)b, sticky, cold ? 0 : 1);
  *ph = PH_INIT;
}
// INIT: every slot of the instance as k_init does (cold only in the first pass of a rollout that follows set_initial_guess)
__global__ void k_roll_init(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la, int cold) {
  const Consts& K = *Kp;
  const Work& W = *Wp;
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int j = tid % la.n_pad, k = tid / la.n_pad;
  if (k >= W.N || j >= la.nact[0]) return;
  const int b = la.act[j];
  if (W.si[(size_t)SI_PHASE * W.Bp + b] != PH_INIT) return;
  d_init_slot(K, W, k, b, cold);
}
// end of an iteration: INIT -> SOLVING; SOLVING and converged -> log, u_prev := u0, PLANT.  count: instances not FINAL.
__global__ void k_roll_finish(Work W, Launch la, double* __restrict__ u_log, int* __restrict__ st_log, int* __restrict__ it_log, int n_ticks,
                              int* __restrict__ count, int* __restrict__ plant_list) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= la.nact[0]) return;
  const int b = la.act[j];
  const int N = W.N;
  int* si = W.si;
  const int ph = STI(SI_PHASE);
  if (ph != PH_FINAL) atomicAdd(count, 1);
  if (ph == PH_INIT) {
    STI(SI_PHASE) = PH_SOLVING;
  } else if (ph == PH_SOLVING && STI(SI_DONE)) {
    const double a = PL(W.U, 0, 0, N), c = PL(W.U, 1, 0, N);
    W.uprev[b] = a, W.uprev[(size_t)W.Bp + b] = c;
    const int t = n_ticks - STI(SI_TICKS);
    if (u_log) u_log[((size_t)b * n_ticks + t) * 2] = a, u_log[((size_t)b * n_ticks + t) * 2 + 1] = c;
    if (st_log) st_log[(size_t)b * n_ticks + t] = STI(SI_STATUS);
    if (it_log) it_log[(size_t)b * n_ticks + t] = STI(SI_ITERS);
    STI(SI_PHASE) = PH_PLANT;
    plant_list[atomicAdd(count + 1, 1)] = b;  // (the plant kernel of this pass starts after this kernel has finished)
  }
}
// PLANT -> READY / FINAL: the plant step of the instances that converged in one pass (its list; another stream, concurrent with
// the solver)
__global__ void k_roll_plant(Consts K, Work W, double* x_rm, double dt, int n_sub, const int* __restrict__ n_list, const int* __restrict__ list) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_list[0]) return;
  const int b = list[j];
  volatile int* ph = &W.si[(size_t)SI_PHASE * W.Bp + b];
  double x[8], y[8], uu[2] = {W.uprev[b], W.uprev[(size_t)W.Bp + b]};
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = x_rm[(size_t)b * 8 + i];
  d_plant(K, x, uu, dt, n_sub, y);
#pragma unroll
  for (int i = 0; i < 8; i++) x_rm[(size_t)b * 8 + i] = y[i];
  const int left = W.si[(size_t)SI_TICKS * W.Bp + b] - 1;
  W.si[(size_t)SI_TICKS * W.Bp + b] = left;
  if (left <= 0) W.si[(size_t)SI_FINAL * W.Bp + b] = 1;
  __threadfence();  // (x_rm, ticks before the phase)
  *ph = left > 0 ? PH_READY : PH_FINAL;
}

}  // namespace ltompc
