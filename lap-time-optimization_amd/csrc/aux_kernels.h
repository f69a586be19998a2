// aux_kernels.h — initial point, warm-start shift, list compaction, I/O, plant step, test hooks.
#pragma once
#include "linearise.h"

namespace ltompc {

// ------------------------------------------------------------------------------------------ k_init
// Cold: do_mpc set_initial_guess (every state slot = x0, inputs 0, multipliers 0).  Warm: keep the previous
// primal/dual solution un-shifted (do_mpc), node 0 := new x0.  Slacks t = max(-h, bound_push), nu = mu/t.
__device__ __forceinline__ void d_init_slot(const Consts& K, const Work& W, const int k, const int b, const int cold) {
  const int N = W.N;
  double x0[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x0[i] = W.x0[(size_t)i * W.Bp + b];
  if (k == 0) {
#pragma unroll
    for (int i = 0; i < 8; i++) PL(W.X, i, 0, N + 1) = x0[i];
  }
  if (cold) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      PL(W.X, i, k + 1, N + 1) = x0[i], PL(W.C, i, k, N) = x0[i];
      PL(W.L1, i, k, N) = 0.0, PL(W.L2, i, k, N) = 0.0;
    }
    PL(W.U, 0, k, N) = 0.0, PL(W.U, 1, k, N) = 0.0;
  }
  // Option warm_reset_on_fail: the multipliers of a solve that did not converge are not worth starting from (they are
  // what diverged): keep its primal point, restart the equality multipliers at 0 and the barrier at the cold mu_init.
  const int prev = W.si[(size_t)SI_PREV * W.Bp + b];
  // Option resto_sticky: an instance that jammed on the hard constraints or whose horizon problem was infeasible starts
  // its next solves in the restoration phase's elastic mode (the counter is kept by k_load_x0); the multipliers of a
  // converged ELASTIC problem (status INFEASIBLE) are then re-used like any others.
  const bool resto_ok = K.o.resto_rho > 0.0 && !(K.o.soft_rho > 0.0);
  // Option infeasible_sticky: the solve before this one ended INFEASIBLE (the solver's verdict, at the largest penalty): this one
  // starts where that one ended, in the escalated elastic problem, from its primal point and multipliers.
  const bool inf_sticky = !cold && resto_ok && K.o.infeasible_sticky && prev == LTOMPC_STATUS_INFEASIBLE;
  const bool start_elastic = inf_sticky || (!cold && resto_ok && K.o.resto_sticky > 0 && W.si[(size_t)SI_STICKY * W.Bp + b] > 0);
  const bool prev_conv = prev == LTOMPC_STATUS_SOLVED || prev == LTOMPC_STATUS_ACCEPTABLE ||
                         (prev == LTOMPC_STATUS_INFEASIBLE && (K.o.resto_sticky > 0 || K.o.infeasible_sticky));
  const bool after_failure = !cold && K.o.warm_reset_on_fail && !prev_conv;
  if (after_failure) {
#pragma unroll
    for (int i = 0; i < 8; i++) PL(W.L1, i, k, N) = 0.0, PL(W.L2, i, k, N) = 0.0;
  }
  double xp[8], c[8], u[2];
#pragma unroll
  for (int i = 0; i < 8; i++) xp[i] = cold ? x0[i] : PL(W.X, i, k + 1, N + 1), c[i] = cold ? x0[i] : PL(W.C, i, k, N);
  u[0] = cold ? 0.0 : PL(W.U, 0, k, N), u[1] = cold ? 0.0 : PL(W.U, 1, k, N);
  if (!cold && K.o.resto_shift_retry && !K.o.warm_shift) {
    // the solve's own starting point, for options.resto_shift_retry (d_pick / d_update); indexed by the caller's instance index
    const size_t ob = W.orig[b];
#pragma unroll
    for (int i = 0; i < 8; i++) W.BK[((size_t)i * N + k) * W.Bp + ob] = xp[i], W.BK[((size_t)(8 + i) * N + k) * W.Bp + ob] = c[i];
    W.BK[((size_t)16 * N + k) * W.Bp + ob] = u[0], W.BK[((size_t)17 * N + k) * W.Bp + ob] = u[1];
  }
  const double mu_s = (!cold && !after_failure && K.o.mu_init_warm > 0) ? K.o.mu_init_warm : K.o.mu_init;
  const double eps = (K.o.smooth_scale > 0 || K.o.smooth_eps_min > 0) ? fmax(K.o.smooth_eps_min, K.o.smooth_scale * mu_s) : 0.0;
  const double rho = inf_sticky ? fmax(K.o.resto_rho, K.o.resto_rho_max) : (start_elastic ? K.o.resto_rho : K.o.soft_rho);  // (the restoration phase replaces it per instance, see d_pick)
  const double mu = mu_s * pen_scale(rho);  // (penalty scale, layout.h: 1 unless rho > RHO_UNIT)
  init_slot_slacks<BoundsAny>(K, W, k, b, mu, eps, rho, xp, c, u);
  if (k == 0) {
    double* st = W.st;
    st[(size_t)ST_MU * W.Bp + b] = mu, st[(size_t)ST_EPS * W.Bp + b] = eps, st[(size_t)ST_EPS_NEXT * W.Bp + b] = eps;
    st[(size_t)ST_DW_LAST * W.Bp + b] = 0.0, st[(size_t)ST_FORCE_REG * W.Bp + b] = 0.0;
    st[(size_t)ST_ALPHA * W.Bp + b] = 0.0, st[(size_t)ST_ADUA * W.Bp + b] = 0.0;
    st[(size_t)ST_E0 * W.Bp + b] = 1e300, st[(size_t)ST_OBJ * W.Bp + b] = 0.0, st[(size_t)ST_TAU * W.Bp + b] = 0.99;
    st[(size_t)ST_THETA0 * W.Bp + b] = -1.0, st[(size_t)ST_THMAX * W.Bp + b] = 0.0, st[(size_t)ST_THMIN * W.Bp + b] = 0.0;
    st[(size_t)ST_DW * W.Bp + b] = 0.0, st[(size_t)ST_DW_TRY * W.Bp + b] = 0.0;
    st[(size_t)ST_C00 * W.Bp + b] = cost_eval(K.p, K.T, eps, x0, false, nullptr, nullptr);
    st[(size_t)ST_RHO * W.Bp + b] = rho, st[(size_t)ST_VIOL * W.Bp + b] = 0.0;
    for (int i = 0; i < SI_NF; i++)
      if (i != SI_PREV && (i != SI_STICKY || cold) && i != SI_PHASE && i != SI_TICKS && i != SI_FINAL) W.si[(size_t)i * W.Bp + b] = 0;
    W.si[(size_t)SI_STATUS * W.Bp + b] = LTOMPC_STATUS_MAX_ITER;
    W.si[(size_t)SI_WARM * W.Bp + b] = cold ? 0 : 1;
    W.si[(size_t)SI_FBARMED * W.Bp + b] = (!cold && !after_failure && K.o.mu_init_warm > 0 && K.o.warm_fallback_iter > 0) ? 1 : 0;
    {  // options.node0_check: the track constraints at the measured state (node 0 of do_mpc's NLP), at the final smoothing
      const double eps0 = (K.o.smooth_scale > 0 || K.o.smooth_eps_min > 0) ? K.o.smooth_eps_min : 0.0;
      double g0[3];
      cons_eval(K.p, K.T, eps0, x0, g0, nullptr, nullptr, nullptr, nullptr, nullptr);
      st[(size_t)ST_G0 * W.Bp + b] = fmax(g0[0], fmax(g0[1], g0[2]));
    }
    if (start_elastic) W.si[(size_t)SI_RESTO * W.Bp + b] = 1, W.si[(size_t)SI_NRESTO * W.Bp + b] = 1, W.si[(size_t)SI_STARTEL * W.Bp + b] = 1;
  }
}
__global__ void k_init(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, int cold) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int b = tid % W.Bp, k = tid / W.Bp;
  if (k >= W.N || b >= W.B) return;
  d_init_slot(K, W, k, b, cold);
}

// ------------------------------------------------------------------------------------------ k_shift
// Option warm_shift: previous solution moved one interval ahead (x_k <- x_{k+1}, c/u/multipliers likewise, the last
// interval repeated).  Two passes through the step buffers so that no thread reads what another one overwrites.
__global__ void k_shift(Work W, int pass) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int b = tid % W.Bp, k = tid / W.Bp;
  const int N = W.N;
  if (k > N || b >= W.B) return;
  if (pass == 0) {
    const int kx = k + 1 <= N ? k + 1 : N, ks = k + 1 <= N - 1 ? k + 1 : N - 1;
#pragma unroll
    for (int i = 0; i < 8; i++) PL(W.dX, i, k, N + 1) = PL(W.X, i, kx, N + 1);
    if (k < N) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        PL(W.dC, i, k, N) = PL(W.C, i, ks, N);
        PL(W.nL1, i, k, N) = PL(W.L1, i, ks, N), PL(W.nL2, i, k, N) = PL(W.L2, i, ks, N);
      }
      PL(W.dU, 0, k, N) = PL(W.U, 0, ks, N), PL(W.dU, 1, k, N) = PL(W.U, 1, ks, N);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) PL(W.X, i, k, N + 1) = PL(W.dX, i, k, N + 1);
    if (k < N) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        PL(W.C, i, k, N) = PL(W.dC, i, k, N);
        PL(W.L1, i, k, N) = PL(W.nL1, i, k, N), PL(W.L2, i, k, N) = PL(W.nL2, i, k, N);
      }
      PL(W.U, 0, k, N) = PL(W.dU, 0, k, N), PL(W.U, 1, k, N) = PL(W.dU, 1, k, N);
    }
  }
}


// ------------------------------------------------------------------------------------------ compaction
__global__ void k_act_identity(int* act, int* nact, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) act[b] = b;
  if (b == 0) nact[0] = B;
}
// Stable compaction of the unfinished instances of `src[0..nsrc)` into `dst`; one workgroup of 1024 threads.
__global__ void __launch_bounds__(1024) k_compact(const int* __restrict__ src, const int* __restrict__ nsrc_p,
                                                   const int* __restrict__ done, int* __restrict__ dst, int* __restrict__ ndst) {
  __shared__ int cnt[1024];
  const int t = threadIdx.x, nsrc = nsrc_p[0];
  const int chunk = (nsrc + 1023) / 1024, lo = t * chunk, hi = min(nsrc, lo + chunk);
  int c = 0;
  for (int j = lo; j < hi; j++) c += done[src[j]] ? 0 : 1;
  cnt[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan
    int v = t >= off ? cnt[t - off] : 0;
    __syncthreads();
    cnt[t] += v;
    __syncthreads();
  }
  int pos = cnt[t] - c;
  for (int j = lo; j < hi; j++) {
    int b = src[j];
    if (!done[b]) dst[pos++] = b;
  }
  if (t == 1023) ndst[0] = cnt[1023];
}

// ------------------------------------------------------------------------------------------ packing
// Index compaction alone leaves the unfinished instances where they are: at 55 % density a wavefront's 64 lanes span
// ~116 instances and every load moves twice the sectors it uses (measured: launches of 4500 instances as slow as
// launches of 8192).  Packing moves the instances themselves: the first `nslots` physical slots are permuted so that
// the unfinished ones come first (stable), the list becomes the identity, `orig` remembers where every slot came from.
// The step buffers (free between two iterations) and the Riccati buffer are the temporaries.
__global__ void __launch_bounds__(1024) k_pack_perm(const int* __restrict__ nslots_p, const int* __restrict__ done, int* __restrict__ perm,
                                                     int* __restrict__ act, int* __restrict__ nact) {
  __shared__ int cnt[1024];
  const int t = threadIdx.x, n = nslots_p[0];
  const int chunk = (n + 1023) / 1024, lo = min(n, t * chunk), hi = min(n, lo + chunk);
  int c = 0;
  for (int j = lo; j < hi; j++) c += done[j] ? 0 : 1;
  cnt[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan
    int v = t >= off ? cnt[t - off] : 0;
    __syncthreads();
    cnt[t] += v;
    __syncthreads();
  }
  const int n_act = cnt[1023];
  int pa = cnt[t] - c, pd = n_act + lo - pa;  // unfinished before this chunk; finished before it = lo - pa
  for (int j = lo; j < hi; j++) {
    if (!done[j]) act[pa] = pa, perm[pa++] = j;
    else perm[pd++] = j;
  }
  if (t == 0) nact[0] = n_act;
}
// inverse of `orig` (for the final un-packing: slot j goes back to orig[j])
__global__ void k_pack_inverse(const int* __restrict__ orig, int* __restrict__ inv, int B) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < B) inv[orig[j]] = j;
}
// pass 0: temporaries[j] = state[perm[j]], pass 1: state[j] = temporaries[j], for the slots j < nslots
__global__ void k_pack(Work W, const int* __restrict__ perm, const int* __restrict__ nslots_p, int nslots_max, int* __restrict__ orig,
                       int ni, int nel, int pass) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = tid % nslots_max, k = tid / nslots_max, N = W.N;
  const int nslots = nslots_p ? nslots_p[0] : nslots_max;
  if (k > N || j >= nslots) return;
  const size_t Bp = W.Bp;
  const size_t s = pass == 0 ? perm[j] : j;  // source slot
  auto mv = [&](gptr<double> dst, gptr<double> src, int nf, int NK) {
    for (int f = 0; f < nf; f++) dst[((size_t)f * NK + k) * Bp + j] = src[((size_t)f * NK + k) * Bp + s];
  };
  if (pass == 0) {
    mv(W.dX, W.X, 8, N + 1);
    if (k < N) mv(W.dC, W.C, 8, N), mv(W.dU, W.U, 2, N), mv(W.nL1, W.L1, 8, N), mv(W.nL2, W.L2, 8, N), mv(W.dT, W.T, ni + 3 + nel, N), mv(W.dNU, W.NU, ni, N);
  } else {
    mv(W.X, W.dX, 8, N + 1);
    if (k < N) mv(W.C, W.dC, 8, N), mv(W.U, W.dU, 2, N), mv(W.L1, W.nL1, 8, N), mv(W.L2, W.nL2, 8, N), mv(W.T, W.dT, ni + 3 + nel, N), mv(W.NU, W.dNU, ni, N);
  }
  if (k == 0) {  // per-instance arrays through the Riccati buffer: x0 (8), uprev (2), st, filt | si, orig (ints)
    gptr<double> tmp = W.RC;
    gptr<int> itmp = (gptr<int>)(W.RC + (size_t)(10 + ST_NF + 2 * FILTER_MAX) * Bp);
    for (int f = 0; f < 10 + ST_NF + 2 * FILTER_MAX; f++) {
      gptr<double> a = f < 8 ? W.x0 + (size_t)f * Bp : f < 10 ? W.uprev + (size_t)(f - 8) * Bp
                       : f < 10 + ST_NF ? W.st + (size_t)(f - 10) * Bp : W.filt + (size_t)(f - 10 - ST_NF) * Bp;
      if (pass == 0) tmp[(size_t)f * Bp + j] = a[s];
      else a[j] = tmp[(size_t)f * Bp + j];
    }
    for (int f = 0; f <= SI_NF; f++) {
      gptr<int> a = f < SI_NF ? W.si + (size_t)f * Bp : (gptr<int>)orig;
      if (pass == 0) itmp[(size_t)f * Bp + j] = a[s];
      else a[j] = itmp[(size_t)f * Bp + j];
    }
  }
}

// ------------------------------------------------------------------------------------------ I/O helpers
// row-major (B x 8) user buffer -> [8][Bp] planes
// (orig != nullptr: the instances are in packed order, slot b holds the caller's instance orig[b])
// (sticky: options.resto_sticky, 0 = off; update = 0 when called for set_initial_guess, whose k_init resets the counter)
__device__ __forceinline__ void d_load_x0(const Work& W, const double* __restrict__ x0_rm, const int b, const size_t r, const int sticky, const int update) {
#pragma unroll
  for (int i = 0; i < 8; i++) W.x0[(size_t)i * W.Bp + b] = x0_rm[r * 8 + i];
  // (the solver's own status: the node-0 rule of options.node0_check may have turned a converged solve into INFEASIBLE)
  const int node0 = W.si[(size_t)SI_NODE0 * W.Bp + b];
  const int prev = node0 ? node0 - 1 : W.si[(size_t)SI_STATUS * W.Bp + b];
  W.si[(size_t)SI_PREV * W.Bp + b] = prev;  // k_init resets the rest
  if (update && sticky > 0) {
    // jammed on the hard constraints (restoration entered from them) or infeasible: the next `sticky` solves start elastic
    const bool jammed = W.si[(size_t)SI_NRESTO * W.Bp + b] > 0 && !W.si[(size_t)SI_STARTEL * W.Bp + b];
    const int c = W.si[(size_t)SI_STICKY * W.Bp + b];
    W.si[(size_t)SI_STICKY * W.Bp + b] = (prev == LTOMPC_STATUS_INFEASIBLE || jammed) ? sticky : (c > 0 ? c - 1 : 0);
  }
}
__global__ void k_load_x0(Work W, const double* __restrict__ x0_rm, const int* __restrict__ orig, int sticky, int update) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
  d_load_x0(W, x0_rm, b, orig ? orig[b] : b, sticky, update);
}
__global__ void k_zero_uprev(Work W) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
  W.uprev[b] = 0.0, W.uprev[(size_t)W.Bp + b] = 0.0;
}
// u0 = U[:,0,:] -> row-major (B x 2) and u_prev := u0 (do_mpc: _u_prev = last returned u0)
__global__ void k_store_u0(Work W, double* __restrict__ u0_rm, const int* __restrict__ orig) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
  const int N = W.N;
  const size_t r = orig ? orig[b] : b;
  double a = PL(W.U, 0, 0, N), c = PL(W.U, 1, 0, N);
  if (u0_rm) u0_rm[r * 2] = a, u0_rm[r * 2 + 1] = c;
  W.uprev[b] = a, W.uprev[(size_t)W.Bp + b] = c;
}

// histogram of the statuses of the last solve (order of the instances does not matter: no un-packing needed) and the
// total number of interior-point iterations
__global__ void k_status_counts(Work W, int* __restrict__ counts, unsigned long long* __restrict__ iters_sum) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
  const int s = W.si[(size_t)SI_STATUS * W.Bp + b];
  atomicAdd(&counts[s < 0 ? 7 : (s > 7 ? 7 : s)], 1);
  // counts[16 .. 23]: the solver's own statuses (before the node-0 rule of options.node0_check)
  const int n0 = W.si[(size_t)SI_NODE0 * W.Bp + b], ss = n0 ? n0 - 1 : s;
  atomicAdd(&counts[16 + (ss < 0 ? 7 : (ss > 7 ? 7 : ss))], 1);
  atomicAdd(iters_sum, (unsigned long long)W.si[(size_t)SI_ITERS * W.Bp + b]);
}

// plant: classical RK4 with n_sub sub-steps, zero-order-hold input (do_mpc Simulator / CVODES stand-in)
__device__ __forceinline__ void d_plant(const Consts& K, const double* x, const double* uu, const double dt, const int n_sub, double* y) {
#pragma unroll
  for (int i = 0; i < 8; i++) y[i] = x[i];
  const double hs = dt / n_sub;
  for (int s = 0; s < n_sub; s++) {
    double k1[8], k2[8], k3[8], k4[8], z[8];
    rhs_val(K.p, K.T, 0.0, y, uu, k1);
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = y[i] + 0.5 * hs * k1[i];
    rhs_val(K.p, K.T, 0.0, z, uu, k2);
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = y[i] + 0.5 * hs * k2[i];
    rhs_val(K.p, K.T, 0.0, z, uu, k3);
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = y[i] + hs * k3[i];
    rhs_val(K.p, K.T, 0.0, z, uu, k4);
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] += hs / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
  }
}
__global__ void k_plant(Consts K, int B, const double* __restrict__ x, const double* __restrict__ u, double dt,
                        int n_sub, double* __restrict__ xn) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double xb[8], y[8], uu[2] = {u[(size_t)b * 2], u[(size_t)b * 2 + 1]};
#pragma unroll
  for (int i = 0; i < 8; i++) xb[i] = x[(size_t)b * 8 + i];
  d_plant(K, xb, uu, dt, n_sub, y);
#pragma unroll
  for (int i = 0; i < 8; i++) xn[(size_t)b * 8 + i] = y[i];
}

__global__ void k_slip_forces(Consts K, int B, const double* __restrict__ x, double* __restrict__ alpha,
                              double* __restrict__ Fy) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const ltompc_params& p = K.p;
  const double* xb = x + (size_t)b * 8;
  double af = atan2(xb[4] + p.length_f * xb[5], xb[3]) - xb[6];
  double ar = atan2(xb[4] - p.length_r * xb[5], xb[3]);
  double L = p.length_f + p.length_r;
  double Fnf = p.length_r * p.mass * p.gravity / L, Fnr = p.length_f * p.mass * p.gravity / L;
  alpha[(size_t)b * 2] = af, alpha[(size_t)b * 2 + 1] = ar;
  Fy[(size_t)b * 2] = -Fnf * p.D_f * sin(p.C_f * atan(p.B_f * af));
  Fy[(size_t)b * 2 + 1] = -Fnr * p.D_r * sin(p.C_r * atan(p.B_r * ar));
}

// test hooks: model derivatives at given points (thread = point)
__global__ void k_test_model(Consts K, int n, double eps, const double* __restrict__ x, const double* __restrict__ lam,
                             double* __restrict__ f, double* __restrict__ J, double* __restrict__ H,
                             double* __restrict__ cval, double* __restrict__ cgrad, double* __restrict__ cH,
                             double* __restrict__ gval, double* __restrict__ ggrad, double* __restrict__ gH) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double xx[8], ll[8], ff[8], JJ[48], HH[36];
#pragma unroll
  for (int i = 0; i < 8; i++) xx[i] = x[(size_t)t * 8 + i], ll[i] = lam[(size_t)t * 8 + i];
#pragma unroll
  for (int i = 0; i < 36; i++) HH[i] = 0.0;
  rhs_derivs(K.p, K.T, eps, xx, ff, JJ, ll, 1.0, HH);
  for (int i = 0; i < 6; i++) f[(size_t)t * 8 + i] = ff[i];
  f[(size_t)t * 8 + 6] = f[(size_t)t * 8 + 7] = 0.0;
  for (int i = 0; i < 48; i++) J[(size_t)t * 64 + i] = JJ[i];
  for (int i = 48; i < 64; i++) J[(size_t)t * 64 + i] = 0.0;
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) H[(size_t)t * 64 + i * 8 + j] = HH[sidx(i, j)];
  for (int term = 0; term < 2; term++) {
    double g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, Hc[36];
    for (int i = 0; i < 36; i++) Hc[i] = 0.0;
    cval[(size_t)t * 2 + term] = cost_eval(K.p, K.T, eps, xx, term == 1, g, Hc);
    for (int i = 0; i < 8; i++) {
      cgrad[((size_t)t * 2 + term) * 8 + i] = g[i];
      for (int j = 0; j < 8; j++) cH[((size_t)t * 2 + term) * 64 + i * 8 + j] = Hc[sidx(i, j)];
    }
  }
  double gv[3], gs[3], gn[3], gm[3], hss[3], hmm[3];
  cons_eval(K.p, K.T, eps, xx, gv, gs, gn, gm, hss, hmm);
  for (int q = 0; q < 3; q++) {
    gval[(size_t)t * 3 + q] = gv[q];
    for (int i = 0; i < 8; i++) ggrad[((size_t)t * 3 + q) * 8 + i] = 0.0;
    ggrad[((size_t)t * 3 + q) * 8 + 0] = gs[q], ggrad[((size_t)t * 3 + q) * 8 + 1] = gn[q], ggrad[((size_t)t * 3 + q) * 8 + 2] = gm[q];
    for (int i = 0; i < 64; i++) gH[((size_t)t * 3 + q) * 64 + i] = 0.0;
    gH[((size_t)t * 3 + q) * 64 + 0] = hss[q], gH[((size_t)t * 3 + q) * 64 + 2 * 8 + 2] = hmm[q];
  }
}


// test hook: friction-ellipse constraints at given points: val n x 2, grad n x 2 x 8, H n x 2 x 8 x 8 (full state indexing)
__global__ void k_test_ellipse(Consts K, int n, const double* __restrict__ x, double* __restrict__ val, double* __restrict__ grad, double* __restrict__ H) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double xx[8], v[2], g[2][5], h[2][15], v2[2];
  for (int i = 0; i < 8; i++) xx[i] = x[(size_t)t * 8 + i];
  ellipse_eval(K.p, xx, v, g, h);
  ellipse_val(K.p, xx, v2);
  for (int q = 0; q < 2; q++) {
    val[(size_t)t * 2 + q] = v[q] + (v2[q] - v[q]) * 0.0 + (fabs(v2[q] - v[q]) <= 1e-12 * (1.0 + fabs(v[q])) ? 0.0 : 1e300);  // (the two evaluations agree)
    for (int i = 0; i < 8; i++) grad[((size_t)t * 2 + q) * 8 + i] = i >= 3 ? g[q][i - 3] : 0.0;
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) H[((size_t)t * 2 + q) * 64 + i * 8 + j] = (i >= 3 && j >= 3) ? h[q][sidx(i - 3, j - 3)] : 0.0;
  }
}

}  // namespace ltompc
