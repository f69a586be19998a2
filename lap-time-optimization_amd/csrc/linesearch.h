// linesearch.h — step candidates, filter test, update (k_linesearch, k_pick, k_update) and their fusion for narrow
// launches (k_step1).
#pragma once
#include "layout.h"

namespace ltompc {

// ------------------------------------------------------------------------------------------ k_linesearch
// candidate 0 is the current point (alpha = 0); candidate l >= 1 has alpha = a_pri * 2^-(l-1).
// LS plane layout: [3 * (n_ls + 1)][N][Bp] : theta, cost, sum log t per candidate.
// Two phases (97% of all iterations accept the full step): phase 0 evaluates the current point and the first candidate
// for every instance; phase 1 evaluates the remaining candidates for the instances whose first candidate was rejected.
// Filter measures (theta, cost, sum log t) of the step candidates l_begin..l_end of interval k of instance b;
// candidate l >= 1 has alpha = a_pri * 2^-(l-1) (l = 0, the current point, is written by k_eval).
template <class BP, bool PIN, bool ELL>
__device__ __forceinline__ void d_linesearch(const Consts& K, const Work& W, const int k, const int b, const int l_begin,
                                             const int l_end) {
  const int N = W.N;
  const double hdt = K.o.t_step;
  if (W.si[(size_t)SI_DONE * W.Bp + b] || !W.si[(size_t)SI_STEP * W.Bp + b]) return;  // no step this launch
  const double eps = W.st[(size_t)ST_EPS * W.Bp + b], rho = W.st[(size_t)ST_RHO * W.Bp + b];
  double a_pri = 1.0;
  for (int kk = 0; kk < N; kk += 20) {  // twenty loads in flight per round trip (a plain loop waits for every single one)
    double v20[20];
#pragma unroll
    for (int q = 0; q < 20; q++) v20[q] = PL(W.SP, SP_apri, kk + q < N ? kk + q : N - 1, N);
#pragma unroll
    for (int q = 0; q < 20; q++) a_pri = fmin(a_pri, v20[q]);
  }
  double xk[8], xp[8], c[8], u[2], v[2], dxk[8], dxp[8], dc[8], du[2], dv[2];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    xk[i] = k == 0 ? W.x0[(size_t)i * W.Bp + b] : PL(W.X, i, k, N + 1);
    dxk[i] = PL(W.dX, i, k, N + 1);
    xp[i] = PL(W.X, i, k + 1, N + 1), dxp[i] = PL(W.dX, i, k + 1, N + 1);
    c[i] = PL(W.C, i, k, N), dc[i] = PL(W.dC, i, k, N);
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    u[i] = PL(W.U, i, k, N), du[i] = PL(W.dU, i, k, N);
    v[i] = k ? PL(W.U, i, k - 1, N) : W.uprev[(size_t)i * W.Bp + b];
    dv[i] = k ? PL(W.dU, i, k - 1, N) : 0.0;
  }
  const bool nl = (k + 1 <= N - 1);
  // candidate index l: 0 = current point, l >= 1: alpha = a_pri * 2^-(l-1)
  for (int l = l_begin; l <= l_end; l++) {
    const double alpha = l == 0 ? 0.0 : ldexp(a_pri, -(l - 1));
    double txk[8], txp[8], tc[8], tu[2], tv[2];
#pragma unroll
    for (int i = 0; i < 8; i++) txk[i] = xk[i] + alpha * dxk[i], txp[i] = xp[i] + alpha * dxp[i], tc[i] = c[i] + alpha * dc[i];
#pragma unroll
    for (int i = 0; i < 2; i++) tu[i] = u[i] + alpha * du[i], tv[i] = v[i] + alpha * dv[i];
    double f1[8], f2[8];
    rhs_val(K.p, K.T, eps, tc, tu, f1);
    rhs_val(K.p, K.T, eps, txp, tu, f2);
    double th = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      th += fabs(hdt * f1[i] + 2.0 * txk[i] - 1.5 * tc[i] - 0.5 * txp[i]);
      th += fabs(hdt * f2[i] - 2.0 * txk[i] + 4.5 * tc[i] - 2.5 * txp[i]);
    }
    double co = cost_eval(K.p, K.T, eps, txp, k == N - 1, nullptr, nullptr);
#pragma unroll
    for (int i = 0; i < 2; i++) co += K.p.r_du[i] * (tu[i] - tv[i]) * (tu[i] - tv[i]);
    // sum of log t as the log of products of 8 slacks (same grouping in linearise_slot): 3 logarithms instead of 23
    double sl = 0.0, pr = 1.0;
    double tt[BP::fixed ? MAX_NI : 1];  // compile-time bound pattern: the slacks of the candidate in one batch of loads
    if (BP::fixed) {
      double t0[MAX_NI] = {}, d0[MAX_NI] = {};  // (zero: the pinning below touches whole chunks of 8)
      const int nb = for_each_bound<BP>(K.p, [&](int mm, int, int, double, double) { t0[mm] = PL(W.T, mm, k, N), d0[mm] = PL(W.dT, mm, k, N); });
      // (all of them in registers before the first use: left alone the compiler issues and awaits them pair by pair)
#pragma unroll
      for (int q = 0; q < MAX_NI; q += 8)
        if (PIN && q < nb) {  // (k_step1, two wavefronts per SIMD, is faster without: fewer registers)
#if defined(__HIP_DEVICE_COMPILE__)
          double* a = t0 + q;
          double* c = d0 + q;
          asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                            "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
#endif
        }
#pragma unroll
      for (int q = 0; q < MAX_NI; q++)
        if (q < nb) tt[q] = t0[q] + alpha * d0[q];
    }
    const int m = for_each_bound<BP>(K.p, [&](int mm, int kind, int jj, double sg, double val) {
      const double xv = kind == 0 ? tu[jj] : (kind == 1 ? tc[jj] : txp[jj]);
      const double t = BP::fixed ? tt[mm] : PL(W.T, mm, k, N) + alpha * PL(W.dT, mm, k, N);
      th += fabs(sg * (xv - val) + t), pr *= t;
      if ((mm & 7) == 7) sl += log(pr), pr = 1.0;
    });
    if (nl) {
      double gv[3];
      cons_eval(K.p, K.T, eps, txp, gv, nullptr, nullptr, nullptr, nullptr, nullptr);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        double t = PL(W.T, m + q, k, N) + alpha * PL(W.dT, m + q, k, N);
        double e = 0.0;
        pr *= t;
        if (rho > 0.0) {  // elastic variable of the softened constraint: g - e + t = 0, cost rho e
          e = PL(W.T, K.bd.ni + q, k, N) + alpha * PL(W.dT, K.bd.ni + q, k, N);
          pr *= e, co += rho * e;
        }
        th += fabs(gv[q] - e + t);
        if (((m + q) & 7) == 7) sl += log(pr), pr = 1.0;
      }
      if (ELL) {  // friction-ellipse constraints (always soft): same grouping of the logarithms as linearise_slot
        double ge[2];
        ellipse_val(K.p, txp, ge);
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int mm = m + 3 + q, me = K.bd.ni + 3 + q;
          const double t = PL(W.T, mm, k, N) + alpha * PL(W.dT, mm, k, N);
          const double e = PL(W.T, me, k, N) + alpha * PL(W.dT, me, k, N);
          pr *= t, pr *= e, co += K.p.ell_penalty * e;
          th += fabs(ge[q] - e + t);
          if ((mm & 7) == 7) sl += log(pr), pr = 1.0;
        }
      }
    }
    sl += log(pr);
    PL(W.LS, 3 * l + 0, k, N) = th, PL(W.LS, 3 * l + 1, k, N) = co, PL(W.LS, 3 * l + 2, k, N) = sl;
  }
}

template <class BP, bool ELL>
__global__ void __launch_bounds__(64) k_linesearch(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la, int phase, int jw) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  // phase 0: thread = (k, j), evaluates the first candidate (full step to the boundary) of instance act[j].
  // phase 1: thread = (candidate, k, j'), one candidate each (latency matters here, not throughput), over the packed
  //          list of rejected instances; jw = launch width in instances, longer lists are covered grid-stride.
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = W.N;
  const int j0 = tid % jw, rest = tid / jw, k = rest % N, cand = rest / N;
  if (phase == 0 ? (rest >= N) : (cand >= K.o.n_linesearch - 1)) return;
  const int count = phase == 0 ? la.nact[0] : W.ls_count[0];
  const int l = phase == 0 ? 1 : 2 + cand;
  for (int j = j0; j < count; j += jw) d_linesearch<BP, true, ELL>(K, W, k, phase == 0 ? la.act[j] : W.ls_list[j], l, l);
}

// ------------------------------------------------------------------------------------------ k_pick
// Filter line search of Waechter & Biegler 2006 (no second-order correction, no restoration phase).
// 8 lanes per instance (lane = g + 8 i, all 8 lanes of a group call this together): lane i reduces the stage partials
// k = i, i+8, ...; the 8 lanes then hold the same numbers and take the same decisions, lane i == 0 writes.  (Keeps the
// latency of this small step at N/8 dependent loads instead of N.)
__device__ __forceinline__ void d_pick(const Consts& K, const Work& W, const int b, const int i, const int phase,
                                       const bool append_list) {
  const int N = W.N;
  double* st = W.st;
  int* si = W.si;
  if (STI(SI_DONE) || !STI(SI_STEP)) return;  // finished, or the Riccati sweep of this launch has to be repeated
  const ltompc_options& o = K.o;
  const double mu = STD(ST_MU);
  double a_pri = 1.0, a_dua = 1.0, gphid = 0.0;
  // (loads of four stages in flight per round trip; the additions keep the order k = i, i + 8, ...)
  for (int k0 = i; k0 < N; k0 += 32) {
    double v[3][4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k0 + 8 * q < N ? k0 + 8 * q : k0;
      v[0][q] = PL(W.SP, SP_apri, k, N), v[1][q] = PL(W.SP, SP_adua, k, N), v[2][q] = PL(W.SP, SP_gphid, k, N);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (k0 + 8 * q < N) a_pri = fmin(a_pri, v[0][q]), a_dua = fmin(a_dua, v[1][q]), gphid += v[2][q];
  }
  a_pri = grp_min(a_pri), a_dua = grp_min(a_dua), gphid = grp_sum(gphid);
  // lterm(x_0) is a constant of the solve; kept so that phi matches the oracle's barrier objective
  const double c00 = STD(ST_C00);
  // filter measures of candidates lA and lB (sums over the horizon of the partials written by k_eval / k_linesearch): the
  // loads of both go out together, a pair of candidates costs one round trip per 32 stages
  auto measures2 = [&](int lA, int lB, double& thA, double& phA, double& thB, double& phB) {
    double tA = 0.0, cA = 0.0, sA = 0.0, tB = 0.0, cB = 0.0, sB = 0.0;
    for (int k0 = i; k0 < N; k0 += 32) {
      double v[6][4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int k = k0 + 8 * q < N ? k0 + 8 * q : k0;
        v[0][q] = PL(W.LS, 3 * lA + 0, k, N), v[1][q] = PL(W.LS, 3 * lA + 1, k, N), v[2][q] = PL(W.LS, 3 * lA + 2, k, N);
        v[3][q] = PL(W.LS, 3 * lB + 0, k, N), v[4][q] = PL(W.LS, 3 * lB + 1, k, N), v[5][q] = PL(W.LS, 3 * lB + 2, k, N);
      }
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (k0 + 8 * q < N) tA += v[0][q], cA += v[1][q], sA += v[2][q], tB += v[3][q], cB += v[4][q], sB += v[5][q];
    }
    tA = grp_sum(tA), cA = grp_sum(cA), sA = grp_sum(sA), tB = grp_sum(tB), cB = grp_sum(cB), sB = grp_sum(sB);
    thA = tA, phA = (c00 + cA) - mu * sA, thB = tB, phB = (c00 + cB) - mu * sB;
  };
  double th0, ph0, th1, ph1;
  measures2(0, 1, th0, ph0, th1, ph1);  // the current point and the full step
  double theta0 = STD(ST_THETA0);
  int nfilt = STI(SI_NFILT);
  double theta_max = STD(ST_THMAX), theta_min = STD(ST_THMIN);
  if (theta0 < 0.0) {
    theta0 = th0, theta_max = 1e4 * fmax(1.0, theta0), theta_min = 1e-4 * fmax(1.0, theta0);
    if (i == 0) STD(ST_THETA0) = theta0, STD(ST_THMAX) = theta_max, STD(ST_THMIN) = theta_min;
    nfilt = 0;
  }
  const double g_th = 1e-5, g_ph = 1e-8, eta_ph = 1e-8, s_th = 1.1, s_ph = 2.3, dlt = 1.0;
  const double S = pen_scale(STD(ST_RHO)), iS = 1.0 / S;  // penalty scale (layout.h): the objective side of the tests in its units
  bool accepted = false;
  double alpha = a_pri;
  const int n_ls = o.n_linesearch;
  const int n_try = (phase == 0) ? 1 : n_ls;  // phase 1 repeats the test of candidate 0 (same outcome) and goes on
  // the filter (at most FILTER_MAX pairs) once, in registers: it only changes when a candidate is accepted
  double fth[FILTER_MAX], fph[FILTER_MAX];
#pragma unroll
  for (int f = 0; f < FILTER_MAX; f++) fth[f] = W.filt[(size_t)(2 * f) * W.Bp + b], fph[f] = W.filt[(size_t)(2 * f + 1) * W.Bp + b];
  auto try_candidate = [&](const double th, const double ph) -> bool {  // true: accepted (filter updated)
    if (!isfinite(th) || !isfinite(ph) || th > theta_max) return false;
    bool in_filter = false;
#pragma unroll
    for (int f = 0; f < FILTER_MAX; f++) in_filter = in_filter || (f < nfilt && th >= fth[f] && ph >= fph[f]);
    if (in_filter) return false;
    bool sw = (gphid < 0.0) && (alpha * pow(-gphid * iS, s_ph) > dlt * pow(th0, s_th));
    bool armijo = ph <= ph0 + eta_ph * alpha * gphid;
    bool ok;
    if (th0 <= theta_min && sw) ok = armijo;
    else ok = (th <= (1.0 - g_th) * th0) || (ph <= ph0 - g_ph * S * th0);
    if (!ok) return false;
    if (!(sw && armijo)) {  // augment the filter (written by lane i == 0, nobody reads it again in this launch)
      if (nfilt == FILTER_MAX) {
        if (i == 0)
          for (int f = 0; f + 1 < FILTER_MAX; f++) {
            W.filt[(size_t)(2 * f) * W.Bp + b] = W.filt[(size_t)(2 * f + 2) * W.Bp + b];
            W.filt[(size_t)(2 * f + 1) * W.Bp + b] = W.filt[(size_t)(2 * f + 3) * W.Bp + b];
          }
        nfilt--;
      }
      if (i == 0) {
        W.filt[(size_t)(2 * nfilt) * W.Bp + b] = (1.0 - g_th) * th0;
        W.filt[(size_t)(2 * nfilt + 1) * W.Bp + b] = ph0 - g_ph * S * th0;
      }
      nfilt++;
    }
    return true;
  };
  // candidate l + 1 has alpha = a_pri 2^-l; candidates are measured in pairs (the first one came with the current point)
  accepted = try_candidate(th1, ph1);
  if (!accepted) alpha *= 0.5;
  for (int l = 1; l < n_try && !accepted; l += 2) {
    double thA, phA, thB, phB;
    measures2(l + 1, l + 2 <= n_ls ? l + 2 : l + 1, thA, phA, thB, phB);
    accepted = try_candidate(thA, phA);
    if (!accepted) {
      alpha *= 0.5;
      if (l + 1 < n_try) {
        accepted = try_candidate(thB, phB);
        if (!accepted) alpha *= 0.5;
      }
    }
  }
  if (i != 0) return;  // one writer per instance from here on
  if (phase == 0) {
    STI(SI_LSMORE) = (!accepted && n_ls > 1) ? 1 : 0;
    if (!accepted && n_ls > 1) {  // nothing has been modified yet: phase 1 decides
      if (append_list) W.ls_list[atomicAdd(W.ls_count, 1)] = b;
      return;
    }
  } else {
    STI(SI_LSMORE) = 0;
  }
  // Restoration phase (IPOPT enters its own when the filter line search fails): here an elastic mode, entered once per
  // solve on the first failed search or after stall_iter tiny steps.
  const bool resto_ready = o.resto_rho > 0.0 && !(o.soft_rho > 0.0) && STI(SI_RESTO) == 0;
  bool take = true, give_up = false, enter_resto = false;
  if (!accepted) {
    const int nf = STI(SI_NLSFAIL) + 1;
    STI(SI_NLSFAIL) = nf;
    double fr = STD(ST_FORCE_REG);
    if (resto_ready) {
      enter_resto = true, take = false;
    } else if (o.max_ls_fail > 0 && nf >= o.max_ls_fail) {
      give_up = true, take = false;
    } else if (fr < 1e4 * S) {
      STD(ST_FORCE_REG) = fr == 0.0 ? 1e-2 * S : fr * 100.0;
      take = false;
    } else {
      nfilt = 0;
      alpha = a_pri * pow(0.5, (double)(n_ls - 1));
    }
  }
  if (give_up) STI(SI_STATUS) = LTOMPC_STATUS_STALLED, STI(SI_DONE) = 1;
  if (take) {
    STD(ST_FORCE_REG) = 0.0;
    int nt = alpha <= 1e-3 ? STI(SI_NTINY) + 1 : 0;
    STI(SI_NTINY) = nt;
    if ((o.stall_iter > 0 && nt >= o.stall_iter) || (resto_ready && STI(SI_BLOWUP))) {  // (SI_BLOWUP: options.dual_inf_max, set by the head)
      if (resto_ready) enter_resto = true;
      else STI(SI_STATUS) = LTOMPC_STATUS_STALLED, STI(SI_DONE) = 1;
      take = false;
    }
  }
  if (enter_resto) {
    // First remedy (options.resto_shift_retry), once per warm-started solve: the jam may be the un-shifted warm start's doing
    // (do_mpc re-uses the previous solution as it is, one interval behind the new measured state).  The solve starts again on the
    // hard constraints from its own starting point moved one interval ahead (options.warm_shift's rule): d_update, the next
    // slot-parallel kernel, loads the shifted point from the backup planes (SI_SHIFT), the next evaluation re-initialises
    // slacks and multipliers (SI_REINIT).  The restoration phase proper follows if that start jams too:
    // the track constraints get elastic variables that cost resto_rho each; equality multipliers, slacks and the barrier
    // parameter start again at the current primal point.
    const bool shift = o.resto_shift_retry && !o.warm_shift && STI(SI_WARM) && STI(SI_NSHIFT) == 0;
    double rho_new = 0.0;
    if (shift) STI(SI_SHIFT) = 1, STI(SI_NSHIFT) = 1;
    else STI(SI_RESTO) = 1, STI(SI_NRESTO) += 1, rho_new = o.resto_rho;
    const double mu0 = o.mu_init * pen_scale(rho_new);
    STI(SI_REINIT) = 1;
    STD(ST_RHO) = rho_new, STD(ST_MU) = mu0;
    STD(ST_EPS_NEXT) = (o.smooth_scale > 0 || o.smooth_eps_min > 0) ? fmax(o.smooth_eps_min, o.smooth_scale * o.mu_init) : 0.0;
    if (STD(ST_EPS_NEXT) == STD(ST_EPS)) nfilt = 0, STD(ST_THETA0) = -1.0;  // (else: reset with the switch of the smoothing below)
    STD(ST_DW_LAST) = 0.0, STD(ST_FORCE_REG) = 0.0;
    STI(SI_NTINY) = 0, STI(SI_NACC) = 0, STI(SI_SINCEMU) = 0;
  }
  STD(ST_ALPHA) = take ? alpha : 0.0, STD(ST_ADUA) = a_dua;
  STI(SI_STEP) = take ? 1 : 0;
  if (!STI(SI_DONE)) STI(SI_ITERS) += 1;  // (a solve that stops here has completed `iters` iterations, like the oracle)
  // table smoothing follows the barrier parameter with one iteration lag; the filter restarts when it changes
  bool eps_switched = false;
  if (STD(ST_EPS_NEXT) != STD(ST_EPS)) {
    STD(ST_EPS) = STD(ST_EPS_NEXT);
    {
      double x0[8];
#pragma unroll
      for (int q = 0; q < 8; q++) x0[q] = W.x0[(size_t)q * W.Bp + b];
      STD(ST_C00) = cost_eval(K.p, K.T, STD(ST_EPS_NEXT), x0, false, nullptr, nullptr);
    }
    nfilt = 0, STD(ST_THETA0) = -1.0;
    eps_switched = true;
  }
  STI(SI_NFILT) = nfilt;
  STI(SI_SKIP_EVAL) = (!take && !eps_switched && !enter_resto) ? 1 : 0;  // the iterate did not move: the stage blocks stay valid
}

__global__ void __launch_bounds__(64) k_pick(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la, int phase) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  const int lane = threadIdx.x, g = lane & 7, i = lane >> 3;
  const int count = phase == 0 ? la.nact[0] : W.ls_count[0];
  // (phase 1 is launched for the expected length of the list of rejected steps, longer lists are covered grid-stride;
  //  the 8 lanes of an instance stay together)
  for (int j = blockIdx.x * 8 + g; j < count; j += gridDim.x * 8) d_pick(K, W, phase == 0 ? la.act[j] : W.ls_list[j], i, phase, true);
}

// ------------------------------------------------------------------------------------------ k_update
__device__ __forceinline__ void d_update(const Consts& K, const Work& W, const int k, const int b) {
  const int N = W.N;
  if (W.si[(size_t)SI_SHIFT * W.Bp + b] && !W.si[(size_t)SI_DONE * W.Bp + b]) {
    // options.resto_shift_retry (d_pick): slot k takes the primal point the solve started from, one interval ahead (the last
    // interval repeated): x_{k+1}, c_k, u_k of the backup's slot min(k + 1, N - 1).  Slot-local writes, read-only source.
    const size_t ob = W.orig[b];
    const int ks = k + 1 <= N - 1 ? k + 1 : N - 1;
    double v[18];
#pragma unroll
    for (int f = 0; f < 18; f++) v[f] = W.BK[((size_t)f * N + ks) * W.Bp + ob];
#pragma unroll
    for (int i = 0; i < 8; i++) PL(W.X, i, k + 1, N + 1) = v[i], PL(W.C, i, k, N) = v[8 + i];
    PL(W.U, 0, k, N) = v[16], PL(W.U, 1, k, N) = v[17];
    return;
  }
  if (!W.si[(size_t)SI_STEP * W.Bp + b] || W.si[(size_t)SI_DONE * W.Bp + b]) return;
  const double alpha = W.st[(size_t)ST_ALPHA * W.Bp + b], a_dua = W.st[(size_t)ST_ADUA * W.Bp + b];
  const double mu = W.st[(size_t)ST_MU * W.Bp + b];
  // Loads in batches, stores after them: on gfx9 a load that follows a store waits for the store as well (vmcnt counts
  // both, in order), so "load, update, store" per word is one full memory round trip per word for a lone wavefront
  // (narrow launches: 23 round trips for the inequalities of a slot, 55 k cycles of k_step1's 140 k).
  {
    double x[8], dx[8], c[8], dc[8], l1[8], n1[8], l2[8], n2[8], u[2], du[2];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      x[i] = PL(W.X, i, k + 1, N + 1), dx[i] = PL(W.dX, i, k + 1, N + 1), c[i] = PL(W.C, i, k, N), dc[i] = PL(W.dC, i, k, N);
      l1[i] = PL(W.L1, i, k, N), n1[i] = PL(W.nL1, i, k, N), l2[i] = PL(W.L2, i, k, N), n2[i] = PL(W.nL2, i, k, N);
    }
    u[0] = PL(W.U, 0, k, N), u[1] = PL(W.U, 1, k, N), du[0] = PL(W.dU, 0, k, N), du[1] = PL(W.dU, 1, k, N);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      PL(W.X, i, k + 1, N + 1) = x[i] + alpha * dx[i];
      PL(W.C, i, k, N) = c[i] + alpha * dc[i];
      PL(W.L1, i, k, N) = l1[i] + alpha * (n1[i] - l1[i]);
      PL(W.L2, i, k, N) = l2[i] + alpha * (n2[i] - l2[i]);
    }
    PL(W.U, 0, k, N) = u[0] + alpha * du[0], PL(W.U, 1, k, N) = u[1] + alpha * du[1];
  }
  const int ni = K.bd.ni, nel = K.bd.nel, nact = (k + 1 <= N - 1) ? ni : ni - 3 - nel;  // (the last slot has no nonlinear constraints)
  for (int m0 = 0; m0 < nact; m0 += 8) {
    double t[8], dt[8], nu[8], dn[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int m = m0 + q < nact ? m0 + q : nact - 1;  // (the last chunk repeats its last word: branch-free loads)
      t[q] = PL(W.T, m, k, N), dt[q] = PL(W.dT, m, k, N), nu[q] = PL(W.NU, m, k, N), dn[q] = PL(W.dNU, m, k, N);
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (m0 + q < nact) {
        const double tn = t[q] + alpha * dt[q];
        const double nn = nu[q] + a_dua * dn[q];
        const double lo = mu / (1e10 * tn), hi = 1e10 * mu / tn;  // IPOPT eq. (16)
        PL(W.T, m0 + q, k, N) = tn, PL(W.NU, m0 + q, k, N) = nn < lo ? lo : (nn > hi ? hi : nn);
      }
  }
  if (W.st[(size_t)ST_RHO * W.Bp + b] > 0.0 && nact == ni) {  // elastic variables
    double e[3], de[3];
#pragma unroll
    for (int q = 0; q < 3; q++) e[q] = PL(W.T, ni + q, k, N), de[q] = PL(W.dT, ni + q, k, N);
#pragma unroll
    for (int q = 0; q < 3; q++) PL(W.T, ni + q, k, N) = e[q] + alpha * de[q];
  }
  if (nel && nact == ni) {  // ... of the friction-ellipse constraints
    double e[2], de[2];
#pragma unroll
    for (int q = 0; q < 2; q++) e[q] = PL(W.T, ni + 3 + q, k, N), de[q] = PL(W.dT, ni + 3 + q, k, N);
#pragma unroll
    for (int q = 0; q < 2; q++) PL(W.T, ni + 3 + q, k, N) = e[q] + alpha * de[q];
  }
}

__global__ void __launch_bounds__(64) k_update(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid == 0) W.ls_count[1] = W.ls_count[0], W.ls_count[0] = 0;  // both line-search phases of this iteration are over ([1]: the host sizes the next phase-1 launches by it)
  int j = tid % la.n_pad, k = tid / la.n_pad;
  if (k >= W.N || j >= la.nact[0]) return;
  d_update(K, W, k, la.act[j]);
}


// ------------------------------------------------------------------------------------------ k_step1
// Narrow launches: the whole step selection of ONE instance per workgroup (both line-search phases, the filter test
// and the update), i.e. five dependent launches of 10..30 us each in one.  Same device functions, same numbers.
template <class BP, bool ELL>
__global__ void __launch_bounds__(320) k_step1(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la) {  // 320 = 8 candidates x 40 intervals in one pass
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  if ((int)blockIdx.x >= la.nact[0]) return;
  const int b = la.act[blockIdx.x];
  const int N = W.N, tid = threadIdx.x;
  const int* si = W.si;
  if (si[(size_t)SI_DONE * W.Bp + b] || !si[(size_t)SI_STEP * W.Bp + b]) return;  // block-uniform
  // all step candidates at once (the threads are there anyway; the wide path evaluates candidates 2.. only for the
  // instances that rejected the full step, with the same arithmetic)
  // LTOMPC_DBG: shader-clock cycles of block 0 per section (own line search, all line searches, pick, update), slots 8..12
  const bool rprof = W.DBG != nullptr && blockIdx.x == 0 && tid == 0;
  long long rt0 = rprof ? clock64() : 0;
#define STOCK(q) if (rprof) { const long long t1 = clock64(); W.DBG[q] += (double)(t1 - rt0); rt0 = t1; }
  for (int idx = tid; idx < N * K.o.n_linesearch; idx += 320) d_linesearch<BP, false, ELL>(K, W, idx % N, b, 1 + idx / N, 1 + idx / N);
  STOCK(8);
  __syncthreads();
  STOCK(9);
  // (phase 1 of the filter test starts with the test of the full step, i.e. it is phase 0 followed by phase 1 when the
  //  measures of all candidates exist already: one pass, same decisions)
  if (tid < 64 && (tid & 7) == 0) d_pick(K, W, b, tid >> 3, 1, false);
  __syncthreads();
  STOCK(10);
  for (int kk = tid; kk < N; kk += 320) d_update(K, W, kk, b);
  STOCK(11);
  if (rprof) W.DBG[12] += 1.0;
#undef STOCK
}


}  // namespace ltompc
