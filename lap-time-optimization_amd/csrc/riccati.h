// riccati.h — termination test, barrier update and the Riccati sweep: one thread per instance (k_riccati, reference
// implementation), 8 instances per wavefront (k_riccati8), one instance per wavefront (k_riccati1).
#pragma once
#include "layout.h"

namespace ltompc {

// ------------------------------------------------------------------------------------------ decisions of the head
// Shared by the three Riccati kernels (one writer per instance calls them).
//
// Node-0 rule (options.node0_check): do_mpc registers the track constraints at node 0 too (controller.py:69-70); they only
// involve the measured state, so they cannot change the minimiser, but a measured state outside the band leaves the
// reference's NLP without a feasible point.  A converged status becomes INFEASIBLE (ST_VIOL = g(x0)) when g(x0) >
// acceptable_tol, SOLVED becomes ACCEPTABLE when tol < g(x0) (IPOPT's error cannot fall below it).
__device__ __forceinline__ int node0_rule(const ltompc_options& o, const double g0, const int term, int& node0) {
  node0 = 0;
  if (!o.node0_check || o.soft_rho > 0.0 || !(term == LTOMPC_STATUS_SOLVED || term == LTOMPC_STATUS_ACCEPTABLE)) return term;
  if (g0 > o.acceptable_tol) {
    node0 = term + 1;
    return LTOMPC_STATUS_INFEASIBLE;
  }
  if (g0 > o.tol && term == LTOMPC_STATUS_SOLVED) {
    node0 = term + 1;
    return LTOMPC_STATUS_ACCEPTABLE;
  }
  return term;
}
// The solve of instance b starts again from its current primal point (the next evaluation kernel re-initialises the slots:
// SI_REINIT): slacks / multipliers / elastic variables for penalty `rho`, equality multipliers 0, barrier at mu_init (in the
// units of the penalty scale), filter and regularisation history dropped.  This launch does no sweep for the instance.
// Used by the penalty escalation of the restoration phase and by the fallback of a tuned warm start; d_pick has the same
// block for the entry of the restoration phase.
__device__ __forceinline__ void restart_from_primal(const Consts& K, const Work& W, const int b, const double rho) {
  double* st = W.st;
  int* si = W.si;
  const ltompc_options& o = K.o;
  const double mu0 = o.mu_init * pen_scale(rho);
  const double eps = (o.smooth_scale > 0 || o.smooth_eps_min > 0) ? fmax(o.smooth_eps_min, o.smooth_scale * o.mu_init) : 0.0;
  STD(ST_RHO) = rho, STD(ST_MU) = mu0, STD(ST_EPS_NEXT) = eps;
  if (eps != STD(ST_EPS)) {  // (d_pick, which switches the smoothing otherwise, does not see the instance in this launch)
    STD(ST_EPS) = eps;
    double x0[8];
#pragma unroll
    for (int q = 0; q < 8; q++) x0[q] = W.x0[(size_t)q * W.Bp + b];
    STD(ST_C00) = cost_eval(K.p, K.T, eps, x0, false, nullptr, nullptr);
  }
  STI(SI_REINIT) = 1;
  STI(SI_NFILT) = 0, STD(ST_THETA0) = -1.0;
  STD(ST_DW_LAST) = 0.0, STD(ST_FORCE_REG) = 0.0;
  STI(SI_NTINY) = 0, STI(SI_NACC) = 0, STI(SI_SINCEMU) = 0;
  STI(SI_STEP) = 0, STI(SI_SKIP_EVAL) = 0;
}

// ------------------------------------------------------------------------------------------ k_riccati
// One thread per instance.  State of the recursion is (x_k, v_k = u_{k-1}) because do_mpc's rterm penalises
// u_k - u_{k-1} (controller.py:40-41): stage cost r |u_k - v_k|^2, v_{k+1} = u_k.
__global__ void __launch_bounds__(64) k_riccati(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la, int it_index) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= la.nact[0]) return;
  const int b = la.act[j];
  const int N = W.N;
  double* st = W.st;
  int* si = W.si;
  if (STI(SI_DONE)) return;
  const ltompc_options& o = K.o;
  // ---- reduce residual partials, KKT error, termination (IPOPT eq. (5),(6)) ----
  double rd = 0.0, rp = 0.0, cmax = 0.0, cmin = 1e300, smult = 0.0, emax = 0.0;
  double obj;
  obj = STD(ST_C00);  // lterm(x_0), kept by k_init / d_pick
  for (int k = 0; k < N; k++) {
    rd = fmax(rd, PL(W.RS, RS_rd, k, N)), rp = fmax(rp, PL(W.RS, RS_rp, k, N));
    cmax = fmax(cmax, PL(W.RS, RS_cmax, k, N)), cmin = fmin(cmin, PL(W.RS, RS_cmin, k, N));
    smult += PL(W.RS, RS_smult, k, N), obj += PL(W.RS, RS_cost, k, N);
    emax = fmax(emax, PL(W.RS, RS_emax, k, N));
  }
  const int n_mult = N * (2 * NX + K.bd.ni) - (3 + K.bd.nel) + (STD(ST_RHO) > 0.0 ? 3 * (N - 1) : 0) + K.bd.nel * (N - 1);  // multipliers counted (last slot has no nl constraints; elastic pairs count twice)
  double mu = STD(ST_MU);
  const double rho = STD(ST_RHO), S = pen_scale(rho), iS = 1.0 / S;  // penalty scale (layout.h): 1 unless rho > RHO_UNIT
  double s_d = fmax(o.s_max, smult * iS / n_mult) / o.s_max;
  double E0 = fmax(fmax(rd * iS / s_d, rp), cmax * iS / s_d);
  double rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
  double Emu = fmax(fmax(rd * iS / s_d, rp), rcmu * iS / s_d);
  STD(ST_E0) = E0, STD(ST_OBJ) = obj, STD(ST_VIOL) = emax;
  STI(SI_REINIT) = 0, STI(SI_SHIFT) = 0;
  int iters = STI(SI_ITERS);
  int passes = STI(SI_SWEEPS);  // (see layout.h: this kernel repeats failed sweeps inside the launch, each one a pass)
  if (passes > o.max_iter) {
    STI(SI_STATUS) = LTOMPC_STATUS_MAX_ITER, STI(SI_DONE) = 1;
    return;
  }
  int term = -1;
  if (!isfinite(E0)) term = LTOMPC_STATUS_NUMERICAL;
  else if (E0 <= o.tol) term = LTOMPC_STATUS_SOLVED;
  else {
    if (E0 <= o.acceptable_tol) {
      int na = STI(SI_NACC) + 1;
      STI(SI_NACC) = na;
      if (na >= o.acceptable_iter || (STI(SI_RESTO) == 1 && S > 1.0)) term = LTOMPC_STATUS_ACCEPTABLE;  // (escalated elastic problem: see d_head8)
    } else STI(SI_NACC) = 0;
    if (term < 0 && (iters >= o.max_iter || passes >= o.max_iter)) term = LTOMPC_STATUS_MAX_ITER;
  }
  STI(SI_SWEEPS) = ++passes;
  if (STI(SI_RESTO) == 1 && (term == LTOMPC_STATUS_SOLVED || term == LTOMPC_STATUS_ACCEPTABLE)) {  // see d_head8
    const double e_tol = term == LTOMPC_STATUS_SOLVED ? o.tol : o.acceptable_tol;
    if (emax <= e_tol) {
      STI(SI_RESTO) = 2, STD(ST_RHO) = 0.0;
      STI(SI_NACC) = 0, STI(SI_NFILT) = 0, STD(ST_THETA0) = -1.0;
      STI(SI_STEP) = 0, STI(SI_SKIP_EVAL) = 0;
      if (it_index >= 0) atomicAdd(&W.active[it_index], 1);
      return;
    }
    if (o.resto_rho_factor > 1.0 && rho < o.resto_rho_max) {  // penalty escalation, see d_head8
      restart_from_primal(K, W, b, fmin(rho * o.resto_rho_factor, o.resto_rho_max));
      STI(SI_NRESTO) += 1;
      if (it_index >= 0) atomicAdd(&W.active[it_index], 1);
      return;
    }
    term = LTOMPC_STATUS_INFEASIBLE;
  }
  {
    int node0;
    term = node0_rule(o, STD(ST_G0), term, node0);
    if (node0) STI(SI_NODE0) = node0;
    if (node0 && term == LTOMPC_STATUS_INFEASIBLE) STD(ST_VIOL) = STD(ST_G0);
  }
  if (term >= 0) {
    STI(SI_STATUS) = term, STI(SI_DONE) = 1;
    return;
  }
  if (it_index >= 0) atomicAdd(&W.active[it_index], 1);
  const bool recoverable = rho == 0.0 && STI(SI_RESTO) == 0 && o.resto_rho > 0.0;  // on the hard constraints, recovery steps still ahead
  STI(SI_BLOWUP) = (o.dual_inf_max > 0.0 && STI(SI_WARM) && recoverable && rd > o.dual_inf_max) ? 1 : 0;  // see d_head8
  // ---- monotone barrier update (IPOPT eq. (7)), in the units of the penalty scale ----
  bool mu_changed = false;
  {
    double mus = mu * iS;
    while (Emu <= o.kappa_eps * mus && mus > o.mu_min) {
      mus = fmax(o.mu_min, fmin(o.kappa_mu * mus, pow(mus, o.theta_mu)));
      mu = mus * S;
      mu_changed = true;
      rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
      Emu = fmax(fmax(rd * iS / s_d, rp), rcmu * iS / s_d);
    }
  }
  if (mu_changed) {
    STD(ST_MU) = mu;
    STD(ST_EPS_NEXT) = (o.smooth_scale > 0 || o.smooth_eps_min > 0) ? fmax(o.smooth_eps_min, o.smooth_scale * (mu * iS)) : 0.0;
    STI(SI_NFILT) = 0, STD(ST_THETA0) = -1.0;
  }
  {  // options.warm_fallback_iter, see d_head8
    const int since = mu_changed ? 0 : STI(SI_SINCEMU) + 1;
    STI(SI_SINCEMU) = since;
    if (STI(SI_WARM) && o.max_mu_stay > 0 && since >= o.max_mu_stay) {  // options.max_mu_stay, see d_head8
      if (!recoverable) {
        STI(SI_STATUS) = LTOMPC_STATUS_STALLED, STI(SI_DONE) = 1;
        return;
      }
      STI(SI_BLOWUP) = 1;
    }
    if (STI(SI_FBARMED) && since >= o.warm_fallback_iter) {
      STI(SI_FBARMED) = 0, STI(SI_NFALLBACK) += 1;
      restart_from_primal(K, W, b, rho);
      return;
    }
  }
  STD(ST_TAU) = fmax(o.tau_min, 1.0 - mu * iS);
  // ---- backward sweep, retried with Hessian regularisation until every Huu is positive definite ----
  const double r2[2] = {2.0 * K.p.r_du[0], 2.0 * K.p.r_du[1]};
  double delta_w = STD(ST_FORCE_REG);
  const double dw_last = STD(ST_DW_LAST);
  if (delta_w == 0.0 && dw_last > DW_KEEP * S) delta_w = dw_last / 3.0;  // see DESIGN.md §3 (deviation from Algorithm IC)
  int tries = 0;
  bool numerical = false;
  for (;;) {
    bool ok = true;
    double P[64], Pxv[16], Pvv[4], pp[8], pv[2];
    // terminal node block (slot N-1)
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
      for (int j = 0; j < 8; j++) P[i * 8 + j] = PG(W.QP, QP_Qx + sidx(i, j), N, QP_NF) + ((i == j) ? delta_w : 0.0);
      pp[i] = PG(W.QP, QP_qx0 + i, N, QP_NF) + mu * PG(W.QP, QP_qx1 + i, N, QP_NF);
      Pxv[i * 2] = Pxv[i * 2 + 1] = 0.0;
    }
    Pvv[0] = Pvv[1] = Pvv[2] = Pvv[3] = 0.0, pv[0] = pv[1] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) PG(W.RC, RC_P + sidx(i, j), N, RC_NF) = P[i * 8 + j];
      PG(W.RC, RC_Pxv + i * 2, N, RC_NF) = 0.0, PG(W.RC, RC_Pxv + i * 2 + 1, N, RC_NF) = 0.0;
      PG(W.RC, RC_pp + i, N, RC_NF) = pp[i];
    }
    for (int k = N - 1; k >= 0; k--) {
      double A[64], Bm[16], bv[8];
#pragma unroll
      for (int i = 0; i < 64; i++) A[i] = PG(W.QP, QP_A + i, k, QP_NF);
#pragma unroll
      for (int i = 0; i < 16; i++) Bm[i] = PG(W.QP, QP_B + i, k, QP_NF);
#pragma unroll
      for (int i = 0; i < 8; i++) bv[i] = PG(W.QP, QP_b + i, k, QP_NF);
      double PA[64], PB[16], Pb[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < 8; l++) s += P[i * 8 + l] * A[l * 8 + j];
          PA[i * 8 + j] = s;
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < 8; l++) s += P[i * 8 + l] * Bm[l * 2 + j];
          PB[i * 2 + j] = s;
        }
        double s = pp[i];
#pragma unroll
        for (int l = 0; l < 8; l++) s += P[i * 8 + l] * bv[l];
        Pb[i] = s;
      }
      double Huu[4], Hux[16], gu[2], uk[2], vk[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        uk[i] = PL(W.U, i, k, N);
        vk[i] = k ? PL(W.U, i, k - 1, N) : W.uprev[(size_t)i * W.Bp + b];
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          double s = PG(W.QP, QP_R + sidx(i, j), k, QP_NF) + Pvv[i * 2 + j];
#pragma unroll
          for (int l = 0; l < 8; l++)
            s += Bm[l * 2 + i] * PB[l * 2 + j] + Bm[l * 2 + i] * Pxv[l * 2 + j] + Pxv[l * 2 + i] * Bm[l * 2 + j];
          Huu[i * 2 + j] = s;
        }
        Huu[i * 2 + i] += r2[i] + delta_w;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          double s = PG(W.QP, QP_S + i * 8 + j, k, QP_NF);
#pragma unroll
          for (int l = 0; l < 8; l++) s += Bm[l * 2 + i] * PA[l * 8 + j] + Pxv[l * 2 + i] * A[l * 8 + j];
          Hux[i * 8 + j] = s;
        }
        double s = PG(W.QP, QP_r0 + i, k, QP_NF) + mu * PG(W.QP, QP_r1 + i, k, QP_NF) + r2[i] * (uk[i] - vk[i]) + pv[i];
#pragma unroll
        for (int l = 0; l < 8; l++) s += Bm[l * 2 + i] * Pb[l] + Pxv[l * 2 + i] * bv[l];
        gu[i] = s;
      }
      double Hxx[64], gx[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          double s = PG(W.QP, QP_Q + sidx(i, j), k, QP_NF) + ((i == j) ? delta_w : 0.0);
          if (k > 0) s += PG(W.QP, QP_Qx + sidx(i, j), k, QP_NF);
#pragma unroll
          for (int l = 0; l < 8; l++) s += A[l * 8 + i] * PA[l * 8 + j];
          Hxx[i * 8 + j] = s;
        }
        double s = PG(W.QP, QP_q0 + i, k, QP_NF) + mu * PG(W.QP, QP_q1 + i, k, QP_NF);
        if (k > 0) s += PG(W.QP, QP_qx0 + i, k, QP_NF) + mu * PG(W.QP, QP_qx1 + i, k, QP_NF);
#pragma unroll
        for (int l = 0; l < 8; l++) s += A[l * 8 + i] * Pb[l];
        gx[i] = s;
      }
      double det = Huu[0] * Huu[3] - Huu[1] * Huu[2];
      if (!(Huu[0] > 0.0) || !(det > 1e-14 * Huu[0] * Huu[3]) || !isfinite(det)) {
        ok = false;
        break;
      }
      const double idet = 1.0 / det;  // one division per stage instead of four (same expression in the three Riccati kernels)
      double Hi[4] = {Huu[3] * idet, -Huu[1] * idet, -Huu[2] * idet, Huu[0] * idet};
      double Kx[16], Kv[4], kff[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) Kx[i * 8 + j] = -(Hi[i * 2 + 0] * Hux[0 * 8 + j] + Hi[i * 2 + 1] * Hux[1 * 8 + j]);
#pragma unroll
        for (int j = 0; j < 2; j++) Kv[i * 2 + j] = Hi[i * 2 + j] * r2[j];
        kff[i] = -(Hi[i * 2 + 0] * gu[0] + Hi[i * 2 + 1] * gu[1]);
      }
      double gv[2] = {-r2[0] * (uk[0] - vk[0]), -r2[1] * (uk[1] - vk[1])};
#pragma unroll
      for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) P[i * 8 + j] = Hxx[i * 8 + j] + Hux[0 * 8 + i] * Kx[0 * 8 + j] + Hux[1 * 8 + i] * Kx[1 * 8 + j];
#pragma unroll
        for (int j = 0; j < 2; j++) Pxv[i * 2 + j] = Hux[0 * 8 + i] * Kv[0 * 2 + j] + Hux[1 * 8 + i] * Kv[1 * 2 + j];
        pp[i] = gx[i] + Hux[0 * 8 + i] * kff[0] + Hux[1 * 8 + i] * kff[1];
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) Pvv[i * 2 + j] = ((i == j) ? r2[i] : 0.0) - r2[i] * Kv[i * 2 + j];
        pv[i] = gv[i] - r2[i] * kff[i];
      }
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < i; j++) {
          double s = 0.5 * (P[i * 8 + j] + P[j * 8 + i]);
          P[i * 8 + j] = s, P[j * 8 + i] = s;
        }
#pragma unroll
      for (int i = 0; i < 16; i++) PG(W.RC, RC_K + i, k, RC_NF) = Kx[i];
#pragma unroll
      for (int i = 0; i < 4; i++) PG(W.RC, RC_Kv + i, k, RC_NF) = Kv[i];
      PG(W.RC, RC_kff + 0, k, RC_NF) = kff[0], PG(W.RC, RC_kff + 1, k, RC_NF) = kff[1];
      if (k > 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
          for (int j = 0; j <= i; j++) PG(W.RC, RC_P + sidx(i, j), k, RC_NF) = P[i * 8 + j];
          PG(W.RC, RC_Pxv + i * 2, k, RC_NF) = Pxv[i * 2], PG(W.RC, RC_Pxv + i * 2 + 1, k, RC_NF) = Pxv[i * 2 + 1];
          PG(W.RC, RC_pp + i, k, RC_NF) = pp[i];
        }
      }
    }
    if (ok) break;
    // inertia correction schedule (Waechter & Biegler 2006, Algorithm IC)
    if (delta_w == 0.0) delta_w = dw_last == 0.0 ? o.delta_w_first * S : fmax(1e-20, dw_last / 3.0);
    else delta_w *= (dw_last == 0.0 ? 100.0 : 8.0);
    STI(SI_NREG) += 1;
    if (++tries > 40 || delta_w > 1e20) {
      numerical = true;
      break;
    }
    if (passes >= o.max_iter) {  // out of passes
      STI(SI_STATUS) = LTOMPC_STATUS_MAX_ITER, STI(SI_DONE) = 1;
      return;
    }
    STI(SI_SWEEPS) = ++passes;
  }
  if (numerical) {
    STI(SI_STATUS) = LTOMPC_STATUS_NUMERICAL, STI(SI_DONE) = 1;
    return;
  }
  if (delta_w > 0.0) STD(ST_DW_LAST) = delta_w > DW_KEEP * S ? delta_w : 0.0;
  else if (dw_last <= DW_KEEP * S) STD(ST_DW_LAST) = 0.0;
  STD(ST_DW) = delta_w;
  // ---- forward rollout ----
  double dx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dv[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 8; i++) PL(W.dX, i, 0, N + 1) = 0.0;
  for (int k = 0; k < N; k++) {
    double du[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      double s = PG(W.RC, RC_kff + i, k, RC_NF) + PG(W.RC, RC_Kv + i * 2, k, RC_NF) * dv[0] + PG(W.RC, RC_Kv + i * 2 + 1, k, RC_NF) * dv[1];
#pragma unroll
      for (int j = 0; j < 8; j++) s += PG(W.RC, RC_K + i * 8 + j, k, RC_NF) * dx[j];
      du[i] = s;
    }
    double dn[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double s = PG(W.QP, QP_b + i, k, QP_NF) + PG(W.QP, QP_B + i * 2, k, QP_NF) * du[0] + PG(W.QP, QP_B + i * 2 + 1, k, QP_NF) * du[1];
#pragma unroll
      for (int j = 0; j < 8; j++) s += PG(W.QP, QP_A + i * 8 + j, k, QP_NF) * dx[j];
      dn[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) dx[i] = dn[i], PL(W.dX, i, k + 1, N + 1) = dn[i];
    dv[0] = du[0], dv[1] = du[1];
    PL(W.dU, 0, k, N) = du[0], PL(W.dU, 1, k, N) = du[1];
  }
  STI(SI_STEP) = 1;
}

// ------------------------------------------------------------------------------------------ head of an iteration
// KKT error, termination test, restoration bookkeeping and the monotone barrier update of one instance, by the 8 lanes
// (lane = g + 8 i) that own it in d_riccati8 / d_riccati1: lane i reduces the partials of the intervals k = i, i + 8, ...;
// the sums over k are formed in the order k = 0..N-1 by every lane (identical to the serial kernel, so that all three
// produce the same bits).  Returns false when no lane of the wavefront has a sweep to do.
__device__ __forceinline__ bool d_head8(const Consts& K, const Work& W, const int i, const int b, const bool valid,
                                        const int active_slot, bool& live, bool& retry, double& mu) {
  const int N = W.N;
  double* st = W.st;
  int* si = W.si;
  const ltompc_options& o = K.o;
  live = valid && !STI(SI_DONE);
  retry = false;
  mu = 0.0;
  if (!__any(live)) return false;
  // One sweep per launch: an instance whose sweep fails the inertia test repeats it in the NEXT launch with a larger
  // delta_w (its blocks stay in HBM, k_eval skips it) instead of looping here, so that a launch never takes longer
  // than one sweep however hard the worst instance of the batch is.
  retry = live && STI(SI_RETRY);
  const int passes = STI(SI_SWEEPS);  // passes used so far (see layout.h); the budget is options.max_iter of them
  if (live && passes > o.max_iter) {
    // (a head that one sweep per launch would never have run: the host loop issues max_iter + 1 of them; reached when sweeps were
    //  repeated inside launches and the pass before was the last one - e.g. the return from the restoration phase, which is
    //  followed by a head of its own)
    if (i == 0) STI(SI_STATUS) = LTOMPC_STATUS_MAX_ITER, STI(SI_DONE) = 1;
    live = false, retry = false;
  }
  if (!__any(live)) return false;
  const double rho = STD(ST_RHO);
  double rd = 0.0, rp = 0.0, cmax = 0.0, cmin = 1e300, emax = 0.0;
  for (int k = i; k < N; k += 8) {
    rd = fmax(rd, PL(W.RS, RS_rd, k, N)), rp = fmax(rp, PL(W.RS, RS_rp, k, N));
    cmax = fmax(cmax, PL(W.RS, RS_cmax, k, N)), cmin = fmin(cmin, PL(W.RS, RS_cmin, k, N));
    emax = fmax(emax, PL(W.RS, RS_emax, k, N));
  }
  rd = grp_max(rd), rp = grp_max(rp), cmax = grp_max(cmax), cmin = grp_min(cmin), emax = grp_max(emax);
  double smult = 0.0, obj;
  obj = STD(ST_C00);  // lterm(x_0), kept by k_init / d_pick
  for (int k0 = 0; k0 < N; k0 += 8) {  // same order of additions as the serial kernel, eight loads in flight per round trip
    double sm8[8], co8[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int k = k0 + q < N ? k0 + q : N - 1;
      sm8[q] = PL(W.RS, RS_smult, k, N), co8[q] = PL(W.RS, RS_cost, k, N);
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (k0 + q < N) smult += sm8[q], obj += co8[q];
  }
  const int n_mult = N * (2 * NX + K.bd.ni) - (3 + K.bd.nel) + (rho > 0.0 ? 3 * (N - 1) : 0) + K.bd.nel * (N - 1);  // multipliers counted (the last slot has no nonlinear constraints; elastic pairs count twice)
  mu = STD(ST_MU);
  const double S = pen_scale(rho), iS = 1.0 / S;  // penalty scale (layout.h): 1 unless rho > RHO_UNIT
  double s_d = fmax(o.s_max, smult * iS / n_mult) / o.s_max;
  double E0 = fmax(fmax(rd * iS / s_d, rp), cmax * iS / s_d);
  double rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
  double Emu = fmax(fmax(rd * iS / s_d, rp), rcmu * iS / s_d);
  int term = -1;
  bool to_hard = false, escalate = false, fallback = false;
  if (valid && i == 0 && STI(SI_REINIT)) STI(SI_REINIT) = 0, STI(SI_SHIFT) = 0;  // the evaluation before this head has re-initialised the slots
  if (live && !retry) {
    int iters = STI(SI_ITERS);
    if (!isfinite(E0)) term = LTOMPC_STATUS_NUMERICAL;
    else if (E0 <= o.tol) term = LTOMPC_STATUS_SOLVED;
    else {
      int na = (E0 <= o.acceptable_tol) ? STI(SI_NACC) + 1 : 0;
      if (i == 0) STI(SI_NACC) = na;
      // (an escalated elastic problem only has to answer "is some elastic variable > 0 at the least violation": the acceptable
      //  level decides that at once - IPOPT's restoration phase does not iterate to the NLP's tolerance either; at S = 1e4 the
      //  unscaled dual residual sits at the rounding floor, 1e-4 on multipliers of 1e7, i.e. 1e-8 scaled)
      if ((na >= o.acceptable_iter || (STI(SI_RESTO) == 1 && S > 1.0)) && E0 <= o.acceptable_tol) term = LTOMPC_STATUS_ACCEPTABLE;
      if (term < 0 && (iters >= o.max_iter || passes >= o.max_iter)) term = LTOMPC_STATUS_MAX_ITER;
    }
    if (STI(SI_RESTO) == 1 && (term == LTOMPC_STATUS_SOLVED || term == LTOMPC_STATUS_ACCEPTABLE)) {
      // The elastic problem of the restoration phase has converged.  All elastic variables at (numerically) zero: its
      // solution is a KKT point of the hard-constrained NLP with the same multipliers (nu < rho): back to the hard
      // constraints, where the termination test is repeated on the hard problem's own KKT error (this launch does no
      // sweep for the instance; the pass is not counted as an iteration).  Otherwise the violation cannot be removed
      // locally: a stationary point of the infeasibility.
      const double e_tol = term == LTOMPC_STATUS_SOLVED ? o.tol : o.acceptable_tol;
      if (emax <= e_tol) to_hard = true, term = -1;
      // Some elastic variable stays > tol: at THIS penalty violating is cheaper than complying (that constraint's multiplier
      // sits at the penalty), which a feasible NLP with multipliers > rho shows as well.  The penalty of this instance goes up
      // (options.resto_rho_factor, in one step to resto_rho_max by default) and the elastic problem is solved again from the
      // current primal point, re-centred as at the entry of the phase; INFEASIBLE only at the largest penalty: a stationary
      // point of objective / resto_rho_max + violation.
      else if (o.resto_rho_factor > 1.0 && rho < o.resto_rho_max) escalate = true, term = -1;
      else term = LTOMPC_STATUS_INFEASIBLE;
    }
    int node0;
    term = node0_rule(o, STD(ST_G0), term, node0);
    if (i == 0) {
      STD(ST_E0) = E0, STD(ST_OBJ) = obj, STD(ST_VIOL) = (node0 && term == LTOMPC_STATUS_INFEASIBLE) ? STD(ST_G0) : emax;
      if (node0) STI(SI_NODE0) = node0;
      if (term >= 0) STI(SI_STATUS) = term, STI(SI_DONE) = 1;
      if (to_hard) {
        STI(SI_RESTO) = 2, STD(ST_RHO) = 0.0;
        STI(SI_NACC) = 0, STI(SI_NFILT) = 0, STD(ST_THETA0) = -1.0;
        STI(SI_STEP) = 0, STI(SI_SKIP_EVAL) = 0;
      }
      if (escalate) {
        restart_from_primal(K, W, b, fmin(rho * o.resto_rho_factor, o.resto_rho_max));
        STI(SI_NRESTO) += 1;
      }
    }
    if (term >= 0) live = false;
  }
  if (live && retry && passes >= o.max_iter) {  // out of passes while repeating a sweep
    if (i == 0) STI(SI_STATUS) = LTOMPC_STATUS_MAX_ITER, STI(SI_DONE) = 1;
    live = false;
  }
  if (live && i == 0) STI(SI_SWEEPS) = passes + 1;
  if (live && i == 0 && active_slot >= 0) atomicAdd(&W.active[active_slot], 1);
  if (to_hard || escalate) live = false;  // (counted as unfinished above)
  if (!__any(live)) return false;
  // ---- monotone barrier update, in the units of the penalty scale
  bool mu_changed = false;
  {
    double mus = mu * iS;
    while (live && !retry && Emu <= o.kappa_eps * mus && mus > o.mu_min) {
      mus = fmax(o.mu_min, fmin(o.kappa_mu * mus, pow(mus, o.theta_mu)));
      mu = mus * S;
      mu_changed = true;
      rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
      Emu = fmax(fmax(rd * iS / s_d, rp), rcmu * iS / s_d);
    }
  }
  if (live && !retry) {
    // options.warm_fallback_iter: a solve that started at the small barrier parameter of a tuned warm start (mu_init_warm) and has
    // not decreased it for that many iterations is cycling around a point that is not central; once per solve it starts again
    // from its current primal point the way a solve after a failed one starts (multipliers 0, barrier at mu_init)
    const int since = mu_changed ? 0 : STI(SI_SINCEMU) + 1;
    const bool warm = STI(SI_WARM) != 0;
    const bool recoverable = rho == 0.0 && STI(SI_RESTO) == 0 && o.resto_rho > 0.0;  // on the hard constraints, recovery steps still ahead
    // options.max_mu_stay (warm-started solves): this many iterations without a decrease of the barrier parameter - the iterates
    // wander or cycle (the filter holds FILTER_MAX pairs and forgets the oldest).  On the hard constraints the recovery steps
    // take over at the end of this iteration, elsewhere the solve ends STALLED.
    const bool stuck = warm && o.max_mu_stay > 0 && since >= o.max_mu_stay;
    fallback = !(stuck && !recoverable) && STI(SI_FBARMED) && since >= o.warm_fallback_iter;
    if (i == 0) {
      STI(SI_SINCEMU) = since;
      if (stuck && !recoverable) STI(SI_STATUS) = LTOMPC_STATUS_STALLED, STI(SI_DONE) = 1;
      // options.dual_inf_max: on the hard constraints of a warm-started solve, a dual infeasibility beyond any scale of the
      // problem (the multipliers diverge while the line search keeps accepting steps of a percent) sends the solve to its
      // recovery steps at the end of this iteration (d_pick) instead of 100+ iterations later
      STI(SI_BLOWUP) = (recoverable && (stuck || (o.dual_inf_max > 0.0 && warm && rd > o.dual_inf_max))) ? 1 : 0;
      if (mu_changed) {
        STD(ST_MU) = mu;
        STD(ST_EPS_NEXT) = (o.smooth_scale > 0 || o.smooth_eps_min > 0) ? fmax(o.smooth_eps_min, o.smooth_scale * (mu * iS)) : 0.0;
        STI(SI_NFILT) = 0, STD(ST_THETA0) = -1.0;
      }
      STD(ST_TAU) = fmax(o.tau_min, 1.0 - mu * iS);
      if (fallback) {
        STI(SI_FBARMED) = 0, STI(SI_NFALLBACK) += 1;
        restart_from_primal(K, W, b, rho);
      }
    }
    if (fallback || (stuck && !recoverable)) live = false;
  }
  return __any(live);
}

// ------------------------------------------------------------------------------------------ k_riccati8
// Wave-cooperative form of k_riccati: a wavefront = 8 instances x 8 lanes, lane (g, i) = (lane & 7, lane >> 3)
// owns ROW i of the 8x8 blocks of instance b = 8 * blockIdx.x + g.  With the instance index fastest in HBM the
// 8 lanes that read one field of 8 neighbouring instances fetch one full 64-byte sector.  Stage blocks A, B, b
// and the row-exchanged products (P A, P B, P b + p, K, P) live in LDS as [field][g] (conflict-free: a
// wave-wide ds_read_b64 touches 8 or 64 consecutive doubles).  Same arithmetic as k_riccati.
struct RicLds {
  // The stage blocks of the wavefront's 8 instances, double-buffered, as [field][g] (the layout of the QP buffer, so the
  // A, B, b the products need are read in place): lane (g, i) fetches fields i, i + 8, ... of instance g one stage
  // ahead (27 loads per lane instead of the 45 values a lane needs itself, and no second register set for them).
  double sb[2][(QP_NF + 1) * 8];  // 27 x 8 fields: lane row 7 fetches one field past the block (padding, never read)
  double PA[64][8], PB[16][8], Pb[8][8], K[16][8], Pxv[16][8];  // (the new P is exchanged through PA: P A is dead by then)
};  // 35.2 kB: four wavefronts per CU


// In a one-wavefront workgroup LDS instructions execute in program order, so exchanging data through LDS needs

struct FwdRegs {
  double K[16], Kv[4], kff[2], A[8], B[2], b;
};

// All 64 lanes of a wavefront call this together; lane (g, i) works on row i of instance b (padding lanes: valid =
// false, they shadow a real instance read-only).  active_slot >= 0: count the unfinished instances there.
__device__ __forceinline__ void d_riccati8(const Consts& K, const Work& W, RicLds& L, const int g, const int i, const int b,
                                           const bool valid, const int active_slot, const int max_sweeps) {
  const int N = W.N;
  double* st = W.st;
  int* si = W.si;
  const ltompc_options& o = K.o;
  bool live, retry;
  double mu;
  if (!d_head8(K, W, i, b, valid, active_slot, live, retry, mu)) return;
  // ---- backward sweep (whole wave in lock-step; an instance whose Huu fails retries with a larger delta_w,
  //      the others recompute the same numbers)
  const double r2[2] = {2.0 * K.p.r_du[0], 2.0 * K.p.r_du[1]};
  const double psc = pen_scale(STD(ST_RHO));  // penalty scale (layout.h): the regularisation schedule in its units
  double delta_w = STD(ST_FORCE_REG);
  const double dw_last = STD(ST_DW_LAST);
  if (delta_w == 0.0 && dw_last > DW_KEEP * psc) delta_w = dw_last / 3.0;  // see DESIGN.md §3 (deviation from Algorithm IC)
  int tries = 0;
  if (retry) delta_w = STD(ST_DW_TRY), tries = STI(SI_TRIES);
  bool numerical = false;
  // max_sweeps = 1 while the launch is wide (a launch then never takes longer than one sweep, however hard the worst
  // instance of the batch is: its further attempts happen in the following launches); a few attempts per launch
  // once only the stragglers are left
  constexpr int NPF = (QP_NF + 7) / 8;  // fields a lane fetches per stage block
  const double up0 = W.uprev[b], up1 = W.uprev[(size_t)W.Bp + b];
  for (int sweep = 0;; sweep++) {
    bool ok = true;
    double Prow[8], pxv[2], ppi, Pvv[4] = {0, 0, 0, 0}, pv[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) Prow[j] = PG(W.QP, QP_Qx + sidx(i, j), N, QP_NF) + ((i == j) ? delta_w : 0.0);
    ppi = PG(W.QP, QP_qx0 + i, N, QP_NF) + mu * PG(W.QP, QP_qx1 + i, N, QP_NF);
    pxv[0] = pxv[1] = 0.0;
    if (live) {
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (j <= i) PG(W.RC, RC_P + sidx(i, j), N, RC_NF) = Prow[j];
      PG(W.RC, RC_Pxv + i * 2, N, RC_NF) = 0.0, PG(W.RC, RC_Pxv + i * 2 + 1, N, RC_NF) = 0.0;
      PG(W.RC, RC_pp + i, N, RC_NF) = ppi;
    }
    WAVE_SYNC();
    L.Pxv[i * 2][g] = 0.0, L.Pxv[i * 2 + 1][g] = 0.0;
    // stage N-1 into buffer (N-1) & 1; inputs u_k, u_{k-1} ride along in registers
    double pf[NPF];
    {
      const double* src = &PG(W.QP, i, N - 1, QP_NF);  // fields i, i + 8, ...: 64 doubles apart
#pragma unroll
      for (int j = 0; j < NPF; j++) pf[j] = src[j * 64];
    }
#pragma unroll
    for (int j = 0; j < NPF; j++) L.sb[(N - 1) & 1][(i + 8 * j) * 8 + g] = pf[j];
    double uk[2] = {PL(W.U, 0, N - 1, N), PL(W.U, 1, N - 1, N)};
    double vk[2];
    {
      const int km = N - 2 > 0 ? N - 2 : 0;
      const double v0 = PL(W.U, 0, km, N), v1 = PL(W.U, 1, km, N);
      vk[0] = N - 1 > 0 ? v0 : up0, vk[1] = N - 1 > 0 ? v1 : up1;
    }
#pragma unroll 1
    for (int k = N - 1; k >= 0; k--) {
      // fetch stage k-1 now (branch-free: for k = 0 block 0 is fetched again and dropped), written to the other LDS
      // buffer at the end of this stage, so that its latency hides behind this stage's arithmetic
      const int kn = k > 0 ? k - 1 : 0, kv = k > 1 ? k - 2 : 0;
      const double vn0 = PL(W.U, 0, kv, N), vn1 = PL(W.U, 1, kv, N);
      WAVE_SYNC();  // the stage block written at the end of the previous stage is visible
      const double* q = L.sb[k & 1];
      const double wn = k > 0 ? 1.0 : 0.0;  // the node block of x_0 does not exist (x_0 is data; its slot holds zeros)
      double Qrow[8], Scol[2], Rm[3], rr[2];
#pragma unroll
      for (int j = 0; j < 8; j++)
        Qrow[j] = q[(QP_Q + sidx(i, j)) * 8 + g] + ((i == j) ? delta_w : 0.0) + wn * q[(QP_Qx + sidx(i, j)) * 8 + g];
      Scol[0] = q[(QP_S + i) * 8 + g], Scol[1] = q[(QP_S + 8 + i) * 8 + g];
      const double qi = q[(QP_q0 + i) * 8 + g] + mu * q[(QP_q1 + i) * 8 + g] + wn * (q[(QP_qx0 + i) * 8 + g] + mu * q[(QP_qx1 + i) * 8 + g]);
      Rm[0] = q[(QP_R + 0) * 8 + g], Rm[1] = q[(QP_R + 1) * 8 + g], Rm[2] = q[(QP_R + 2) * 8 + g];
      rr[0] = q[(QP_r0 + 0) * 8 + g] + mu * q[(QP_r1 + 0) * 8 + g], rr[1] = q[(QP_r0 + 1) * 8 + g] + mu * q[(QP_r1 + 1) * 8 + g];
#define LA(x) q[(QP_A + (x)) * 8 + g]
#define LB(x) q[(QP_B + (x)) * 8 + g]
#define Lb(x) q[(QP_b + (x)) * 8 + g]
      // 1. row i of P A, P B, P b + p.  Operands are pulled from LDS into registers in batches and then used (here: four
      //    rows of A, B, b at a time, 44 reads in flight): read where they are used, the compiler kept two or three reads
      //    in flight and a lone wavefront paid most of the LDS latency of every one of the ~250 reads of a stage
      double PAr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, PBr[2] = {0, 0}, Pbi = ppi;
      double Bm[16], bl[8];  // B_k and b_k stay in registers for steps 2 and 3
#pragma unroll
      for (int h = 0; h < 2; h++) {
        double a[32];
#pragma unroll
        for (int l = 0; l < 4; l++) {
#pragma unroll
          for (int j = 0; j < 8; j++) a[l * 8 + j] = LA((h * 4 + l) * 8 + j);
          Bm[(h * 4 + l) * 2] = LB((h * 4 + l) * 2), Bm[(h * 4 + l) * 2 + 1] = LB((h * 4 + l) * 2 + 1), bl[h * 4 + l] = Lb(h * 4 + l);
        }
#pragma unroll
        for (int l = 0; l < 4; l++) {
          const double pl = Prow[h * 4 + l];
#pragma unroll
          for (int j = 0; j < 8; j++) PAr[j] += pl * a[l * 8 + j];
          PBr[0] += pl * Bm[(h * 4 + l) * 2], PBr[1] += pl * Bm[(h * 4 + l) * 2 + 1];
          Pbi += pl * bl[h * 4 + l];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) L.PA[i * 8 + j][g] = PAr[j];
      L.PB[i * 2][g] = PBr[0], L.PB[i * 2 + 1][g] = PBr[1], L.Pb[i][g] = Pbi;
      WAVE_SYNC();
      // 3. (before 2: its operands are few, and the division that follows it can overlap the products of step 2)
      //    Huu, gu (same numbers in the 8 lanes of an instance)
      double Xm[16], PBm[16], Pbv[8];
#pragma unroll
      for (int l = 0; l < 8; l++) {
        Xm[l * 2] = L.Pxv[l * 2][g], Xm[l * 2 + 1] = L.Pxv[l * 2 + 1][g];
        PBm[l * 2] = L.PB[l * 2][g], PBm[l * 2 + 1] = L.PB[l * 2 + 1][g];
        Pbv[l] = L.Pb[l][g];
      }
      double Huu[4], gu[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int d = 0; d < 2; d++) {
          double s = Rm[sidx(c, d)] + Pvv[c * 2 + d];
#pragma unroll
          for (int l = 0; l < 8; l++)
            s += Bm[l * 2 + c] * PBm[l * 2 + d] + Bm[l * 2 + c] * Xm[l * 2 + d] + Xm[l * 2 + c] * Bm[l * 2 + d];
          Huu[c * 2 + d] = s;
        }
        Huu[c * 2 + c] += r2[c] + delta_w;
        double s = rr[c] + r2[c] * (uk[c] - vk[c]) + pv[c];
#pragma unroll
        for (int l = 0; l < 8; l++) s += Bm[l * 2 + c] * Pbv[l] + Xm[l * 2 + c] * bl[l];
        gu[c] = s;
      }
      // 2. row i of Hxx = Q + A^T (P A), of Hux^T, and gx_i
      double Hxx[8], Hxu[2], gx = qi;
#pragma unroll
      for (int j = 0; j < 8; j++) Hxx[j] = Qrow[j];
      Hxu[0] = Scol[0], Hxu[1] = Scol[1];
      {
        double Ai[8], PAi[8];
#pragma unroll
        for (int l = 0; l < 8; l++) Ai[l] = LA(l * 8 + i), PAi[l] = L.PA[l * 8 + i][g];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          double pa[32];
#pragma unroll
          for (int l = 0; l < 4; l++)
#pragma unroll
            for (int j = 0; j < 8; j++) pa[l * 8 + j] = L.PA[(h * 4 + l) * 8 + j][g];
#pragma unroll
          for (int l = 0; l < 4; l++) {
            const double ali = Ai[h * 4 + l];
#pragma unroll
            for (int j = 0; j < 8; j++) Hxx[j] += ali * pa[l * 8 + j];
          }
        }
#pragma unroll
        for (int l = 0; l < 8; l++) {
          const double ali = Ai[l], pali = PAi[l];
          Hxu[0] += Bm[l * 2] * pali + Xm[l * 2] * ali;
          Hxu[1] += Bm[l * 2 + 1] * pali + Xm[l * 2 + 1] * ali;
          gx += ali * Pbv[l];
        }
      }
      // (the next stage block is requested here: its 27 registers per lane are live for half a stage only)
      {
        const double* src = &PG(W.QP, i, kn, QP_NF);
#pragma unroll
        for (int j = 0; j < NPF; j++) pf[j] = src[j * 64];
      }
#undef LA
#undef LB
#undef Lb
      double det = Huu[0] * Huu[3] - Huu[1] * Huu[2];
      bool bad = !(Huu[0] > 0.0) || !(det > 1e-14 * Huu[0] * Huu[3]) || !isfinite(det);
      if (bad && live) ok = false;
      if (bad) det = 1.0, Huu[0] = Huu[3] = 1.0, Huu[1] = Huu[2] = 0.0;  // keep the lock-step arithmetic finite
      const double idet = 1.0 / det;  // one division per stage instead of four (same expression in the three Riccati kernels)
      double Hi[4] = {Huu[3] * idet, -Huu[1] * idet, -Huu[2] * idet, Huu[0] * idet};
      double Kc[2], Kv[4], kff[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        Kc[c] = -(Hi[c * 2 + 0] * Hxu[0] + Hi[c * 2 + 1] * Hxu[1]);  // K[c][i]
        Kv[c * 2 + 0] = Hi[c * 2 + 0] * r2[0], Kv[c * 2 + 1] = Hi[c * 2 + 1] * r2[1];
        kff[c] = -(Hi[c * 2 + 0] * gu[0] + Hi[c * 2 + 1] * gu[1]);
      }
      L.K[i][g] = Kc[0], L.K[8 + i][g] = Kc[1];
      WAVE_SYNC();
      // 4. cost-to-go of (x_k, v_k)
      double Pn[8];
#pragma unroll
      for (int j = 0; j < 8; j++) Pn[j] = Hxx[j] + Hxu[0] * L.K[j][g] + Hxu[1] * L.K[8 + j][g];
      pxv[0] = Hxu[0] * Kv[0] + Hxu[1] * Kv[2], pxv[1] = Hxu[0] * Kv[1] + Hxu[1] * Kv[3];
      ppi = gx + Hxu[0] * kff[0] + Hxu[1] * kff[1];
      double gv[2] = {-r2[0] * (uk[0] - vk[0]), -r2[1] * (uk[1] - vk[1])};
#pragma unroll
      for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int d = 0; d < 2; d++) Pvv[c * 2 + d] = ((c == d) ? r2[c] : 0.0) - r2[c] * Kv[c * 2 + d];
        pv[c] = gv[c] - r2[c] * kff[c];
      }
#pragma unroll
      for (int j = 0; j < 8; j++) L.PA[i * 8 + j][g] = Pn[j];
      WAVE_SYNC();
      {
        double Pt[8];  // column i of the new P: one batch of reads (as one conditional expression per element: eight basic blocks, each waiting for its own load)
#pragma unroll
        for (int j = 0; j < 8; j++) Pt[j] = L.PA[j * 8 + i][g];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(Pt[0]), "+v"(Pt[1]), "+v"(Pt[2]), "+v"(Pt[3]), "+v"(Pt[4]), "+v"(Pt[5]), "+v"(Pt[6]), "+v"(Pt[7]));
#endif
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const double sy = 0.5 * (Pn[j] + Pt[j]);
          Prow[j] = (j == i) ? Pn[j] : sy;
        }
      }
      L.Pxv[i * 2][g] = pxv[0], L.Pxv[i * 2 + 1][g] = pxv[1];  // read after the next stage's first barrier
      // the stage block fetched above goes to the other buffer (last read one stage ago), BEFORE this stage's stores
      // are issued: waiting for the loads then does not wait for the stores
#pragma unroll
      for (int j = 0; j < NPF; j++) L.sb[(k & 1) ^ 1][(i + 8 * j) * 8 + g] = pf[j];
      uk[0] = vk[0], uk[1] = vk[1];  // u_{k-1} is the v of stage k
      vk[0] = k > 1 ? vn0 : up0, vk[1] = k > 1 ? vn1 : up1;
      if (live) {
        PG(W.RC, RC_K + i, k, RC_NF) = Kc[0], PG(W.RC, RC_K + 8 + i, k, RC_NF) = Kc[1];
        // (selected, not indexed: a register array indexed by the lane's row lives in scratch memory, and a scratch reload
        //  waits for the global stores issued before it - vmcnt counts both)
        if (i < 4) PG(W.RC, RC_Kv + i, k, RC_NF) = i == 0 ? Kv[0] : (i == 1 ? Kv[1] : (i == 2 ? Kv[2] : Kv[3]));
        if (i < 2) PG(W.RC, RC_kff + i, k, RC_NF) = i == 0 ? kff[0] : kff[1];
        if (k > 0) {
#pragma unroll
          for (int j = 0; j < 8; j++)
            if (j <= i) PG(W.RC, RC_P + sidx(i, j), k, RC_NF) = Prow[j];
          PG(W.RC, RC_Pxv + i * 2, k, RC_NF) = pxv[0], PG(W.RC, RC_Pxv + i * 2 + 1, k, RC_NF) = pxv[1];
          PG(W.RC, RC_pp + i, k, RC_NF) = ppi;
        }
      }
    }
    // inertia correction schedule per instance (Waechter & Biegler 2006, Algorithm IC)
    const bool failed = live && !ok;
    if (failed) {
      if (delta_w == 0.0) delta_w = dw_last == 0.0 ? o.delta_w_first * psc : fmax(1e-20, dw_last / 3.0);
      else delta_w *= (dw_last == 0.0 ? 100.0 : 8.0);
      if (++tries > 40 || delta_w > 1e20) numerical = true;
      if (i == 0) STI(SI_NREG) += 1;
    }
    const int passes_used = failed ? STI(SI_SWEEPS) : 0;
    const bool again = failed && !numerical && sweep + 1 < max_sweeps && passes_used < o.max_iter;  // (a repeated sweep is a pass)
    if (again && i == 0) STI(SI_SWEEPS) = passes_used + 1;
    if (failed && !again) {  // continue in the next launch (or give up)
      if (i == 0) {
        STI(SI_STEP) = 0;
        if (numerical) STI(SI_STATUS) = LTOMPC_STATUS_NUMERICAL, STI(SI_DONE) = 1;
        else STI(SI_RETRY) = 1, STI(SI_TRIES) = tries, STD(ST_DW_TRY) = delta_w;
      }
      live = false;
    }
    if (!__any(again)) break;
  }
  if (live && i == 0) {
    STD(ST_DW_LAST) = delta_w > DW_KEEP * psc ? delta_w : 0.0;
    STD(ST_DW) = delta_w;
    STI(SI_RETRY) = 0, STI(SI_SKIP_EVAL) = 0;
    STI(SI_STEP) = 1;
  }
  if (!__any(live)) return;
  // ---- forward rollout: lane (g,i) carries dx_i; the full vector is gathered with wave shuffles.  A_k, B_k, b_k (88
  //      fields of the QP block) and the gains (22 fields of the RC block) are staged like the blocks of the sweep.
  double dxi = 0.0, dv[2] = {0.0, 0.0};
  if (live) PL(W.dX, i, 0, N + 1) = 0.0;
  constexpr int NFQ = 11, NFR = 3, FK = 88;  // fields i + 8 j: 11 per lane of A, B, b; 3 per lane of K, Kv, kff (stored at FK..)
  double fq[NFQ], fr[NFR];
  {
    const double* sq = &PG(W.QP, i, 0, QP_NF);
    const double* sr = &PG(W.RC, i, 0, RC_NF);  // (row 7 reads fields 7, 15, 23: the last one is P, not a gain, never used)
#pragma unroll
    for (int j = 0; j < NFQ; j++) fq[j] = sq[j * 64];
#pragma unroll
    for (int j = 0; j < NFR; j++) fr[j] = sr[j * 64];
  }
  WAVE_SYNC();
#pragma unroll
  for (int j = 0; j < NFQ; j++) L.sb[0][(i + 8 * j) * 8 + g] = fq[j];
#pragma unroll
  for (int j = 0; j < NFR; j++) L.sb[0][(FK + i + 8 * j) * 8 + g] = fr[j];  // (FK + 22, FK + 23: unused slots)
#pragma unroll 1
  for (int k = 0; k < N; k++) {
    const int kn = k + 1 < N ? k + 1 : k;
    {
      const double* sq = &PG(W.QP, i, kn, QP_NF);
      const double* sr = &PG(W.RC, i, kn, RC_NF);
#pragma unroll
      for (int j = 0; j < NFQ; j++) fq[j] = sq[j * 64];
#pragma unroll
      for (int j = 0; j < NFR; j++) fr[j] = sr[j * 64];
    }
    WAVE_SYNC();
    const double* q = L.sb[k & 1];
    double dx[8];
#pragma unroll
    for (int j = 0; j < 8; j++) dx[j] = __shfl(dxi, g + 8 * j);
    double du[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      double s = q[(FK + 20 + c) * 8 + g] + q[(FK + 16 + c * 2) * 8 + g] * dv[0] + q[(FK + 16 + c * 2 + 1) * 8 + g] * dv[1];
#pragma unroll
      for (int j = 0; j < 8; j++) s += q[(FK + c * 8 + j) * 8 + g] * dx[j];
      du[c] = s;
    }
    double s = q[(QP_b + i) * 8 + g] + q[(QP_B + i * 2) * 8 + g] * du[0] + q[(QP_B + i * 2 + 1) * 8 + g] * du[1];
#pragma unroll
    for (int j = 0; j < 8; j++) s += q[(QP_A + i * 8 + j) * 8 + g] * dx[j];
    dxi = s;
    dv[0] = du[0], dv[1] = du[1];
#pragma unroll
    for (int j = 0; j < NFQ; j++) L.sb[(k & 1) ^ 1][(i + 8 * j) * 8 + g] = fq[j];
#pragma unroll
    for (int j = 0; j < NFR; j++) L.sb[(k & 1) ^ 1][(FK + i + 8 * j) * 8 + g] = fr[j];
    if (live) {
      PL(W.dX, i, k + 1, N + 1) = dxi;
      if (i < 2) PL(W.dU, i, k, N) = du[i];
    }
  }
}

// ---- single-instance form of the sweep (k_riccati1): all 64 lanes work on ONE instance, lane (g, i) computes column g of
// row i of the 8x8 products instead of all 8 columns, with the SAME per-element expressions as d_riccati8, so that the
// bits do not depend on which of the two a solve goes through.  LDS slot [field][0] is shared by the 8 column lanes.
struct Stage1Regs {
  double a, bb, b, q_elem, S[2], q, R[3], r[2], u[2], v[2];
};
// Stage data of the ONE instance of a k_riccati1 block, staged in LDS once per launch: with a single wavefront per
// instance the sweep is a chain of N dependent stages, and fetching each stage from HBM/L2 (even one stage ahead) costs
// more than the stage's arithmetic.  q: [N][QP_NF] (copy of the instance's QP blocks), u: [N][2], kk: [N][22] gains.
// element i (run-time) of a register array: selects instead of an indexed (scratch) access
__device__ __forceinline__ double sel8(const double* a, const int i) {
  double r = a[0];
#pragma unroll
  for (int j = 1; j < 8; j++) r = (i == j) ? a[j] : r;
  return r;
}
struct StageLds {  // views into the dynamic LDS of a k_riccati1 block
  double* q;   // [N][QP_NF]
  double* u;   // [N][2]
  double* kk;  // [N][22]
};
struct Ric1Lds {
  double PA[64], PB[16], Pb[8], K[16], P[64], Pxv[16];
};
__host__ __device__ constexpr size_t ric1_lds_bytes(int N) { return sizeof(double) * (size_t)N * (QP_NF + 24) + sizeof(Ric1Lds); }
__device__ __forceinline__ void load_stage1(const StageLds& S, const double p0, const double p1, int i, int g, int k, double mu,
                                            double delta_w, Stage1Regs& s) {
  const int km = k > 0 ? k - 1 : 0;
  const double wn = k > 0 ? 1.0 : 0.0;
  const double* q = S.q + k * QP_NF;
  s.a = q[QP_A + i * 8 + g];
  s.bb = q[QP_B + i * 2 + (g & 1)];
  s.b = q[QP_b + i];
  const double qa = q[QP_Q + sidx(i, g)], qb = q[QP_Qx + sidx(i, g)];
  s.S[0] = q[QP_S + i], s.S[1] = q[QP_S + 8 + i];
  const double q0 = q[QP_q0 + i], q1 = q[QP_q1 + i];
  const double x0 = q[QP_qx0 + i], x1 = q[QP_qx1 + i];
  s.R[0] = q[QP_R + 0], s.R[1] = q[QP_R + 1], s.R[2] = q[QP_R + 2];
  const double r00 = q[QP_r0 + 0], r01 = q[QP_r0 + 1];
  const double r10 = q[QP_r1 + 0], r11 = q[QP_r1 + 1];
  s.u[0] = S.u[k * 2], s.u[1] = S.u[k * 2 + 1];
  const double v0 = S.u[km * 2], v1 = S.u[km * 2 + 1];
  s.q_elem = qa + ((i == g) ? delta_w : 0.0) + wn * qb;
  s.q = q0 + mu * q1 + wn * (x0 + mu * x1);
  s.r[0] = r00 + mu * r10, s.r[1] = r01 + mu * r11;
  s.v[0] = k > 0 ? v0 : p0, s.v[1] = k > 0 ? v1 : p1;
}

__device__ __forceinline__ void d_riccati1(const Consts& K, const Work& W, Ric1Lds& L, const StageLds& S, const int g, const int i, const int b,
                                           const bool valid, const int active_slot, const int max_sweeps) {
  const int N = W.N;
  double* st = W.st;
  int* si = W.si;
  const ltompc_options& o = K.o;
  // LTOMPC_DBG: shader-clock cycles of block 0 per section (head, staging, backward sweeps, forward), summed over launches
  const bool rprof = W.DBG != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  long long rt0 = rprof ? clock64() : 0;
#define RTOCK(q) if (rprof) { const long long t1 = clock64(); W.DBG[q] += (double)(t1 - rt0); rt0 = t1; }
  bool live, retry;
  double mu;
  if (!d_head8(K, W, i, b, valid, active_slot, live, retry, mu)) return;
  // ---- backward sweep (whole wave in lock-step; an instance whose Huu fails retries with a larger delta_w,
  //      the others recompute the same numbers)
  const double r2[2] = {2.0 * K.p.r_du[0], 2.0 * K.p.r_du[1]};
  const double psc = pen_scale(STD(ST_RHO));  // penalty scale (layout.h): the regularisation schedule in its units
  double delta_w = STD(ST_FORCE_REG);
  const double dw_last = STD(ST_DW_LAST);
  if (delta_w == 0.0 && dw_last > DW_KEEP * psc) delta_w = dw_last / 3.0;  // see DESIGN.md §3 (deviation from Algorithm IC)
  int tries = 0;
  if (retry) delta_w = STD(ST_DW_TRY), tries = STI(SI_TRIES);
  bool numerical = false;
  // max_sweeps = 1 while the launch is wide (a launch then never takes longer than one sweep, however hard the worst
  // instance of the batch is: its further attempts happen in the following launches); a few attempts per launch
  // once only the stragglers are left
  RTOCK(0);
  // stage the instance's QP blocks and inputs in LDS (all 64 lanes, independent loads)
  {
    const int lane = i * 8 + g;
    // four stage blocks per round: fields lane, lane + 64, lane + 128, lane + 192 of each (16 independent loads in flight
    // per lane; no index arithmetic beyond a stride - a flat index costs an integer division per word, and the staging was
    // bound by those: 24 k -> 18 k cycles)
    constexpr int FPL = (QP_NF + 63) / 64;
    for (int k0 = 0; k0 < N; k0 += 4) {
      double v[4 * FPL];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int kq = k0 + u < N ? k0 + u : N - 1;
        const double* src = &PG(W.QP, 0, kq, QP_NF);
#pragma unroll
        for (int j = 0; j < FPL; j++) {
          const int fq = lane + 64 * j;
          v[u * FPL + j] = src[(fq < QP_NF ? fq : QP_NF - 1) * 8];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int j = 0; j < FPL; j++)
          if (k0 + u < N && lane + 64 * j < QP_NF) S.q[(k0 + u) * QP_NF + lane + 64 * j] = v[u * FPL + j];
    }
    for (int idx = lane; idx < N * 2; idx += 64) S.u[idx] = PL(W.U, idx & 1, idx >> 1, N);
  }
  WAVE_SYNC();
  RTOCK(1);
  // (read once: a global load inside the stage loop would wait, on vmcnt, for the RC stores of the previous stage)
  const double up0 = W.uprev[b], up1 = W.uprev[(size_t)W.Bp + b];
  mu = __shfl(mu, 8 * i);  // the column lanes (g > 0) do real work here: give them the live lane's barrier parameter
  // lane constants of the stage: which element of Huu / gu the lane forms (c1, d1), and what it stores
  const int c1 = (g >> 1) & 1, d1 = g & 1;
  const double r2c = c1 ? r2[1] : r2[0], r2d = d1 ? r2[1] : r2[0];
  const int ofld = g == 0 ? RC_Pxv + i * 2 : (g == 1 ? RC_Pxv + i * 2 + 1 : (g == 2 ? RC_pp + i : (g == 3 ? RC_K + i : (g == 4 ? RC_K + 8 + i : (g == 5 ? RC_Kv + i : RC_kff + i)))));
  const int ofld_p = RC_P + sidx(i, g);
  const bool o_gain = g == 3 || g == 4 || (g == 5 && i < 4) || (g == 6 && i < 2);  // K, Kv, kff: every stage
  const bool o_cost = g < 3;                                                        // Pxv, p: stages k > 0
  for (int sweep = 0;; sweep++) {
    delta_w = __shfl(delta_w, 8 * i);  // ... and its regularisation
    bool ok = true;
    double Prow[8], pxv[2], ppi, Pvv[4] = {0, 0, 0, 0}, pv[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) Prow[j] = PG(W.QP, QP_Qx + sidx(i, j), N, QP_NF) + ((i == j) ? delta_w : 0.0);  // (terminal node: not staged)
    ppi = PG(W.QP, QP_qx0 + i, N, QP_NF) + mu * PG(W.QP, QP_qx1 + i, N, QP_NF);
    pxv[0] = pxv[1] = 0.0;
    if (live) {
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (j <= i) PG(W.RC, RC_P + sidx(i, j), N, RC_NF) = Prow[j];
      PG(W.RC, RC_Pxv + i * 2, N, RC_NF) = 0.0, PG(W.RC, RC_Pxv + i * 2 + 1, N, RC_NF) = 0.0;
      PG(W.RC, RC_pp + i, N, RC_NF) = ppi;
    }
    WAVE_SYNC();
    L.Pxv[i * 2] = 0.0, L.Pxv[i * 2 + 1] = 0.0;
    const bool live_w = __any(live);  // (one instance per wavefront: its column lanes store too)
#pragma unroll 1
    for (int k = N - 1; k >= 0; k--) {
      Stage1Regs cur;
      load_stage1(S, up0, up1, i, g, k, mu, delta_w, cur);
      const double Rm[3] = {cur.R[0], cur.R[1], cur.R[2]}, rr[2] = {cur.r[0], cur.r[1]};
      const double uk[2] = {cur.u[0], cur.u[1]}, vk[2] = {cur.v[0], cur.v[1]};
      const double* qk = S.q + k * QP_NF;  // A_k, B_k, b_k are read in place
      // 1. element (i, g) of P A, element (i, g < 2) of P B, P b + p (same expressions as d_riccati8, one column per lane).
      //    Every phase first pulls what it needs from LDS into registers, branch-free, and then computes: a wave waits
      //    once per phase instead of once per operand.
      double Ag[8], Bg[8], bl[8];
#pragma unroll
      for (int l = 0; l < 8; l++) Ag[l] = qk[QP_A + l * 8 + g], Bg[l] = qk[QP_B + l * 2 + (g & 1)], bl[l] = qk[QP_b + l];
      double pa = 0.0, pb = 0.0, Pbi = ppi;
#pragma unroll
      for (int l = 0; l < 8; l++) {
        pa += Prow[l] * Ag[l];
        pb += Prow[l] * Bg[l];  // (lanes g >= 2 repeat column g & 1 and drop it)
        Pbi += Prow[l] * bl[l];
      }
      L.PA[i * 8 + g] = pa;
      if (g < 2) L.PB[i * 2 + g] = pb;
      L.Pb[i] = Pbi;
      WAVE_SYNC();
      // 2. element (i, g) of Hxx, row i of Hux^T, gx_i; the operands of the lane's element of Huu / gu (step 3) are
      //    fetched here as well, with the lane's own column offsets (c1, d1) - selecting them from B0 / B1 / X0 / X1
      //    costs two v_cndmask per double, 130 VALU instructions per stage
      //    Two batches of four rows (l): all eight at once are 104 doubles in flight, and the stage's loop invariants move to
      //    the accumulation registers and back (37 v_accvgpr_read per stage).
      double hxx = cur.q_elem, Hxu[2] = {cur.S[0], cur.S[1]}, gx = cur.q;
      // 3. Huu, gu: one element per lane (g = 0..3: Huu[g>>1][g&1], g = 4, 5: gu[g-4]), gathered with wave shuffles.
      //    Both sums are formed by every lane (no divergent branches, no run-time indices into register arrays: those
      //    would live in scratch, and a scratch reload waits for the RC stores of the stage before); Bg = column d1 of B
      const double rm = (c1 && d1) ? Rm[2] : ((c1 || d1) ? Rm[1] : Rm[0]);     // Rm[sidx(c, d)]
      const double pvv = c1 ? (d1 ? Pvv[3] : Pvv[2]) : (d1 ? Pvv[1] : Pvv[0]);  // Pvv[c * 2 + d]
      double s = rm + pvv;
      double sg = (d1 ? rr[1] : rr[0]) + r2d * ((d1 ? uk[1] : uk[0]) - (d1 ? vk[1] : vk[0])) + (d1 ? pv[1] : pv[0]);  // gu[c], c = g & 1
#pragma unroll
      for (int h = 0; h < 2; h++) {
        double Ai[4], PAg[4], PAi[4], B0[4], B1[4], X0[4], X1[4], Pbv[4], Bc[4], Xc[4], Xd[4], PBd[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int l = h * 4 + q;
          Ai[q] = qk[QP_A + l * 8 + i], PAg[q] = L.PA[l * 8 + g], PAi[q] = L.PA[l * 8 + i];
          B0[q] = qk[QP_B + l * 2], B1[q] = qk[QP_B + l * 2 + 1], X0[q] = L.Pxv[l * 2], X1[q] = L.Pxv[l * 2 + 1];
          Pbv[q] = L.Pb[l];
          Bc[q] = qk[QP_B + l * 2 + c1], Xc[q] = L.Pxv[l * 2 + c1], Xd[q] = L.Pxv[l * 2 + d1], PBd[q] = L.PB[l * 2 + d1];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          double ali = Ai[q];
          hxx += ali * PAg[q];
          double pali = PAi[q];
          Hxu[0] += B0[q] * pali + X0[q] * ali;
          Hxu[1] += B1[q] * pali + X1[q] * ali;
          gx += ali * Pbv[q];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) s += Bc[q] * PBd[q] + Bc[q] * Xd[q] + Xc[q] * Bg[h * 4 + q];
#pragma unroll
        for (int q = 0; q < 4; q++) sg += Bg[h * 4 + q] * Pbv[q] + Xd[q] * bl[h * 4 + q];
      }
      double heH = s;
      if (c1 == d1) heH += r2c + delta_w;
      const double he = g < 4 ? heH : (g < 6 ? sg : 0.0);
      double Huu[4], gu[2];
#pragma unroll
      for (int q = 0; q < 4; q++) Huu[q] = __shfl(he, q + 8 * i);
      gu[0] = __shfl(he, 4 + 8 * i), gu[1] = __shfl(he, 5 + 8 * i);
      double det = Huu[0] * Huu[3] - Huu[1] * Huu[2];
      bool bad = !(Huu[0] > 0.0) || !(det > 1e-14 * Huu[0] * Huu[3]) || !isfinite(det);
      if (bad && live) ok = false;
      if (bad) det = 1.0, Huu[0] = Huu[3] = 1.0, Huu[1] = Huu[2] = 0.0;
      const double idet = 1.0 / det;  // one division per stage instead of four (same expression in the three Riccati kernels)
      double Hi[4] = {Huu[3] * idet, -Huu[1] * idet, -Huu[2] * idet, Huu[0] * idet};
      double Kc[2], Kv[4], kff[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        Kc[c] = -(Hi[c * 2 + 0] * Hxu[0] + Hi[c * 2 + 1] * Hxu[1]);  // K[c][i]
        Kv[c * 2 + 0] = Hi[c * 2 + 0] * r2[0], Kv[c * 2 + 1] = Hi[c * 2 + 1] * r2[1];
        kff[c] = -(Hi[c * 2 + 0] * gu[0] + Hi[c * 2 + 1] * gu[1]);
      }
      L.K[i] = Kc[0], L.K[8 + i] = Kc[1];
      WAVE_SYNC();
      // 4. cost-to-go: element (i, g)
      const double pn = hxx + Hxu[0] * L.K[g] + Hxu[1] * L.K[8 + g];
      pxv[0] = Hxu[0] * Kv[0] + Hxu[1] * Kv[2], pxv[1] = Hxu[0] * Kv[1] + Hxu[1] * Kv[3];
      ppi = gx + Hxu[0] * kff[0] + Hxu[1] * kff[1];
      double gv[2] = {-r2[0] * (uk[0] - vk[0]), -r2[1] * (uk[1] - vk[1])};
#pragma unroll
      for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int d = 0; d < 2; d++) Pvv[c * 2 + d] = ((c == d) ? r2[c] : 0.0) - r2[c] * Kv[c * 2 + d];
        pv[c] = gv[c] - r2[c] * kff[c];
      }
      L.P[i * 8 + g] = pn;
      WAVE_SYNC();
      // row i of the symmetrised P for the next stage: row and column fetched in one batch, then combined (written as
      // one conditional expression per element the compiler made eight basic blocks of it, each waiting for its own loads)
      double Pr[8], Pc[8];
#pragma unroll
      for (int j = 0; j < 8; j++) Pr[j] = L.P[i * 8 + j], Pc[j] = L.P[j * 8 + i];
      double Pt = L.P[g * 8 + i];  // transposed partner of the lane's own element
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(Pr[0]), "+v"(Pr[1]), "+v"(Pr[2]), "+v"(Pr[3]), "+v"(Pr[4]), "+v"(Pr[5]), "+v"(Pr[6]), "+v"(Pr[7]));
      asm volatile("" : "+v"(Pc[0]), "+v"(Pc[1]), "+v"(Pc[2]), "+v"(Pc[3]), "+v"(Pc[4]), "+v"(Pc[5]), "+v"(Pc[6]), "+v"(Pc[7]), "+v"(Pt));
#endif
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const double sy = 0.5 * (Pr[j] + Pc[j]);
        Prow[j] = (j == i) ? Pr[j] : sy;
      }
      L.Pxv[i * 2] = pxv[0], L.Pxv[i * 2 + 1] = pxv[1];
      // Stores: one element per lane.  Lane (g, i) holds element (i, g) of P (same expression as Prow[g] of row i);
      // the remaining words of the stage go out through the lanes g = 0..6 of every row (all column lanes of a row hold the
      // same K, Kv, kff, Pxv, p): two store instructions per stage instead of fourteen conditional ones
      const double sy_own = 0.5 * (pn + Pt);
      const double psym = (g == i) ? pn : sy_own;
      const double Kvi = i == 0 ? Kv[0] : (i == 1 ? Kv[1] : (i == 2 ? Kv[2] : Kv[3]));
      const double kfi = i == 0 ? kff[0] : kff[1];
      const double oval = g == 0 ? pxv[0] : (g == 1 ? pxv[1] : (g == 2 ? ppi : (g == 3 ? Kc[0] : (g == 4 ? Kc[1] : (g == 5 ? Kvi : kfi)))));
      if (o_gain) S.kk[k * 22 + ofld] = oval;  // the gains stay in LDS for the forward rollout (kk offsets = RC fields)
      if (live_w) {
        if (o_gain || (o_cost && k > 0)) PG(W.RC, ofld, k, RC_NF) = oval;
        if (g <= i && k > 0) PG(W.RC, ofld_p, k, RC_NF) = psym;
      }
      if (__any(!ok)) break;  // (one instance per wavefront) the sweep has failed - its remaining stages would be thrown away
    }
    // inertia correction schedule per instance (Waechter & Biegler 2006, Algorithm IC)
    const bool failed = live && !ok;
    if (failed) {
      if (delta_w == 0.0) delta_w = dw_last == 0.0 ? o.delta_w_first * psc : fmax(1e-20, dw_last / 3.0);
      else delta_w *= (dw_last == 0.0 ? 100.0 : 8.0);
      if (++tries > 40 || delta_w > 1e20) numerical = true;
      if (i == 0) STI(SI_NREG) += 1;
    }
    const int passes_used = failed ? STI(SI_SWEEPS) : 0;
    const bool again = failed && !numerical && sweep + 1 < max_sweeps && passes_used < o.max_iter;  // (a repeated sweep is a pass)
    if (again && i == 0) STI(SI_SWEEPS) = passes_used + 1;
    if (failed && !again) {  // continue in the next launch (or give up)
      if (i == 0) {
        STI(SI_STEP) = 0;
        if (numerical) STI(SI_STATUS) = LTOMPC_STATUS_NUMERICAL, STI(SI_DONE) = 1;
        else STI(SI_RETRY) = 1, STI(SI_TRIES) = tries, STD(ST_DW_TRY) = delta_w;
      }
      live = false;
    }
    if (!__any(again)) break;
  }
  RTOCK(2);
  if (live && i == 0) {
    STD(ST_DW_LAST) = delta_w > DW_KEEP * psc ? delta_w : 0.0;
    STD(ST_DW) = delta_w;
    STI(SI_RETRY) = 0, STI(SI_SKIP_EVAL) = 0;
    STI(SI_STEP) = 1;
  }
  if (!__any(live)) return;
  // ---- forward rollout: lane (g,i) carries dx_i; the full vector is gathered with wave shuffles
  double dxi = 0.0, dv[2] = {0.0, 0.0};
  if (live) PL(W.dX, i, 0, N + 1) = 0.0;
  WAVE_SYNC();
#pragma unroll 1
  for (int k = 0; k < N; k++) {
    FwdRegs fc;
#pragma unroll
    for (int j = 0; j < 16; j++) fc.K[j] = S.kk[k * 22 + j];
#pragma unroll
    for (int j = 0; j < 4; j++) fc.Kv[j] = S.kk[k * 22 + 16 + j];
    fc.kff[0] = S.kk[k * 22 + 20], fc.kff[1] = S.kk[k * 22 + 21];
#pragma unroll
    for (int j = 0; j < 8; j++) fc.A[j] = S.q[k * QP_NF + QP_A + i * 8 + j];
    fc.B[0] = S.q[k * QP_NF + QP_B + i * 2], fc.B[1] = S.q[k * QP_NF + QP_B + i * 2 + 1];
    fc.b = S.q[k * QP_NF + QP_b + i];
    double dx[8];
#pragma unroll
    for (int j = 0; j < 8; j++) dx[j] = __shfl(dxi, g + 8 * j);
    double du[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      double s = fc.kff[c] + fc.Kv[c * 2] * dv[0] + fc.Kv[c * 2 + 1] * dv[1];
#pragma unroll
      for (int j = 0; j < 8; j++) s += fc.K[c * 8 + j] * dx[j];
      du[c] = s;
    }
    double s = fc.b + fc.B[0] * du[0] + fc.B[1] * du[1];
#pragma unroll
    for (int j = 0; j < 8; j++) s += fc.A[j] * dx[j];
    dxi = s;
    dv[0] = du[0], dv[1] = du[1];
    if (live) {
      PL(W.dX, i, k + 1, N + 1) = dxi;
      if (i < 2) PL(W.dU, i, k, N) = du[i];
    }
  }
  RTOCK(3);
  if (rprof) W.DBG[4] += 1.0;
#undef RTOCK
}

// ---- four-wavefront form of the single-instance sweep (k_riccati1q).  In d_riccati1 every lane forms, beside its own element of
// the 8x8 products, the row's Hxu / gx and an element of Huu / gu: ~100 fp64 operations and ~130 LDS reads per lane and stage
// where its own element needs 18, and with one wavefront on the SIMD the stage is as long as its instruction count (~3600 cycles).
// Here the ROLES run beside each other on four wavefronts (one per SIMD of the CU), exchanging through LDS at three workgroup
// barriers per stage:
//   phase A   w0 lane (i, g): P A;                  w1 lanes (i, c): P B, lanes 16 + i: P b + p
//   phase B   w0: Hxx element;                      w1: Hxu(c, i), gx(i);      w2 lanes (c, d): Huu;      w3 lanes d: gu
//   phase C   w0: new P element (own K from Hxu of row g);  w1: K, Pxv, p;     w2: Kv, Pvv;               w3: kff, pv
// Every value is formed by ONE lane with the statement d_riccati1 / d_riccati8 use for it (same operands, same order), so the
// Riccati buffer is the same, word for word (scratch/rc_cmp.py, tests: the kernel paths agree bit for bit).
struct Ric1qLds {
  double PA[64], PB[16], Pb[8], Hxu[16], gx[8], Huu[4], gu[2], Pn[64], Pxv[16], pp[8], Pvv[4], pv[2];
  double mu, delta_w;
  int go, live, again, pad_;
};
__host__ __device__ constexpr size_t ric1q_lds_bytes(int N) { return sizeof(double) * (size_t)N * (QP_NF + 24) + sizeof(Ric1qLds); }

// Workgroup barrier for data exchanged through LDS only: waits for the wavefront's own LDS operations (lgkmcnt), NOT for its
// outstanding global stores - __syncthreads() also drains vmcnt, and every barrier of a stage would then wait for the Riccati
// words the stage has just stored to HBM (measured: a stage of k_riccati1q 3500 cycles with __syncthreads()).
#define WG_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#if !defined(LTOMPC_HOST_HARNESS)
__device__ __forceinline__ void d_riccati1q(const Consts& K, const Work& W, Ric1qLds& X, const StageLds& S, const int t, const int b,
                                            const int active_slot, const int max_sweeps) {
  const int N = W.N;
  double* st = W.st;
  int* si = W.si;
  const ltompc_options& o = K.o;
  const int w = t >> 6, lane = t & 63, i = lane >> 3, g = lane & 7;
  const bool rprof = W.DBG != nullptr && blockIdx.x == 0 && t == 0;
  long long rt0 = rprof ? clock64() : 0;
#define RTOCK(q) if (rprof) { const long long t1 = clock64(); W.DBG[q] += (double)(t1 - rt0); rt0 = t1; }
  // ---- head (wavefront 0) beside the staging of the instance's stage blocks in LDS (wavefronts 1 - 3)
  bool live = false, retry = false;  // (wavefront 0, lanes g == 0)
  double mu = 0.0;
  double psc = 1.0, dw_last = 0.0;  // (wavefront 0: read after the head, which may change the instance's penalty)
  int tries = 0;
  bool numerical = false;
  if (w == 0) {
    const bool go = d_head8(K, W, i, b, g == 0, active_slot, live, retry, mu);
    psc = pen_scale(STD(ST_RHO));  // penalty scale (layout.h): the regularisation schedule in its units
    dw_last = STD(ST_DW_LAST);
    double delta_w = STD(ST_FORCE_REG);
    if (delta_w == 0.0 && dw_last > DW_KEEP * psc) delta_w = dw_last / 3.0;  // see DESIGN.md §3 (deviation from Algorithm IC)
    if (retry) delta_w = STD(ST_DW_TRY), tries = STI(SI_TRIES);
    if (t == 0) X.go = go ? 1 : 0, X.live = live ? 1 : 0, X.mu = mu, X.delta_w = delta_w;
  } else {
    const int ln = t - 64;  // 0 .. 191: fields ln and ln + 192 of a stage block, eight stage blocks per round (16 independent loads)
    for (int k0 = 0; k0 < N; k0 += 8) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int kq = k0 + u < N ? k0 + u : N - 1;
        const double* src = &PG(W.QP, 0, kq, QP_NF);
        v[u * 2] = src[ln * 8];
        v[u * 2 + 1] = src[(ln + 192 < QP_NF ? ln + 192 : QP_NF - 1) * 8];
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (k0 + u < N) {
          S.q[(k0 + u) * QP_NF + ln] = v[u * 2];
          if (ln + 192 < QP_NF) S.q[(k0 + u) * QP_NF + ln + 192] = v[u * 2 + 1];
        }
    }
    for (int idx = ln; idx < N * 2; idx += 192) S.u[idx] = PL(W.U, idx & 1, idx >> 1, N);
  }
  WG_SYNC_LDS();
  RTOCK(0);
  if (!X.go) return;
  const bool live_w = X.live != 0;  // the instance has a sweep to do (head8 returns true only then: one instance per workgroup)
  mu = X.mu;
  const double up0 = W.uprev[b], up1 = W.uprev[(size_t)W.Bp + b];
  const double r2[2] = {2.0 * K.p.r_du[0], 2.0 * K.p.r_du[1]};
  // roles of wavefront 1: lanes 0 .. 15 = (row, input) pairs, lanes 16 .. 23 = rows
  const int r1 = lane < 16 ? (lane >> 1) : (lane - 16), c1w = lane & 1;
  const int row = w == 0 ? i : (r1 & 7);
  RTOCK(1);
  for (int sweep = 0;; sweep++) {
    const double delta_w = X.delta_w;
    bool ok = true;
    // terminal node: P_N (as a full symmetric matrix in Pn: the row builder below then returns it unchanged), p_N, Pxv_N = 0
    if (w == 0) {
      const double pnn = PG(W.QP, QP_Qx + sidx(i, g), N, QP_NF) + ((i == g) ? delta_w : 0.0);
      X.Pn[i * 8 + g] = pnn;
      if (live_w && g <= i) PG(W.RC, RC_P + sidx(i, g), N, RC_NF) = pnn;
    } else if (w == 1) {
      if (lane < 16) {
        X.Pxv[lane] = 0.0;
        if (live_w) PG(W.RC, RC_Pxv + lane, N, RC_NF) = 0.0;
      } else if (lane < 24) {
        const double ppn = PG(W.QP, QP_qx0 + r1, N, QP_NF) + mu * PG(W.QP, QP_qx1 + r1, N, QP_NF);
        X.pp[r1] = ppn;
        if (live_w) PG(W.RC, RC_pp + r1, N, RC_NF) = ppn;
      }
    } else if (w == 2) {
      if (lane < 4) X.Pvv[lane] = 0.0;
    } else {
      if (lane < 2) X.pv[lane] = 0.0;
    }
    WG_SYNC_LDS();
    double pn_prev = 0.0;
#pragma unroll 1
    for (int k = N - 1; k >= 0; k--) {
      const double* qk = S.q + k * QP_NF;
      const int km = k > 0 ? k - 1 : 0;
      const double wn = k > 0 ? 1.0 : 0.0;
      // ---- phase A: row `row` of the symmetrised P of stage k + 1, then P A / P B / P b + p
      double hxx = 0.0;
      if (w == 0 || (w == 1 && lane < 24)) {
        double Pr[8], Pc[8], Prow[8];
#pragma unroll
        for (int j = 0; j < 8; j++) Pr[j] = X.Pn[row * 8 + j], Pc[j] = X.Pn[j * 8 + row];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const double sy = 0.5 * (Pr[j] + Pc[j]);
          Prow[j] = (j == row) ? Pr[j] : sy;
        }
        if (w == 0) {
          if (live_w && g <= i && k + 1 < N) {  // the element of stage k + 1 this lane formed in its phase C (d_riccati1: psym)
            const double Pt = X.Pn[g * 8 + i];
            const double sy_own = 0.5 * (pn_prev + Pt);
            PG(W.RC, RC_P + sidx(i, g), k + 1, RC_NF) = (g == i) ? pn_prev : sy_own;
          }
          double Ag[8];
#pragma unroll
          for (int l = 0; l < 8; l++) Ag[l] = qk[QP_A + l * 8 + g];
          double pa = 0.0;
#pragma unroll
          for (int l = 0; l < 8; l++) pa += Prow[l] * Ag[l];
          X.PA[i * 8 + g] = pa;
          const double qa = qk[QP_Q + sidx(i, g)], qb = qk[QP_Qx + sidx(i, g)];
          hxx = qa + ((i == g) ? delta_w : 0.0) + wn * qb;
        } else if (lane < 16) {
          double Bg[8];
#pragma unroll
          for (int l = 0; l < 8; l++) Bg[l] = qk[QP_B + l * 2 + c1w];
          double pb = 0.0;
#pragma unroll
          for (int l = 0; l < 8; l++) pb += Prow[l] * Bg[l];
          X.PB[r1 * 2 + c1w] = pb;
        } else {
          double bl[8];
#pragma unroll
          for (int l = 0; l < 8; l++) bl[l] = qk[QP_b + l];
          double Pbi = X.pp[r1];
#pragma unroll
          for (int l = 0; l < 8; l++) Pbi += Prow[l] * bl[l];
          X.Pb[r1] = Pbi;
        }
      }
      WG_SYNC_LDS();
      // ---- phase B: Hxx, Hxu, gx, Huu, gu - one element per lane
      if (w == 0) {
        double Ai[8], PAg[8];
#pragma unroll
        for (int l = 0; l < 8; l++) Ai[l] = qk[QP_A + l * 8 + i], PAg[l] = X.PA[l * 8 + g];
#pragma unroll
        for (int l = 0; l < 8; l++) {
          double ali = Ai[l];
          hxx += ali * PAg[l];
        }
      } else if (w == 1) {
        if (lane < 16) {
          double Ai[8], PAi[8], Bc[8], Xc[8];
#pragma unroll
          for (int l = 0; l < 8; l++) Ai[l] = qk[QP_A + l * 8 + r1], PAi[l] = X.PA[l * 8 + r1], Bc[l] = qk[QP_B + l * 2 + c1w], Xc[l] = X.Pxv[l * 2 + c1w];
          double h = qk[QP_S + c1w * 8 + r1];
#pragma unroll
          for (int l = 0; l < 8; l++) {
            double ali = Ai[l];
            double pali = PAi[l];
            h += Bc[l] * pali + Xc[l] * ali;
          }
          X.Hxu[c1w * 8 + r1] = h;
        } else if (lane < 24) {
          double Ai[8], Pbv[8];
#pragma unroll
          for (int l = 0; l < 8; l++) Ai[l] = qk[QP_A + l * 8 + r1], Pbv[l] = X.Pb[l];
          const double q0 = qk[QP_q0 + r1], q1 = qk[QP_q1 + r1], x0 = qk[QP_qx0 + r1], x1 = qk[QP_qx1 + r1];
          double gx = q0 + mu * q1 + wn * (x0 + mu * x1);
#pragma unroll
          for (int l = 0; l < 8; l++) {
            double ali = Ai[l];
            gx += ali * Pbv[l];
          }
          X.gx[r1] = gx;
        }
      } else if (w == 2) {
        if (lane < 4) {
          const int c1 = lane >> 1, d1 = lane & 1;
          double Bc[8], PBd[8], Xd[8], Xc[8], Bg[8];
#pragma unroll
          for (int l = 0; l < 8; l++)
            Bc[l] = qk[QP_B + l * 2 + c1], PBd[l] = X.PB[l * 2 + d1], Xd[l] = X.Pxv[l * 2 + d1], Xc[l] = X.Pxv[l * 2 + c1], Bg[l] = qk[QP_B + l * 2 + d1];
          const double Rm[3] = {qk[QP_R + 0], qk[QP_R + 1], qk[QP_R + 2]};
          const double rm = (c1 && d1) ? Rm[2] : ((c1 || d1) ? Rm[1] : Rm[0]);  // Rm[sidx(c, d)]
          const double pvv = X.Pvv[c1 * 2 + d1];
          double s = rm + pvv;
#pragma unroll
          for (int l = 0; l < 8; l++) s += Bc[l] * PBd[l] + Bc[l] * Xd[l] + Xc[l] * Bg[l];
          const double r2c = c1 ? r2[1] : r2[0];
          double heH = s;
          if (c1 == d1) heH += r2c + delta_w;
          X.Huu[lane] = heH;
        }
      } else {
        if (lane < 2) {
          const int d1 = lane;
          double Bg[8], Pbv[8], Xd[8], bl[8];
#pragma unroll
          for (int l = 0; l < 8; l++) Bg[l] = qk[QP_B + l * 2 + d1], Pbv[l] = X.Pb[l], Xd[l] = X.Pxv[l * 2 + d1], bl[l] = qk[QP_b + l];
          const double rr = qk[QP_r0 + d1] + mu * qk[QP_r1 + d1];
          const double uk = S.u[k * 2 + d1], vk = k > 0 ? S.u[km * 2 + d1] : (d1 ? up1 : up0);
          const double r2d = d1 ? r2[1] : r2[0];
          double sg = rr + r2d * (uk - vk) + X.pv[d1];
#pragma unroll
          for (int l = 0; l < 8; l++) sg += Bg[l] * Pbv[l] + Xd[l] * bl[l];
          X.gu[d1] = sg;
        }
      }
      WG_SYNC_LDS();
      // ---- phase C: gains and cost-to-go
      {
        double Huu[4] = {X.Huu[0], X.Huu[1], X.Huu[2], X.Huu[3]}, gu[2] = {X.gu[0], X.gu[1]};
        double det = Huu[0] * Huu[3] - Huu[1] * Huu[2];
        const bool bad = !(Huu[0] > 0.0) || !(det > 1e-14 * Huu[0] * Huu[3]) || !isfinite(det);
        if (bad) ok = false;
        if (bad) det = 1.0, Huu[0] = Huu[3] = 1.0, Huu[1] = Huu[2] = 0.0;
        const double idet = 1.0 / det;  // one division per stage instead of four (same expression in the Riccati kernels)
        const double Hi[4] = {Huu[3] * idet, -Huu[1] * idet, -Huu[2] * idet, Huu[0] * idet};
        if (w == 0) {
          const double Hxi[2] = {X.Hxu[i], X.Hxu[8 + i]}, Hxu[2] = {X.Hxu[g], X.Hxu[8 + g]};
          double Kc[2];  // K[c][g]: the statement of the row-g lanes of d_riccati1
#pragma unroll
          for (int c = 0; c < 2; c++) Kc[c] = -(Hi[c * 2 + 0] * Hxu[0] + Hi[c * 2 + 1] * Hxu[1]);
          const double pn = hxx + Hxi[0] * Kc[0] + Hxi[1] * Kc[1];
          X.Pn[i * 8 + g] = pn;
          pn_prev = pn;
        } else if (w == 1) {
          if (lane < 24) {
            const double Hxu[2] = {X.Hxu[r1], X.Hxu[8 + r1]};
            double Kc[2], Kv[4], kff[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
              Kc[c] = -(Hi[c * 2 + 0] * Hxu[0] + Hi[c * 2 + 1] * Hxu[1]);  // K[c][i]
              Kv[c * 2 + 0] = Hi[c * 2 + 0] * r2[0], Kv[c * 2 + 1] = Hi[c * 2 + 1] * r2[1];
              kff[c] = -(Hi[c * 2 + 0] * gu[0] + Hi[c * 2 + 1] * gu[1]);
            }
            if (lane < 16) {
              double pxv[2];
              pxv[0] = Hxu[0] * Kv[0] + Hxu[1] * Kv[2], pxv[1] = Hxu[0] * Kv[1] + Hxu[1] * Kv[3];
              const double pxv_own = c1w ? pxv[1] : pxv[0], K_own = c1w ? Kc[1] : Kc[0];
              X.Pxv[r1 * 2 + c1w] = pxv_own;
              S.kk[k * 22 + RC_K + c1w * 8 + r1] = K_own;  // the gains stay in LDS for the forward rollout (kk offsets = RC fields)
              if (live_w) {
                PG(W.RC, RC_K + c1w * 8 + r1, k, RC_NF) = K_own;
                if (k > 0) PG(W.RC, RC_Pxv + r1 * 2 + c1w, k, RC_NF) = pxv_own;
              }
            } else {
              const double ppi = X.gx[r1] + Hxu[0] * kff[0] + Hxu[1] * kff[1];
              X.pp[r1] = ppi;
              if (live_w && k > 0) PG(W.RC, RC_pp + r1, k, RC_NF) = ppi;
            }
          }
        } else if (w == 2) {
          if (lane < 4) {
            const int c = lane >> 1, d = lane & 1;
            double Kv[4];
#pragma unroll
            for (int cc = 0; cc < 2; cc++) Kv[cc * 2 + 0] = Hi[cc * 2 + 0] * r2[0], Kv[cc * 2 + 1] = Hi[cc * 2 + 1] * r2[1];
            const double Kv_own = lane == 0 ? Kv[0] : (lane == 1 ? Kv[1] : (lane == 2 ? Kv[2] : Kv[3]));
            const double r2c = c ? r2[1] : r2[0];
            X.Pvv[lane] = ((c == d) ? r2c : 0.0) - r2c * Kv_own;
            S.kk[k * 22 + RC_Kv + lane] = Kv_own;
            if (live_w) PG(W.RC, RC_Kv + lane, k, RC_NF) = Kv_own;
          }
        } else {
          if (lane < 2) {
            const int c = lane;
            double kff[2];
#pragma unroll
            for (int cc = 0; cc < 2; cc++) kff[cc] = -(Hi[cc * 2 + 0] * gu[0] + Hi[cc * 2 + 1] * gu[1]);
            const double kf = c ? kff[1] : kff[0];
            const double uk = S.u[k * 2 + c], vk = k > 0 ? S.u[km * 2 + c] : (c ? up1 : up0);
            const double r2c = c ? r2[1] : r2[0];
            const double gv = -r2c * (uk - vk);
            X.pv[c] = gv - r2c * kf;
            S.kk[k * 22 + RC_kff + c] = kf;
            if (live_w) PG(W.RC, RC_kff + c, k, RC_NF) = kf;
          }
        }
      }
      WG_SYNC_LDS();
      if (!ok) break;  // (uniform: every thread tests the same Huu) the sweep has failed - its remaining stages would be thrown away
    }
    // inertia correction schedule (Waechter & Biegler 2006, Algorithm IC): thread 0 keeps the instance's books
    if (t == 0) {
      double dw = delta_w;
      const bool failed = live && !ok;
      if (failed) {
        if (dw == 0.0) dw = dw_last == 0.0 ? o.delta_w_first * psc : fmax(1e-20, dw_last / 3.0);
        else dw *= (dw_last == 0.0 ? 100.0 : 8.0);
        if (++tries > 40 || dw > 1e20) numerical = true;
        STI(SI_NREG) += 1;
      }
      const int passes_used = failed ? STI(SI_SWEEPS) : 0;
      const bool again = failed && !numerical && sweep + 1 < max_sweeps && passes_used < o.max_iter;  // (a repeated sweep is a pass)
      if (again) STI(SI_SWEEPS) = passes_used + 1;
      if (failed && !again) {  // continue in the next launch (or give up)
        STI(SI_STEP) = 0;
        if (numerical) STI(SI_STATUS) = LTOMPC_STATUS_NUMERICAL, STI(SI_DONE) = 1;
        else STI(SI_RETRY) = 1, STI(SI_TRIES) = tries, STD(ST_DW_TRY) = dw;
        live = false;
      }
      X.delta_w = dw, X.again = again ? 1 : 0, X.live = live ? 1 : 0;
    }
    WG_SYNC_LDS();
    if (!X.again) break;
  }
  RTOCK(2);
  if (w != 0) return;
  const bool live_f = X.live != 0;  // the sweep succeeded: the step is rolled out
  if (t == 0 && live_f) {
    const double delta_w = X.delta_w;
    STD(ST_DW_LAST) = delta_w > DW_KEEP * psc ? delta_w : 0.0;
    STD(ST_DW) = delta_w;
    STI(SI_RETRY) = 0, STI(SI_SKIP_EVAL) = 0;
    STI(SI_STEP) = 1;
  }
  if (!live_f) return;
  // ---- forward rollout (wavefront 0, as in d_riccati1): lane (g, i) carries dx_i; the full vector is gathered with wave shuffles
  const bool lane_live = g == 0;
  double dxi = 0.0, dv[2] = {0.0, 0.0};
  if (lane_live) PL(W.dX, i, 0, N + 1) = 0.0;
#pragma unroll 1
  for (int k = 0; k < N; k++) {
    FwdRegs fc;
#pragma unroll
    for (int j = 0; j < 16; j++) fc.K[j] = S.kk[k * 22 + j];
#pragma unroll
    for (int j = 0; j < 4; j++) fc.Kv[j] = S.kk[k * 22 + 16 + j];
    fc.kff[0] = S.kk[k * 22 + 20], fc.kff[1] = S.kk[k * 22 + 21];
#pragma unroll
    for (int j = 0; j < 8; j++) fc.A[j] = S.q[k * QP_NF + QP_A + i * 8 + j];
    fc.B[0] = S.q[k * QP_NF + QP_B + i * 2], fc.B[1] = S.q[k * QP_NF + QP_B + i * 2 + 1];
    fc.b = S.q[k * QP_NF + QP_b + i];
    double dx[8];
#pragma unroll
    for (int j = 0; j < 8; j++) dx[j] = __shfl(dxi, g + 8 * j);
    double du[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      double s = fc.kff[c] + fc.Kv[c * 2] * dv[0] + fc.Kv[c * 2 + 1] * dv[1];
#pragma unroll
      for (int j = 0; j < 8; j++) s += fc.K[c * 8 + j] * dx[j];
      du[c] = s;
    }
    double s = fc.b + fc.B[0] * du[0] + fc.B[1] * du[1];
#pragma unroll
    for (int j = 0; j < 8; j++) s += fc.A[j] * dx[j];
    dxi = s;
    dv[0] = du[0], dv[1] = du[1];
    if (lane_live) {
      PL(W.dX, i, k + 1, N + 1) = dxi;
      if (i < 2) PL(W.dU, i, k, N) = du[i];
    }
  }
  RTOCK(3);
  if (rprof) W.DBG[4] += 1.0;
#undef RTOCK
}
#endif

__global__ void __launch_bounds__(64) k_riccati8(Consts K, Work W, Launch la, int it_index, int max_sweeps) {
  __shared__ RicLds L;
  const int lane = threadIdx.x, g = lane & 7, i = lane >> 3;
  const int jj = blockIdx.x * 8 + g;
  const bool valid = jj < la.nact[0];
  d_riccati8(K, W, L, g, i, la.act[valid ? jj : 0], valid, it_index, max_sweeps);
}

// Four wavefronts per instance (see d_riccati1q): dynamic LDS ric1q_lds_bytes(N).
__global__ void __launch_bounds__(256) k_riccati1q(Consts K, Work W, Launch la, int it_index, int max_sweeps) {
#if defined(LTOMPC_HOST_HARNESS)
  (void)K, (void)W, (void)la, (void)it_index, (void)max_sweeps;  // (never run by the harness)
#else
  extern __shared__ double lds1q[];
  if ((int)blockIdx.x >= la.nact[0]) return;
  const int N = W.N;
  StageLds S{lds1q, lds1q + (size_t)N * QP_NF, lds1q + (size_t)N * (QP_NF + 2)};
  Ric1qLds& X = *reinterpret_cast<Ric1qLds*>(lds1q + (size_t)N * (QP_NF + 24));
  d_riccati1q(K, W, X, S, threadIdx.x, la.act[blockIdx.x], it_index, max_sweeps);
#endif
}

// One wavefront per instance (narrow launches: once few instances are left, a launch is as long as one wavefront's
// sweep, and 8 instances per wavefront make that sweep ~3x longer than it has to be).  Dynamic LDS: ric1_lds_bytes(N).
__global__ void __launch_bounds__(64) k_riccati1(Consts K, Work W, Launch la, int it_index, int max_sweeps) {
#if defined(LTOMPC_HOST_HARNESS)
  static double lds1[1];  // (never run by the harness)
#else
  extern __shared__ double lds1[];
#endif
  const int lane = threadIdx.x, g = lane & 7, i = lane >> 3;
  if ((int)blockIdx.x >= la.nact[0]) return;
  const int N = W.N;
  StageLds S{lds1, lds1 + (size_t)N * QP_NF, lds1 + (size_t)N * (QP_NF + 2)};
  Ric1Lds& L = *reinterpret_cast<Ric1Lds*>(lds1 + (size_t)N * (QP_NF + 24));
  d_riccati1(K, W, L, S, g, i, la.act[blockIdx.x], g == 0, it_index, max_sweeps);
}


}  // namespace ltompc
