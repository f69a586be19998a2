// linearise.h — one (interval, instance) slot per thread: linearisation, elimination of the collocation point,
// stage-QP assembly (k_eval) and expansion of the Riccati step (k_expand).
#pragma once
#include "layout.h"

namespace ltompc {

// ------------------------------------------------------------------------------------------ small dense LA
__device__ __forceinline__ double sym_get(const double* H, int i, int j) { return H[sidx(i, j)]; }


// Which simple bounds exist, as a type.  BoundsAny reads it from the parameters at run time (one uniform branch per
// possible bound: every bound becomes its own basic block, and the loads of its slack and multiplier cannot be moved
// out of it - measured on k_eval: 46 serialised HBM round trips per wavefront).  BoundsFixed states it at compile time:
// straight-line code, the loads are issued in batches.  BoundsRef is the pattern of the reference's controller
// (controller.py:79-95: both input bounds; s >= 0, |mu| <= pi/2, vx >= 0, |delta| <= pi/4, |T| <= 1); a handle whose
// parameters have exactly this pattern runs the kernels instantiated for it (same arithmetic, same bits).
struct BoundsAny {
  static constexpr bool fixed = false;
  static constexpr unsigned ulb = 0, uub = 0, xlb = 0, xub = 0;
};
template <unsigned ULB, unsigned UUB, unsigned XLB, unsigned XUB>
struct BoundsFixed {
  static constexpr bool fixed = true;
  static constexpr unsigned ulb = ULB, uub = UUB, xlb = XLB, xub = XUB;  // bit i: variable i has the bound
};
using BoundsRef = BoundsFixed<0x3, 0x3, 0xCD, 0xC4>;

// Visits the inequalities of a slot in their storage order (input bounds, Radau-point bounds, node bounds; per
// variable lower then upper, only the bounds that are set).  `f(m, kind, i, sg, val)` gets the running index m,
// kind 0/1/2 = u / c / x+, and the variable index i as a value that is a compile-time constant after unrolling,
// so that per-variable arrays stay in registers (a run-time index would force them into scratch memory).
template <class BP, typename F>
__device__ __forceinline__ int for_each_bound(const ltompc_params& p, F&& f) {
  int m = 0;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    if (BP::fixed ? ((BP::ulb >> i) & 1u) != 0 : p.u_lb[i] > -LTOMPC_NO_BOUND) f(m++, 0, i, -1.0, p.u_lb[i]);
    if (BP::fixed ? ((BP::uub >> i) & 1u) != 0 : p.u_ub[i] < LTOMPC_NO_BOUND) f(m++, 0, i, 1.0, p.u_ub[i]);
  }
#pragma unroll
  for (int kind = 1; kind <= 2; kind++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (BP::fixed ? ((BP::xlb >> i) & 1u) != 0 : p.x_lb[i] > -LTOMPC_NO_BOUND) f(m++, kind, i, -1.0, p.x_lb[i]);
      if (BP::fixed ? ((BP::xub >> i) & 1u) != 0 : p.x_ub[i] < LTOMPC_NO_BOUND) f(m++, kind, i, 1.0, p.x_ub[i]);
    }
  }
  return m;  // index of the first track constraint
}

// ------------------------------------------------------------------------------------------ track constraints
// Barrier terms of one track constraint g <= 0 with slack t and multiplier nu: Hessian weight Sg on grad g grad g^T,
// gradient multiplier s0 + mu s1.  Hard (rho = 0): g + t = 0.  Soft (options.soft_rho = rho > 0): g - e + t = 0 with
// the elastic variable e >= 0 that costs rho e; its bound multiplier is z = rho - nu (stationarity in e is linear, so
// z needs no storage), and eliminating (dt, de, dnu) from the Newton system
//   grad g . dx - de + dt = -(g - e + t) ,  t dnu + nu dt = mu - t nu ,  -e dnu + z de = mu - e z
// gives dnu = Sg (grad g . dx + g + mu / nu - mu / z) with Sg = 1 / (t / nu + e / z).
__device__ __forceinline__ void track_barrier(const double rho, const double gv, const double t, const double nu,
                                              const double e, double& Sg, double& s0, double& s1) {
  if (rho > 0.0) {
    const double inu = 1.0 / nu, iz = 1.0 / (rho - nu);
    Sg = 1.0 / (t * inu + e * iz), s0 = nu + Sg * gv, s1 = Sg * (inu - iz);
  } else {
    const double it = 1.0 / t;
    Sg = nu * it, s0 = nu * (gv + t) * it, s1 = it;
  }
}
// steps of (t, nu, e) of a soft track constraint from gd = grad g . dx (see track_barrier), the bounds they put on the
// step lengths (r_pri = max(-dt / t, -de / e); 0 < nu + a dnu < rho) and their part of the directional derivative of
// the barrier objective
__device__ __forceinline__ void track_soft_step(const double rho, const double mu, const double tau, const double gv,
                                                const double gd, const double t, const double nu, const double e,
                                                double& dtt, double& dn, double& dee, double& r_pri, double& a_dua,
                                                double& gphid) {
  const double z = rho - nu, inu = 1.0 / nu, iz = 1.0 / z, it = 1.0 / t, ie = 1.0 / e;
  const double Sg = 1.0 / (t * inu + e * iz);
  dn = Sg * (gd + gv + mu * inu - mu * iz);
  dtt = mu * inu - t - t * inu * dn;
  dee = mu * iz - e + e * iz * dn;
  r_pri = fmax(r_pri, fmax(-dtt * it, -dee * ie));
  if (dn < 0.0) a_dua = fmin(a_dua, -tau * nu / dn);
  if (dn > 0.0) a_dua = fmin(a_dua, tau * z / dn);
  gphid += rho * dee - mu * dtt * it - mu * dee * ie;
}

// Slacks, inequality multipliers (and elastic variables) of slot k from its primal point (k_init at the start of a
// solve, reinit_slot when the restoration phase is entered): t = max(-h, bound_push), nu = mu / t.
template <class BP>
__device__ __forceinline__ void init_slot_slacks(const Consts& K, const Work& W, const int k, const int b, const double mu,
                                                 const double eps, const double rho, const double* xp, const double* c,
                                                 const double* u) {
  const int N = W.N;
  const int m = for_each_bound<BP>(K.p, [&](int mm, int kind, int j, double sg, double val) {
    const double xv = kind == 0 ? u[j] : (kind == 1 ? c[j] : xp[j]);
    const double hv = sg * (xv - val);
    const double t = -hv > K.o.bound_push ? -hv : K.o.bound_push;
    PL(W.T, mm, k, N) = t, PL(W.NU, mm, k, N) = mu / t;
  });
  double gv[3] = {-1.0, -1.0, -1.0};
  if (k + 1 <= N - 1) cons_eval(K.p, K.T, eps, xp, gv, nullptr, nullptr, nullptr, nullptr, nullptr);
#pragma unroll
  for (int q = 0; q < 3; q++) {
    double t = -gv[q] > K.o.bound_push ? -gv[q] : K.o.bound_push, e = 0.0;
    if (rho > 0.0) {
      // softened: slack and multiplier as for the hard constraint (a violated constraint starts as an infeasibility of
      // g - e + t = 0 that the Newton steps remove, not as a large elastic variable with nu ~ rho that the barrier lets
      // go of only slowly); the elastic variable on the central path of its own pair, e (rho - nu) = mu.
      // t >= 2 mu / rho keeps nu <= rho / 2.
      t = fmax(t, 2.0 * mu / rho);
      e = mu / (rho - mu / t);
      if (gv[q] > ELASTIC_CP_VIOL) {
        // grossly violated (here the Newton steps would have to shrink t by rho / nu ~ 1e4 at 1 % of a step per
        // iteration): on the central path of the elastic pair instead, e - t = g, t nu = mu, e (rho - nu) = mu
        const double g = gv[q], bq = rho * g - 2.0 * mu;
        t = (-bq + sqrt(bq * bq + 4.0 * rho * mu * g)) / (2.0 * rho);
        e = g + t;
      }
    }
    PL(W.T, m + q, k, N) = t, PL(W.T, K.bd.ni + q, k, N) = e, PL(W.NU, m + q, k, N) = mu / t;  // (elastic planes follow the ni slacks)
  }
  if (K.bd.nel) {  // friction-ellipse constraints: always soft, penalty params.ell_penalty, same start as a softened track constraint
    const double pen = K.p.ell_penalty;
    double ge[2] = {-1.0, -1.0};
    if (k + 1 <= N - 1) ellipse_val(K.p, xp, ge);
#pragma unroll
    for (int q = 0; q < 2; q++) {
      double t = fmax(-ge[q] > K.o.bound_push ? -ge[q] : K.o.bound_push, 2.0 * mu / pen), e = mu / (pen - mu / t);
      if (ge[q] > ELASTIC_CP_VIOL) {
        const double g = ge[q], bq = pen * g - 2.0 * mu;
        t = (-bq + sqrt(bq * bq + 4.0 * pen * mu * g)) / (2.0 * pen);
        e = g + t;
      }
      PL(W.T, m + 3 + q, k, N) = t, PL(W.T, K.bd.ni + 3 + q, k, N) = e, PL(W.NU, m + 3 + q, k, N) = mu / t;
    }
  }
}
// Restoration entry (SI_REINIT, set by d_pick): the slot's slacks, multipliers and elastic variables start again from its
// primal point with the instance's new barrier parameter and penalty, the collocation multipliers at zero.  Everything
// written here belongs to slot k alone (the one neighbour's access, L1 / L2 of slot k + 1 in the dual residual of d_eval,
// is replaced by zeros while the flag is set).
template <class BP>
__device__ __forceinline__ void reinit_slot(const Consts& K, const Work& W, const int k, const int b) {
  const int N = W.N;
  const double mu = W.st[(size_t)ST_MU * W.Bp + b], eps = W.st[(size_t)ST_EPS * W.Bp + b], rho = W.st[(size_t)ST_RHO * W.Bp + b];
  double xp[8], c[8], u[2];
#pragma unroll
  for (int i = 0; i < 8; i++) xp[i] = PL(W.X, i, k + 1, N + 1), c[i] = PL(W.C, i, k, N);
  u[0] = PL(W.U, 0, k, N), u[1] = PL(W.U, 1, k, N);
  init_slot_slacks<BP>(K, W, k, b, mu, eps, rho, xp, c, u);
#pragma unroll
  for (int i = 0; i < 8; i++) PL(W.L1, i, k, N) = 0.0, PL(W.L2, i, k, N) = 0.0;
}

// ------------------------------------------------------------------------------------------ slot linearisation
// Slot k owns (u_k, c_k, x_{k+1}) and the collocation equations of interval k in do_mpc's Radau-IIA(2) form
//   G1 = h f(c,u) + 2 x_k - 1.5 c - 0.5 x+ = 0 ,  G2 = h f(x+,u) - 2 x_k + 4.5 c - 2.5 x+ = 0   
struct Slot {
  double xk[8], xp[8], c[8], u[2];
  double E1[64], E2[64], G1[8], G2[8];
  double Hc[36], gc0[8], gc1[8];     // QP block of c_k : gradient = gc0 + mu gc1 (barrier terms included)
  double Hxp[36], gxp0[8], gxp1[8];  // QP block of x_{k+1}
  double Du[2], gub0[2], gub1[2];    // input-bound barrier
  double dcd[8], dxd[8], dud[2];     // parts of grad_z L that do not involve the collocation multipliers
  double gcost[8];
  double gs[3], gn[3], gm[3];        // gradients of gL, gR+, gR-
  double gv[3];                      // values of gL, gR+, gR- at x_{k+1}
  double gve[2], ge[2][5];           // ELL: values and gradients (over states 3..7) of the two friction-ellipse constraints
  double rp_ineq, cmax, cmin, smult; // WITH_DUAL: max |h + t|, max / min t nu, sum |nu| over the slot's inequalities
  double th_ineq, sumlog;            // WITH_DUAL: sum |h + t|, sum log t (filter measures of the current point)
  double emax;                       // WITH_DUAL: largest elastic variable of the slot (0 on hard constraints)
  double cost;
  int m_nl;                          // storage index of gL
  bool nl;
};

template <bool WITH_DUAL, class BP, bool ELL>
__device__ __forceinline__ void linearise_slot(const Consts& K, const Work& W, int k, int b, double eps, Slot& S) {
  const int N = W.N;
  const double hdt = K.o.t_step, rho = W.st[(size_t)ST_RHO * W.Bp + b];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    S.xk[i] = k == 0 ? W.x0[(size_t)i * W.Bp + b] : PL(W.X, i, k, N + 1);
    S.xp[i] = PL(W.X, i, k + 1, N + 1);
    S.c[i] = PL(W.C, i, k, N);
  }
  S.u[0] = PL(W.U, 0, k, N), S.u[1] = PL(W.U, 1, k, N);
  double l1[8], l2[8];
#pragma unroll
  for (int i = 0; i < 8; i++) l1[i] = PL(W.L1, i, k, N), l2[i] = PL(W.L2, i, k, N);
#pragma unroll
  for (int i = 0; i < 36; i++) S.Hc[i] = 0.0, S.Hxp[i] = 0.0;
  double f1[8], f2[8], J[48];
  rhs_derivs(K.p, K.T, eps, S.c, f1, J, l1, hdt, S.Hc);
#pragma unroll
  for (int i = 0; i < 64; i++) S.E1[i] = 0.0, S.E2[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 48; i++) S.E1[i] = hdt * J[i];
  rhs_derivs(K.p, K.T, eps, S.xp, f2, J, l2, hdt, S.Hxp);
#pragma unroll
  for (int i = 0; i < 48; i++) S.E2[i] = hdt * J[i];
  f1[6] = f2[6] = S.u[0], f1[7] = f2[7] = S.u[1];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    S.E1[i * 8 + i] -= 1.5, S.E2[i * 8 + i] -= 2.5;
    S.G1[i] = hdt * f1[i] + 2.0 * S.xk[i] - 1.5 * S.c[i] - 0.5 * S.xp[i];
    S.G2[i] = hdt * f2[i] - 2.0 * S.xk[i] + 4.5 * S.c[i] - 2.5 * S.xp[i];
  }
#pragma unroll
  for (int i = 0; i < 8; i++) S.gcost[i] = 0.0, S.gc0[i] = 0.0, S.gc1[i] = 0.0, S.gxp1[i] = 0.0, S.dcd[i] = 0.0;
  S.cost = cost_eval(K.p, K.T, eps, S.xp, k == N - 1, S.gcost, S.Hxp);
#pragma unroll
  for (int i = 0; i < 8; i++) S.gxp0[i] = S.gcost[i], S.dxd[i] = S.gcost[i];
  S.Du[0] = S.Du[1] = 0.0, S.gub0[0] = S.gub0[1] = 0.0, S.gub1[0] = S.gub1[1] = 0.0, S.dud[0] = S.dud[1] = 0.0;
  // inequalities: u bounds, c bounds, x+ bounds, nl constraints.  Barrier: Sigma = nu/t on the Hessian,
  // sigma = (mu + nu (h + t))/t = nu (h+t)/t + mu (1/t) on the gradient.
  S.rp_ineq = 0.0, S.cmax = 0.0, S.cmin = 1e300, S.smult = 0.0, S.th_ineq = 0.0, S.sumlog = 0.0, S.emax = 0.0;
  double lprod = 1.0;  // sum of log t = log of products of 8 slacks (3 logarithms per slot, see d_linesearch)
  const int m_nl = for_each_bound<BP>(K.p, [&](int m, int kind, int j, double sg, double val) {
    const double xv = kind == 0 ? S.u[j] : (kind == 1 ? S.c[j] : S.xp[j]);
    const double hv = sg * (xv - val);
    const double t = PL(W.T, m, k, N), nu = PL(W.NU, m, k, N), it = 1.0 / t;
    const double Sg = nu * it, g0 = sg * nu * (hv + t) * it, g1 = sg * it;
    if (kind == 0) {
      S.Du[j] += Sg, S.gub0[j] += g0, S.gub1[j] += g1;
      if (WITH_DUAL) S.dud[j] += sg * nu;
    } else if (kind == 1) {
      S.Hc[sidx(j, j)] += Sg, S.gc0[j] += g0, S.gc1[j] += g1;
      if (WITH_DUAL) S.dcd[j] += sg * nu;
    } else {
      S.Hxp[sidx(j, j)] += Sg, S.gxp0[j] += g0, S.gxp1[j] += g1;
      if (WITH_DUAL) S.dxd[j] += sg * nu;
    }
    if (WITH_DUAL) {
      S.rp_ineq = fmax(S.rp_ineq, fabs(hv + t));
      S.cmax = fmax(S.cmax, t * nu), S.cmin = fmin(S.cmin, t * nu), S.smult += fabs(nu);
      S.th_ineq += fabs(hv + t), lprod *= t;
      if ((m & 7) == 7) S.sumlog += log(lprod), lprod = 1.0;
    }
  });
  S.m_nl = m_nl;
  S.nl = (k + 1 <= N - 1);  // nl_cons are checked at nodes 1..N-1 (node 0 is data, node N is not checked)
  if (S.nl) {
    // slacks / multipliers (/ elastic variables, stored after the slacks) of the three track constraints: fetched
    // together, before the constraint evaluation with its two table look-ups
    double tq[3], nq[3], eq[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      tq[q] = PL(W.T, m_nl + q, k, N), nq[q] = PL(W.NU, m_nl + q, k, N);
      eq[q] = rho > 0.0 ? PL(W.T, K.bd.ni + q, k, N) : 0.0;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(nq[0]), "+v"(nq[1]), "+v"(nq[2]), "+v"(eq[0]), "+v"(eq[1]), "+v"(eq[2]));
#endif
    double hss[3], hmm[3];
    cons_eval(K.p, K.T, eps, S.xp, S.gv, S.gs, S.gn, S.gm, hss, hmm);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      int mm = m_nl + q;
      double t = tq[q], nu = nq[q];
      const double e = eq[q];
      double Sg, s0, s1;
      track_barrier(rho, S.gv[q], t, nu, e, Sg, s0, s1);
      double g3[3] = {S.gs[q], S.gn[q], S.gm[q]};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        S.gxp0[a] += s0 * g3[a], S.gxp1[a] += s1 * g3[a];
        if (WITH_DUAL) S.dxd[a] += nu * g3[a];
#pragma unroll
        for (int c = 0; c <= a; c++) S.Hxp[sidx(a, c)] += Sg * g3[a] * g3[c];
      }
      S.Hxp[sidx(0, 0)] += nu * hss[q];
      S.Hxp[sidx(2, 2)] += nu * hmm[q];
      if (WITH_DUAL) {
        S.rp_ineq = fmax(S.rp_ineq, fabs(S.gv[q] - e + t));
        S.cmax = fmax(S.cmax, t * nu), S.cmin = fmin(S.cmin, t * nu), S.smult += fabs(nu);
        S.th_ineq += fabs(S.gv[q] - e + t), lprod *= t;
        if (rho > 0.0) {
          const double ez = e * (rho - nu);
          S.cmax = fmax(S.cmax, ez), S.cmin = fmin(S.cmin, ez), S.smult += fabs(rho - nu);
          lprod *= e, S.cost += rho * e, S.emax = fmax(S.emax, e);
        }
        if ((mm & 7) == 7) S.sumlog += log(lprod), lprod = 1.0;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 3; q++) S.gv[q] = -1.0, S.gs[q] = S.gn[q] = S.gm[q] = 0.0;
  }
  if (ELL && S.nl) {
    // friction-ellipse constraints (always soft, penalty params.ell_penalty): node block over the states 3..7
    const double pen = K.p.ell_penalty;
    double te[2], ne[2], ee[2];
#pragma unroll
    for (int q = 0; q < 2; q++) te[q] = PL(W.T, m_nl + 3 + q, k, N), ne[q] = PL(W.NU, m_nl + 3 + q, k, N), ee[q] = PL(W.T, K.bd.ni + 3 + q, k, N);
    double h15[2][15];
    ellipse_eval(K.p, S.xp, S.gve, S.ge, h15);
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int mm = m_nl + 3 + q;
      const double t = te[q], nu = ne[q], e = ee[q];
      double Sg, s0, s1;
      track_barrier(pen, S.gve[q], t, nu, e, Sg, s0, s1);
#pragma unroll
      for (int a = 0; a < 5; a++) {
        S.gxp0[3 + a] += s0 * S.ge[q][a], S.gxp1[3 + a] += s1 * S.ge[q][a];
        if (WITH_DUAL) S.dxd[3 + a] += nu * S.ge[q][a];
#pragma unroll
        for (int c = 0; c <= a; c++) S.Hxp[sidx(3 + a, 3 + c)] += Sg * S.ge[q][a] * S.ge[q][c] + nu * h15[q][sidx(a, c)];
      }
      if (WITH_DUAL) {
        const double ez = e * (pen - nu);
        S.rp_ineq = fmax(S.rp_ineq, fabs(S.gve[q] - e + t));
        S.cmax = fmax(S.cmax, fmax(t * nu, ez)), S.cmin = fmin(S.cmin, fmin(t * nu, ez)), S.smult += fabs(nu) + fabs(pen - nu);
        S.th_ineq += fabs(S.gve[q] - e + t), lprod *= t, lprod *= e, S.cost += pen * e;
        if ((mm & 7) == 7) S.sumlog += log(lprod), lprod = 1.0;
      }
    }
  }
  if (WITH_DUAL) S.sumlog += log(lprod);
}

// Elimination of the collocation point: with M8 = 4.5 I + 2 E2 E1,
//   M8 dc = (2I - 4E2) dx - (I + 2E2) Bu du - G2 - 2 E2 G1 ,   dx+ = 2 (E1 dc + 2 dx + Bu du + G1)
// Y = [Ac | Bc | bc] (8 x 11), AB = [A | B | b] (8 x 11).
//
// Structure.  With the states grouped a = (s, n, mu), b = (vx, vy, r), c = (delta, T): the kinematic rows of the
// model do not depend on c, the dynamic rows do not depend on a, and the rows of c are d/dt = u.  Hence E1, E2 are
// block UPPER triangular in (a, b, c) with E_ac = 0 and E_cc = -1.5 I / -2.5 I, so are M8 (with M_cc = 12 I), the
// first 8 columns of Y and A (with A_cc = I): the elimination is two 3x3 inverses and block back-substitutions, and
// every product below runs over the structurally non-zero range only (less than half of the dense flops).
__host__ __device__ constexpr int gs_(int i) { return i < 3 ? 0 : (i < 6 ? 3 : 6); }  // first index of i's group
__host__ __device__ constexpr int ge_(int i) { return i < 3 ? 2 : (i < 6 ? 5 : 7); }  // last index of i's group
// row i of E1 / E2 is non-zero in columns elo_(i) .. ehi_(i)
__host__ __device__ constexpr int elo_(int i) { return i < 6 ? gs_(i) : i; }
__host__ __device__ constexpr int ehi_(int i) { return i < 3 ? 5 : (i < 6 ? 7 : i); }
// column col of Y = [Ac | Bc | bc] is non-zero in rows 0 .. yrow_(col)
__host__ __device__ constexpr int yrow_(int col) { return col < 8 ? ge_(col) : 7; }
// Hessian of the Lagrangian of the collocation equations in c_k: second derivatives of the kinematic rows over
// (s, n, mu, vx, vy), of the dynamic rows over (vx, vy, r, delta), and the diagonal barrier terms of the bounds
__host__ __device__ constexpr bool hnz_(int i, int j) { return (i <= 4 && j <= 4) || (i >= 3 && i <= 6 && j >= 3 && j <= 6) || i == j; }
__host__ __device__ constexpr int imin_(int x, int y) { return x < y ? x : y; }
__host__ __device__ constexpr int imax_(int x, int y) { return x > y ? x : y; }

struct M8Blocks {  // M8 = [[Maa Mab Mac], [0 Mbb Mbc], [0 0 12 I]]
  double iaa[9], ibb[9];  // inverses of the diagonal blocks
  double ab[9], ac[6], bc[6];
};

__device__ __forceinline__ bool inv33(const double* m, double* r) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const double id = 1.0 / det;
  r[0] = c00 * id, r[1] = (m[2] * m[7] - m[1] * m[8]) * id, r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  r[3] = c01 * id, r[4] = (m[0] * m[8] - m[2] * m[6]) * id, r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r[6] = c02 * id, r[7] = (m[1] * m[6] - m[0] * m[7]) * id, r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return fabs(det) > 1e-12;
}

// y = M8^-1 v for a right-hand side whose rows > RMAX are structurally zero (those of y are then zero too, not written)
template <int RMAX>
__device__ __forceinline__ void m8_solve(const M8Blocks& M, const double* v, double* y) {
  double yc[2] = {0.0, 0.0}, yb[3] = {0.0, 0.0, 0.0};
  if (RMAX >= 6) {
    yc[0] = v[6] * (1.0 / 12.0), yc[1] = v[7] * (1.0 / 12.0);
    y[6] = yc[0], y[7] = yc[1];
  }
  if (RMAX >= 3) {
    double rb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      rb[i] = v[3 + i];
      if (RMAX >= 6) rb[i] -= M.bc[i * 2] * yc[0] + M.bc[i * 2 + 1] * yc[1];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) yb[i] = M.ibb[i * 3] * rb[0] + M.ibb[i * 3 + 1] * rb[1] + M.ibb[i * 3 + 2] * rb[2], y[3 + i] = yb[i];
  }
  double ra[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    ra[i] = v[i];
    if (RMAX >= 3) ra[i] -= M.ab[i * 3] * yb[0] + M.ab[i * 3 + 1] * yb[1] + M.ab[i * 3 + 2] * yb[2];
    if (RMAX >= 6) ra[i] -= M.ac[i * 2] * yc[0] + M.ac[i * 2 + 1] * yc[1];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) y[i] = M.iaa[i * 3] * ra[0] + M.iaa[i * 3 + 1] * ra[1] + M.iaa[i * 3 + 2] * ra[2];
}
// x = M8^-T v (dense v): forward substitution through the transposed blocks
__device__ __forceinline__ void m8_solve_t(const M8Blocks& M, const double* v, double* x) {
#pragma unroll
  for (int i = 0; i < 3; i++) x[i] = M.iaa[i] * v[0] + M.iaa[3 + i] * v[1] + M.iaa[6 + i] * v[2];
  double rb[3];
#pragma unroll
  for (int i = 0; i < 3; i++) rb[i] = v[3 + i] - (M.ab[i] * x[0] + M.ab[3 + i] * x[1] + M.ab[6 + i] * x[2]);
#pragma unroll
  for (int i = 0; i < 3; i++) x[3 + i] = M.ibb[i] * rb[0] + M.ibb[3 + i] * rb[1] + M.ibb[6 + i] * rb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    double s = v[6 + i];
#pragma unroll
    for (int l = 0; l < 3; l++) s -= M.ac[l * 2 + i] * x[l] + M.bc[l * 2 + i] * x[3 + l];
    x[6 + i] = s * (1.0 / 12.0);
  }
}

template <int COL>
__device__ __forceinline__ void condense_column(const Consts& K, const Slot& S, const M8Blocks& M, double* Y, double* AB) {
  const double hdt = K.o.t_step;
  constexpr int RM = yrow_(COL);
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, y[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (COL < 8) {
#pragma unroll
    for (int i = 0; i <= RM; i++) v[i] = ((i == COL) ? 2.0 : 0.0) - ((COL >= elo_(i) && COL <= ehi_(i)) ? 4.0 * S.E2[i * 8 + COL] : 0.0);
  } else if (COL < 10) {
    constexpr int j = 6 + COL - 8;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = -hdt * (((i == j) ? 1.0 : 0.0) + ((j >= elo_(i) && j <= ehi_(i)) ? 2.0 * S.E2[i * 8 + j] : 0.0));
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double s = -S.G2[i];
#pragma unroll
      for (int l = elo_(i); l <= ehi_(i); l++) s -= 2.0 * S.E2[i * 8 + l] * S.G1[l];
      v[i] = s;
    }
  }
  m8_solve<RM>(M, v, y);
#pragma unroll
  for (int i = 0; i <= RM; i++) Y[i * 11 + COL] = y[i];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double s;
    if (COL < 8) s = (i == COL) ? 2.0 : 0.0;
    else if (COL < 10) s = (i == 6 + COL - 8) ? hdt : 0.0;
    else s = S.G1[i];
#pragma unroll
    for (int l = elo_(i); l <= imin_(ehi_(i), RM); l++) s += S.E1[i * 8 + l] * y[l];
    AB[i * 11 + COL] = 2.0 * s;
  }
}

// M8 = 4.5 I + 2 E2 E1 in block form, diagonal blocks inverted
__device__ __forceinline__ bool factor_m8(const Slot& S, M8Blocks& M) {
  // M8(i, j) = 4.5 delta_ij + 2 sum_l E2(i, l) E1(l, j): l runs where row i of E2 and column j of E1 overlap
  double maa[9], mbb[9];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = gs_(i); j < 8; j++) {
      double s = (i == j) ? 4.5 : 0.0;
#pragma unroll
      for (int l = elo_(i); l <= imin_(ehi_(i), ge_(j)); l++)
        if (j >= elo_(l) && j <= ehi_(l)) s += 2.0 * S.E2[i * 8 + l] * S.E1[l * 8 + j];
      if (i < 3) {
        if (j < 3) maa[i * 3 + j] = s;
        else if (j < 6) M.ab[i * 3 + j - 3] = s;
        else M.ac[i * 2 + j - 6] = s;
      } else {
        if (j < 6) mbb[(i - 3) * 3 + j - 3] = s;
        else M.bc[(i - 3) * 2 + j - 6] = s;
      }
    }
  return inv33(maa, M.iaa) & inv33(mbb, M.ibb);
}

__device__ __forceinline__ bool condense_slot(const Consts& K, const Slot& S, M8Blocks& M, double* Y, double* AB) {
  const bool ok = factor_m8(S, M);
#pragma unroll
  for (int q = 0; q < 88; q++) Y[q] = 0.0, AB[q] = 0.0;
  condense_column<0>(K, S, M, Y, AB), condense_column<1>(K, S, M, Y, AB), condense_column<2>(K, S, M, Y, AB);
  condense_column<3>(K, S, M, Y, AB), condense_column<4>(K, S, M, Y, AB), condense_column<5>(K, S, M, Y, AB);
  condense_column<6>(K, S, M, Y, AB), condense_column<7>(K, S, M, Y, AB), condense_column<8>(K, S, M, Y, AB);
  condense_column<9>(K, S, M, Y, AB), condense_column<10>(K, S, M, Y, AB);
  return ok;
}


// ------------------------------------------------------------------------------------------ k_eval
template <class BP, bool ELL>
__device__ __forceinline__ void d_eval(const Consts& K, const Work& W, const int k, const int b, const bool force) {
  const int N = W.N;
  if (W.si[(size_t)SI_DONE * W.Bp + b]) return;
  if (!force && (W.si[(size_t)SI_RETRY * W.Bp + b] || W.si[(size_t)SI_SKIP_EVAL * W.Bp + b])) return;  // blocks of the last launch are still valid
  const double hdt = K.o.t_step;
  const double eps = W.st[(size_t)ST_EPS * W.Bp + b];
  const bool reinit = W.si[(size_t)SI_REINIT * W.Bp + b] != 0;  // the restoration phase starts with this evaluation
  if (reinit) reinit_slot<BP>(K, W, k, b);
  Slot S;
  linearise_slot<true, BP, ELL>(K, W, k, b, eps, S);
  // ---- node block of x_{k+1}: complete after the linearisation, stored now so that its 52 registers are free during
  //      the elimination (stores issued late also cost more than their bandwidth: on gfx9 a later scratch reload
  //      has to wait for every store before it, vmcnt counts both)
#pragma unroll
  for (int i = 0; i < 36; i++) PG(W.QP, QP_Qx + i, k + 1, QP_NF) = S.Hxp[i];
#pragma unroll
  for (int i = 0; i < 8; i++) PG(W.QP, QP_qx0 + i, k + 1, QP_NF) = S.gxp0[i], PG(W.QP, QP_qx1 + i, k + 1, QP_NF) = S.gxp1[i];
  // ---- residual partials (IPOPT's E_mu ingredients) ----
  {
    double l1[8], l2[8], l1n[8], l2n[8], rd = 0.0, rp = 0.0, sm = 0.0;
    // (the next slot's multipliers in the same batch: read where they are used, each pair sat in a basic block of its own
    //  behind the test below, eight serialised round trips; the last slot reads its own and does not use them)
    const int kn = k + 1 < N ? k + 1 : k;
    const bool nxt = k + 1 < N && !reinit;  // (re-initialised: the next slot's multipliers are zeros)
#pragma unroll
    for (int i = 0; i < 8; i++) l1[i] = PL(W.L1, i, k, N), l2[i] = PL(W.L2, i, k, N), l1n[i] = PL(W.L1, i, kn, N), l2n[i] = PL(W.L2, i, kn, N);
#pragma unroll
    for (int a = 0; a < 8; a++) {
      double rcx = S.dcd[a] + 4.5 * l2[a];
      double rxp = S.dxd[a] - 0.5 * l1[a];
#pragma unroll
      for (int i = 0; i <= ge_(a); i++)  // column a of E1 / E2: rows of the groups up to a's
        if (a >= elo_(i) && a <= ehi_(i)) rcx += S.E1[i * 8 + a] * l1[i], rxp += S.E2[i * 8 + a] * l2[i];
      {
        const double t = 2.0 * l1n[a] - 2.0 * l2n[a];
        const double r2 = rxp + t;
        rxp = nxt ? r2 : rxp;
      }
      rd = fmax(rd, fmax(fabs(rcx), fabs(rxp)));
      rp = fmax(rp, fmax(fabs(S.G1[a]), fabs(S.G2[a])));
      sm += fabs(l1[a]) + fabs(l2[a]);
    }
    double cost = S.cost;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      double v = k ? PL(W.U, i, k - 1, N) : W.uprev[(size_t)i * W.Bp + b];
      double du = S.u[i] - v;
      cost += K.p.r_du[i] * du * du;
      double ru = S.dud[i] + 2.0 * K.p.r_du[i] * du + hdt * (l1[6 + i] + l2[6 + i]);
      if (k + 1 < N) ru -= 2.0 * K.p.r_du[i] * (PL(W.U, i, k + 1, N) - S.u[i]);
      rd = fmax(rd, fabs(ru));
    }
    rp = fmax(rp, S.rp_ineq), sm += S.smult;
    const double cmax = S.cmax, cmin = S.cmin;
    // filter measures of the current point (candidate 0 of the line search) come for free here
    double th0 = S.th_ineq;
#pragma unroll
    for (int a = 0; a < 8; a++) th0 += fabs(S.G1[a]) + fabs(S.G2[a]);
    PL(W.LS, 0, k, N) = th0, PL(W.LS, 1, k, N) = cost, PL(W.LS, 2, k, N) = S.sumlog;
    PL(W.RS, RS_rd, k, N) = rd, PL(W.RS, RS_rp, k, N) = rp, PL(W.RS, RS_cmax, k, N) = cmax;
    PL(W.RS, RS_cmin, k, N) = cmin, PL(W.RS, RS_smult, k, N) = sm, PL(W.RS, RS_cost, k, N) = cost;
    PL(W.RS, RS_emax, k, N) = S.emax;
  }
  // ---- eliminate the collocation point, project its QP block onto (x_k, u_k) ----
  M8Blocks M8;
  double Y[88], AB[88];
  condense_slot(K, S, M8, Y, AB);
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) PG(W.QP, QP_A + i * 8 + j, k, QP_NF) = AB[i * 11 + j];
    PG(W.QP, QP_B + i * 2 + 0, k, QP_NF) = AB[i * 11 + 8], PG(W.QP, QP_B + i * 2 + 1, k, QP_NF) = AB[i * 11 + 9];
    PG(W.QP, QP_b + i, k, QP_NF) = AB[i * 11 + 10];
  }
  double HY[88];  // Hc * [Ac | Bc | bc]; column col of Y is non-zero in rows 0 .. yrow_(col)
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int col = 0; col < 11; col++) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l <= yrow_(col); l++)
        if (hnz_(i, l)) s += sym_get(S.Hc, i, l) * Y[l * 11 + col];
      HY[i * 11 + col] = s;
    }
  // Q = Ac^T Hc Ac, S = Bc^T Hc Ac, R = Bc^T Hc Bc + Du; q = [Ac|Bc]^T (Hc bc + gc0 + mu gc1) (+ gub)
#pragma unroll
  for (int i = 0; i < 10; i++) {
#pragma unroll
    for (int j = 0; j < 10; j++) {
      if (j > i) continue;
      double s = 0.0;
#pragma unroll
      for (int l = 0; l <= yrow_(i); l++) s += Y[l * 11 + i] * HY[l * 11 + j];
      if (i < 8) PG(W.QP, QP_Q + sidx(i, j), k, QP_NF) = s;
      else if (j < 8) PG(W.QP, QP_S + (i - 8) * 8 + j, k, QP_NF) = s;
      else PG(W.QP, QP_R + sidx(i - 8, j - 8), k, QP_NF) = s + ((i == j) ? S.Du[i - 8] : 0.0);
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int l = 0; l <= yrow_(i); l++) s0 += Y[l * 11 + i] * (HY[l * 11 + 10] + S.gc0[l]), s1 += Y[l * 11 + i] * S.gc1[l];
    if (i < 8) PG(W.QP, QP_q0 + i, k, QP_NF) = s0, PG(W.QP, QP_q1 + i, k, QP_NF) = s1;
    else PG(W.QP, QP_r0 + i - 8, k, QP_NF) = s0 + S.gub0[i - 8], PG(W.QP, QP_r1 + i - 8, k, QP_NF) = s1 + S.gub1[i - 8];
  }
}

template <class BP, bool ELL>
__global__ void __launch_bounds__(64) k_eval(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int j = tid % la.n_pad, k = tid / la.n_pad;
  if (k >= W.N || j >= la.nact[0]) return;
  d_eval<BP, ELL>(K, W, k, la.act[j], la.force_eval != 0);
}


// ------------------------------------------------------------------------------------------ k_expand
template <class BP, bool ELL>
__device__ __forceinline__ void d_expand(const Consts& K, const Work& W, const int k, const int b) {
  const int N = W.N;
  if (W.si[(size_t)SI_DONE * W.Bp + b] || !W.si[(size_t)SI_STEP * W.Bp + b]) return;  // no step this launch
  const double mu = W.st[(size_t)ST_MU * W.Bp + b], eps = W.st[(size_t)ST_EPS * W.Bp + b];
  const double tau = W.st[(size_t)ST_TAU * W.Bp + b], rho = W.st[(size_t)ST_RHO * W.Bp + b];
  // costate pi_{k+1} = P_{k+1} dx_{k+1} + Pxv_{k+1} du_k + p_{k+1}: needs nothing of the linearisation, so it comes
  // first (its 60 words of the Riccati buffer are fetched in one batch while few registers are live; placed after the
  // elimination the compiler issued them one at a time, 40 serialised round trips).  Only pi is kept across the
  // elimination, the steps are fetched again afterwards.
  double pi[8];
  {
    double dxp0[8], du0[2];
#pragma unroll
    for (int i = 0; i < 8; i++) dxp0[i] = PL(W.dX, i, k + 1, N + 1);
    du0[0] = PL(W.dU, 0, k, N), du0[1] = PL(W.dU, 1, k, N);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double s = PG(W.RC, RC_pp + i, k + 1, RC_NF) + PG(W.RC, RC_Pxv + i * 2, k + 1, RC_NF) * du0[0] +
                 PG(W.RC, RC_Pxv + i * 2 + 1, k + 1, RC_NF) * du0[1];
#pragma unroll
      for (int j = 0; j < 8; j++) s += PG(W.RC, RC_P + sidx(i, j), k + 1, RC_NF) * dxp0[j];
      pi[i] = s;
    }
  }
  // (nothing may be scheduled across this point: without it the compiler hoists the ~100 loads of the linearisation
  //  above the costate, runs out of registers and spills the Riccati words one by one, each spill waiting for its load)
  __builtin_amdgcn_sched_barrier(0);
  Slot S;
  linearise_slot<false, BP, ELL>(K, W, k, b, eps, S);
  M8Blocks M8;
  factor_m8(S, M8);
  double dxk[8], dxp[8], du[2], dc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) dxk[i] = PL(W.dX, i, k, N + 1), dxp[i] = PL(W.dX, i, k + 1, N + 1);
  du[0] = PL(W.dU, 0, k, N), du[1] = PL(W.dU, 1, k, N);
  // dc_k from the eliminated equations directly, one solve with the actual right-hand side (k_eval needs the 11
  // columns of [Ac | Bc | bc] for the projection of the Hessian, this kernel only their product with (dx, du, 1)):
  //   M8 dc = 2 dx - h Bu du - G2 - 2 E2 w ,   w = 2 dx + h Bu du + G1        (see condense_column)
  {
    const double hdt = K.o.t_step;
    double w[8], v[8];
#pragma unroll
    for (int l = 0; l < 8; l++) w[l] = 2.0 * dxk[l] + (l >= 6 ? hdt * du[l - 6] : 0.0) + S.G1[l];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double s = 2.0 * dxk[i] - (i >= 6 ? hdt * du[i - 6] : 0.0) - S.G2[i];
#pragma unroll
      for (int l = elo_(i); l <= ehi_(i); l++) s -= 2.0 * S.E2[i * 8 + l] * w[l];
      v[i] = s;
    }
    m8_solve<7>(M8, v, dc);
#pragma unroll
    for (int i = 0; i < 8; i++) PL(W.dC, i, k, N) = dc[i];
  }
  // new collocation multipliers:  M8^T l2 = -(Hc dc + gc) - 2 E1^T pi ;  l1 = 2 (E2^T l2 + pi)
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double s = S.gc0[i] + mu * S.gc1[i];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (hnz_(i, j)) s += sym_get(S.Hc, i, j) * dc[j];
      if (i >= elo_(j) && i <= ehi_(j)) s += 2.0 * S.E1[j * 8 + i] * pi[j];
    }
    v[i] = -s;
  }
  {
    double l2[8];
    m8_solve_t(M8, v, l2);
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = l2[i];
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double s = pi[i];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (i >= elo_(j) && i <= ehi_(j)) s += S.E2[j * 8 + i] * v[j];
    PL(W.nL1, i, k, N) = 2.0 * s;
    PL(W.nL2, i, k, N) = v[i];
  }
  // slack / multiplier steps, fraction to the boundary, directional derivative of the barrier objective
  double r_pri = 0.0, a_dua = 1.0, gphid = 0.0;  // r_pri = max(-dt / t) over the slot's inequalities
#pragma unroll
  for (int i = 0; i < 2; i++) {
    double v0 = k ? PL(W.U, i, k - 1, N) : W.uprev[(size_t)i * W.Bp + b];
    double dv0 = k ? PL(W.dU, i, k - 1, N) : 0.0;
    gphid += 2.0 * K.p.r_du[i] * (S.u[i] - v0) * (du[i] - dv0);
  }
#pragma unroll
  for (int a = 0; a < 8; a++) gphid += S.gcost[a] * dxp[a];
  // (Code-generation hazard on ROCm 7.2 / gfx950, seen twice in this function: run-to-run varying values in the LAST
  //  interval only (gphid; later dT / dNU of three Radau-point bounds), each time in a build of this kernel with
  //  100 - 160 spilled SGPRs (v_writelane / v_readlane), gone after an unrelated change of the source.  The kernel
  //  now reads K and W through pointers (under 30 spilled SGPRs); tests/test_gpu_parity.py guards against a return:
  //  test_full_size_batch_properties (repeatability) and the bit-equality of the two instantiations in
  //  test_compaction_and_serial_riccati_do_not_change_results.)
  // With a compile-time bound pattern the slacks and multipliers are fetched in one batch before the loop (on gfx9 a
  // load that follows a store waits for the store as well, vmcnt counts both in order: interleaved with the stores
  // of dT / dNU every bound cost a load AND a store round trip).  Run-time pattern: the index m is not a constant,
  // an array indexed by it would live in scratch memory, so that path loads where it uses.
  double tt[BP::fixed ? MAX_NI + 3 : 1], nn[BP::fixed ? MAX_NI : 1];
  if (BP::fixed) {
    const int m0 = for_each_bound<BP>(K.p, [&](int m, int, int, double, double) { tt[m] = PL(W.T, m, k, N), nn[m] = PL(W.NU, m, k, N); });
    if (S.nl) {
#pragma unroll
      for (int q = 0; q < 3; q++) {
        tt[m0 + q] = PL(W.T, m0 + q, k, N), nn[m0 + q] = PL(W.NU, m0 + q, k, N);
        tt[m0 + 3 + q] = rho > 0.0 ? PL(W.T, K.bd.ni + q, k, N) : 0.0;
      }
    }
  }
  for_each_bound<BP>(K.p, [&](int m, int kind, int j, double sg, double val) {
    const double xv = kind == 0 ? S.u[j] : (kind == 1 ? S.c[j] : S.xp[j]);
    const double dv = kind == 0 ? du[j] : (kind == 1 ? dc[j] : dxp[j]);
    const double t = BP::fixed ? tt[m] : PL(W.T, m, k, N), nu = BP::fixed ? nn[m] : PL(W.NU, m, k, N), it = 1.0 / t;
    const double dtt = -(sg * (xv - val) + t) - sg * dv;
    const double dn = (mu - nu * dtt) * it - nu;
    PL(W.dT, m, k, N) = dtt, PL(W.dNU, m, k, N) = dn;
    r_pri = fmax(r_pri, -dtt * it);  // fraction to the boundary: alpha <= tau t / (-dt) for dt < 0, i.e. tau / max(-dt / t)
    if (dn < 0.0) a_dua = fmin(a_dua, -tau * nu / dn);
    gphid -= mu * dtt * it;
  });
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int m = S.m_nl + q;
    if (S.nl) {
      const double t = BP::fixed ? tt[m] : PL(W.T, m, k, N), nu = BP::fixed ? nn[m] : PL(W.NU, m, k, N), it = 1.0 / t;
      const double gd = S.gs[q] * dxp[0] + S.gn[q] * dxp[1] + S.gm[q] * dxp[2];
      if (rho > 0.0) {
        double dtt, dn, dee;
        track_soft_step(rho, mu, tau, S.gv[q], gd, t, nu, BP::fixed ? tt[m + 3] : PL(W.T, K.bd.ni + q, k, N), dtt, dn, dee, r_pri, a_dua, gphid);
        PL(W.dT, m, k, N) = dtt, PL(W.dNU, m, k, N) = dn, PL(W.dT, K.bd.ni + q, k, N) = dee;
      } else {
        const double dtt = -(S.gv[q] + t) - gd;
        const double dn = (mu - nu * dtt) * it - nu;
        PL(W.dT, m, k, N) = dtt, PL(W.dNU, m, k, N) = dn;
        r_pri = fmax(r_pri, -dtt * it);
        if (dn < 0.0) a_dua = fmin(a_dua, -tau * nu / dn);
        gphid -= mu * dtt * it;
      }
    } else {
      PL(W.dT, m, k, N) = 0.0, PL(W.dNU, m, k, N) = 0.0, PL(W.dT, K.bd.ni + q, k, N) = 0.0;
    }
  }
  if (ELL) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int m = S.m_nl + 3 + q, me = K.bd.ni + 3 + q;
      if (S.nl) {
        const double t = PL(W.T, m, k, N), nu = PL(W.NU, m, k, N), e = PL(W.T, me, k, N);
        double gd = 0.0;
#pragma unroll
        for (int a = 0; a < 5; a++) gd += S.ge[q][a] * dxp[3 + a];
        double dtt, dn, dee;
        track_soft_step(K.p.ell_penalty, mu, tau, S.gve[q], gd, t, nu, e, dtt, dn, dee, r_pri, a_dua, gphid);
        PL(W.dT, m, k, N) = dtt, PL(W.dNU, m, k, N) = dn, PL(W.dT, me, k, N) = dee;
      } else {
        PL(W.dT, m, k, N) = 0.0, PL(W.dNU, m, k, N) = 0.0, PL(W.dT, me, k, N) = 0.0;
      }
    }
  }
  const double a_pri = r_pri > tau ? tau / r_pri : 1.0;
  PL(W.SP, SP_apri, k, N) = a_pri, PL(W.SP, SP_adua, k, N) = a_dua, PL(W.SP, SP_gphid, k, N) = gphid;
}

template <class BP, bool ELL>
__global__ void __launch_bounds__(64) k_expand(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int j = tid % la.n_pad, k = tid / la.n_pad;
  if (k >= W.N || j >= la.nact[0]) return;
  d_expand<BP, ELL>(K, W, k, la.act[j]);
}

}  // namespace ltompc
