// Wave-cooperative evaluation kernels: 8 lanes per (interval, instance) slot, 8 slots per wavefront.
//
// The thread-per-slot kernels (k_eval / k_expand in linearise.h) keep ~450 doubles of one slot alive in one lane: 512
// registers per lane, one wavefront per SIMD, hundreds of spilled registers, and an in-order pipeline that shows the
// full latency of every dependent fp64 operation (measured: 13 cycles per instruction, 22 % VALU-active).  Here lane
// (g, i) owns ROW i of the slot's 8x8 blocks (g = slot in the wavefront), the model is evaluated once per lane (lanes
// 0-3 at the collocation point c_k, lanes 4-7 at the node x_{k+1}, straight into LDS) and the linear algebra exchanges
// rows through LDS: ~60 live doubles per lane, several wavefronts per SIMD, an eighth of the work per wavefront.
//
// Same quantities as linearise_slot / condense_slot / d_eval / d_expand (the formulas are derived there); the
// elimination of the collocation point uses a row-parallel Gauss-Jordan sweep instead of LU + substitutions.
#pragma once
#include "linearise.h"

namespace ltompc {

struct E8Lds {       // per slot
  double r1[96];     // J(c) (6 x 8) | J(x+) (6 x 8)                      -> later Y = [Ac | Bc | bc] (8 x 11)
  double r2[88];     // Hc (36, packed) | Hx+ (36) | f(c) (8) | f(x+) (8)  -> later Hc Y (8 x 11)
  double misc[32];   // gradient of the node cost (8)
  double piv[28];    // pivot row of the elimination: M (8) | V (<= 19) | 1 / pivot
  double vec[64];    // 8 x 8 exchange buffer (transposed sums) / shared vectors / dump for the unused half of a call
};

// sum over the 8 lanes j of a slot of v_j[i], delivered to lane i (v: the 8 values lane j contributes)
__device__ __forceinline__ double tsum8(E8Lds& L, const int i, const double* v) {
  WAVE_SYNC();
#pragma unroll
  for (int j = 0; j < 8; j++) L.vec[i * 8 + j] = v[j];
  WAVE_SYNC();
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; j++) s += L.vec[j * 8 + i];
  return s;
}
// every lane of the slot contributes one value, all get the vector
__device__ __forceinline__ void share8(E8Lds& L, const int i, const double mine, double* all) {
  WAVE_SYNC();
  L.vec[i] = mine;
  WAVE_SYNC();
#pragma unroll
  for (int j = 0; j < 8; j++) all[j] = L.vec[j];
}

struct Lin8 {  // what lane (g, i) holds of its slot after lin8()
  double xp[8], u[2];
  double c_i, xp_i;
  double E1r[8], E2r[8], Hcr[8], Hxr[8];  // rows i
  double G1, G2;
  double gc0, gc1, gx0, gx1;    // QP gradients of c_k and x_{k+1}, component i (= g0 + mu g1)
  double Du, gub0, gub1;        // lanes i < 2: input-bound barrier of u_i
  double dcd, dxd, dud;         // DUAL: parts of grad_z L without the collocation multipliers
  double gcost;                 // d cost / d x+_i
  double cost;                  // node cost (same in all lanes)
  double gv[3], gs[3], gn[3], gm[3];  // track constraints at x+ (same in all lanes)
  double rp_ineq, cmax, cmin, smult, th_ineq, sumlog;  // DUAL: per-lane partials over the inequalities this lane owns
  double csoft;                 // DUAL: penalty rho e of the soft track constraint this lane owns
  double emax;                  // DUAL: its elastic variable
  double Yr[11], ABr[11];       // rows i of [Ac | Bc | bc] and [A | B | b]
  double Mir[8];                // EXPAND: row i of M8^-1
  int m_nl;
  bool nl;
};

template <bool DUAL, bool EXPAND>
__device__ __forceinline__ void lin8(const Consts& K, const Work& W, E8Lds& L, const int i, const int k, const int b,
                                     const double eps, Lin8& S) {
  const int N = W.N;
  const double hdt = K.o.t_step, rho = W.st[(size_t)ST_RHO * W.Bp + b];
  const int pt = i >> 2;  // 0: this lane evaluates the model at c_k, 1: at x_{k+1}
  // ---- inputs
  double px[8], lam[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    S.xp[j] = PL(W.X, j, k + 1, N + 1);
    const double cj = PL(W.C, j, k, N);
    px[j] = pt ? S.xp[j] : cj;
    const double a = PL(W.L1, j, k, N), c = PL(W.L2, j, k, N);
    lam[j] = pt ? c : a;
  }
  S.u[0] = PL(W.U, 0, k, N), S.u[1] = PL(W.U, 1, k, N);
  const double xk_i = k == 0 ? W.x0[(size_t)i * W.Bp + b] : PL(W.X, i, k, N + 1);
  S.c_i = PL(W.C, i, k, N), S.xp_i = PL(W.X, i, k + 1, N + 1);
  // ---- model, straight into LDS (the 4 lanes of a half write the same values to the same addresses)
  WAVE_SYNC();
#pragma unroll
  for (int q = 0; q < 9; q++) L.r2[i * 9 + q] = 0.0;  // Hc, Hx+
  L.misc[i] = 0.0;
  WAVE_SYNC();
  {
    double f[8];
    rhs_derivs(K.p, K.T, eps, px, f, &L.r1[pt * 48], lam, hdt, &L.r2[pt * 36]);
#pragma unroll
    for (int q = 0; q < 6; q++) L.r2[72 + pt * 8 + q] = f[q];
  }
  S.cost = cost_eval(K.p, K.T, eps, S.xp, k == N - 1, pt ? &L.misc[0] : &L.vec[0], pt ? &L.r2[36] : &L.vec[8]);
  S.nl = (k + 1 <= N - 1);
  double hss[3] = {0, 0, 0}, hmm[3] = {0, 0, 0};
  if (S.nl) {
    cons_eval(K.p, K.T, eps, S.xp, S.gv, S.gs, S.gn, S.gm, hss, hmm);
  } else {
#pragma unroll
    for (int q = 0; q < 3; q++) S.gv[q] = -1.0, S.gs[q] = S.gn[q] = S.gm[q] = 0.0;
  }
  WAVE_SYNC();
  // ---- rows
  double f1, f2;
  if (i < 6) f1 = L.r2[72 + i], f2 = L.r2[80 + i];
  else f1 = f2 = (i == 6 ? S.u[0] : S.u[1]);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const double jc = i < 6 ? L.r1[i * 8 + j] : 0.0, jx = i < 6 ? L.r1[48 + i * 8 + j] : 0.0;
    S.E1r[j] = hdt * jc - (i == j ? 1.5 : 0.0);
    S.E2r[j] = hdt * jx - (i == j ? 2.5 : 0.0);
    S.Hcr[j] = L.r2[sidx(i, j)];
    S.Hxr[j] = L.r2[36 + sidx(i, j)];
  }
  S.gcost = L.misc[i];
  S.G1 = hdt * f1 + 2.0 * xk_i - 1.5 * S.c_i - 0.5 * S.xp_i;
  S.G2 = hdt * f2 - 2.0 * xk_i + 4.5 * S.c_i - 2.5 * S.xp_i;
  // ---- inequalities: the lane that owns component j adds the barrier terms of the bounds on it
  S.gc0 = 0.0, S.gc1 = 0.0, S.gx0 = S.gcost, S.gx1 = 0.0, S.dcd = 0.0, S.dxd = S.gcost;
  S.Du = 0.0, S.gub0 = 0.0, S.gub1 = 0.0, S.dud = 0.0;
  S.rp_ineq = 0.0, S.cmax = 0.0, S.cmin = 1e300, S.smult = 0.0, S.th_ineq = 0.0, S.sumlog = 0.0, S.csoft = 0.0, S.emax = 0.0;
  double hcd = 0.0, hxd = 0.0;  // additions to the diagonal entries Hc[i][i], Hx+[i][i]
  double lprod = 1.0;           // product of the slacks this lane owns (at most 6): one logarithm per lane
  S.m_nl = for_each_bound<BoundsAny>(K.p, [&](int m, int kind, int j, double sg, double val) {
    if (j != i) return;
    const double xv = kind == 0 ? (j == 0 ? S.u[0] : S.u[1]) : (kind == 1 ? S.c_i : S.xp_i);
    const double hv = sg * (xv - val);
    const double t = PL(W.T, m, k, N), nu = PL(W.NU, m, k, N), it = 1.0 / t;
    const double Sg = nu * it, g0 = sg * nu * (hv + t) * it, g1 = sg * it;
    if (kind == 0) {
      S.Du += Sg, S.gub0 += g0, S.gub1 += g1;
      if (DUAL) S.dud += sg * nu;
    } else if (kind == 1) {
      hcd += Sg, S.gc0 += g0, S.gc1 += g1;
      if (DUAL) S.dcd += sg * nu;
    } else {
      hxd += Sg, S.gx0 += g0, S.gx1 += g1;
      if (DUAL) S.dxd += sg * nu;
    }
    if (DUAL) {
      S.rp_ineq = fmax(S.rp_ineq, fabs(hv + t));
      S.cmax = fmax(S.cmax, t * nu), S.cmin = fmin(S.cmin, t * nu), S.smult += fabs(nu);
      S.th_ineq += fabs(hv + t), lprod *= t;
    }
  });
  if (S.nl) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int mm = S.m_nl + q;
      const double t = PL(W.T, mm, k, N), nu = PL(W.NU, mm, k, N);
      const double e = rho > 0.0 ? PL(W.T, mm + 3, k, N) : 0.0;
      double Sg, s0, s1;
      track_barrier(rho, S.gv[q], t, nu, e, Sg, s0, s1);
      const double g3[3] = {S.gs[q], S.gn[q], S.gm[q]};
      const double ga = i == 0 ? g3[0] : (i == 1 ? g3[1] : g3[2]);
      if (i < 3) {
        S.gx0 += s0 * ga, S.gx1 += s1 * ga;
        if (DUAL) S.dxd += nu * ga;
#pragma unroll
        for (int c = 0; c < 3; c++) S.Hxr[c] += Sg * ga * g3[c];
      }
      if (i == 0) S.Hxr[0] += nu * hss[q];
      if (i == 2) S.Hxr[2] += nu * hmm[q];
      if (DUAL && i == q) {
        S.rp_ineq = fmax(S.rp_ineq, fabs(S.gv[q] - e + t));
        S.cmax = fmax(S.cmax, t * nu), S.cmin = fmin(S.cmin, t * nu), S.smult += fabs(nu);
        S.th_ineq += fabs(S.gv[q] - e + t), lprod *= t;
        if (rho > 0.0) {
          const double ez = e * (rho - nu);
          S.cmax = fmax(S.cmax, ez), S.cmin = fmin(S.cmin, ez), S.smult += fabs(rho - nu);
          lprod *= e, S.csoft = rho * e, S.emax = e;
        }
      }
    }
  }
  if (DUAL) S.sumlog = log(lprod);
#pragma unroll
  for (int j = 0; j < 8; j++)
    if (j == i) S.Hcr[j] += hcd, S.Hxr[j] += hxd;
  // ---- M8 = 4.5 I + 2 E2 E1 (row i) and the right-hand sides V = [2I - 4 E2 | -h (e + 2 E2 e) | -G2 - 2 E2 G1 | I]
  constexpr int NV = EXPAND ? 19 : 11;
  double M[8], V[NV];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < 6; l++) s += S.E2r[l] * (hdt * L.r1[l * 8 + j]);
    s -= 1.5 * S.E2r[j];  // the -1.5 I part of E1 (rows 6, 7 of E1 are only that)
    M[j] = (i == j ? 4.5 : 0.0) + 2.0 * s;
  }
  double G1all[8];
  share8(L, i, S.G1, G1all);
#pragma unroll
  for (int c = 0; c < 8; c++) V[c] = (i == c ? 2.0 : 0.0) - 4.0 * S.E2r[c];
  V[8] = -hdt * ((i == 6 ? 1.0 : 0.0) + 2.0 * S.E2r[6]);
  V[9] = -hdt * ((i == 7 ? 1.0 : 0.0) + 2.0 * S.E2r[7]);
  {
    double s = -S.G2;
#pragma unroll
    for (int l = 0; l < 8; l++) s -= 2.0 * S.E2r[l] * G1all[l];
    V[10] = s;
  }
  if (EXPAND) {
#pragma unroll
    for (int c = 0; c < 8; c++) V[11 + c] = (i == c ? 1.0 : 0.0);
  }
  // ---- row-parallel Gauss-Jordan (no pivoting: M8 = 4.5 I + 2 E2 E1 has eigenvalues 12 - 8 lambda h + 2 (lambda h)^2)
  double dinv = 1.0;
#pragma unroll
  for (int kk = 0; kk < 8; kk++) {
    WAVE_SYNC();
    if (i == kk) {
      const double rp = 1.0 / M[kk];
      dinv = rp;
#pragma unroll
      for (int j = kk + 1; j < 8; j++) L.piv[j] = M[j];
#pragma unroll
      for (int c = 0; c < NV; c++) L.piv[8 + c] = V[c];
      L.piv[27] = rp;
    }
    WAVE_SYNC();
    if (i != kk) {
      const double fac = M[kk] * L.piv[27];
#pragma unroll
      for (int j = kk + 1; j < 8; j++) M[j] -= fac * L.piv[j];
#pragma unroll
      for (int c = 0; c < NV; c++) V[c] -= fac * L.piv[8 + c];
    }
  }
#pragma unroll
  for (int c = 0; c < 11; c++) S.Yr[c] = V[c] * dinv;
  if (EXPAND) {
#pragma unroll
    for (int c = 0; c < 8; c++) S.Mir[c] = V[11 + c] * dinv;
  }
  // ---- Y to LDS (over the Jacobians: every lane has its rows by now), [A | B | b] = 2 ([2I | h e | G1] + E1 Y)
  WAVE_SYNC();
#pragma unroll
  for (int c = 0; c < 11; c++) L.r1[i * 11 + c] = S.Yr[c];
  WAVE_SYNC();
}

// ------------------------------------------------------------------------------------------ k_eval8
__device__ __forceinline__ void d_eval8(const Consts& K, const Work& W, E8Lds& L, const int i, const int k, const int b,
                                        const bool live) {
  const int N = W.N;
  const double hdt = K.o.t_step;
  const double eps = W.st[(size_t)ST_EPS * W.Bp + b];
  const bool reinit = W.si[(size_t)SI_REINIT * W.Bp + b] != 0;  // the restoration phase starts with this evaluation (see d_eval)
  if (reinit) {
    if (live && i == 0) reinit_slot<BoundsAny>(K, W, k, b);
    WAVE_SYNC();
    __threadfence();  // the other lanes of the slot read what lane 0 has just stored
  }
  Lin8 S;
  lin8<true, false>(K, W, L, i, k, b, eps, S);
  // ---- residual partials (IPOPT's E_mu ingredients), see d_eval
  {
    const double l1_i = PL(W.L1, i, k, N), l2_i = PL(W.L2, i, k, N);
    double v[8];
#pragma unroll
    for (int a = 0; a < 8; a++) v[a] = S.E1r[a] * l1_i;
    double rcx = S.dcd + 4.5 * l2_i + tsum8(L, i, v);
#pragma unroll
    for (int a = 0; a < 8; a++) v[a] = S.E2r[a] * l2_i;
    double rxp = S.dxd - 0.5 * l1_i + tsum8(L, i, v);
    if (k + 1 < N && !reinit) rxp += 2.0 * PL(W.L1, i, k + 1, N) - 2.0 * PL(W.L2, i, k + 1, N);
    double rd = fmax(fabs(rcx), fabs(rxp));
    double rp = fmax(fmax(fabs(S.G1), fabs(S.G2)), S.rp_ineq);
    double sm = fabs(l1_i) + fabs(l2_i) + S.smult;
    double cost_u = 0.0;
    if (i < 2) {
      const double ui = i == 0 ? S.u[0] : S.u[1];
      const double v0 = k ? PL(W.U, i, k - 1, N) : W.uprev[(size_t)i * W.Bp + b];
      const double du = ui - v0;
      cost_u = K.p.r_du[i] * du * du;
      double ru = S.dud + 2.0 * K.p.r_du[i] * du + hdt * (PL(W.L1, 6 + i, k, N) + PL(W.L2, 6 + i, k, N));
      if (k + 1 < N) ru -= 2.0 * K.p.r_du[i] * (PL(W.U, i, k + 1, N) - ui);
      rd = fmax(rd, fabs(ru));
    }
    rd = grp_max(rd), rp = grp_max(rp), sm = grp_sum(sm);
    const double cmax = grp_max(S.cmax), cmin = grp_min(S.cmin);
    const double cost = S.cost + grp_sum(cost_u + S.csoft);
    const double th0 = grp_sum(S.th_ineq + fabs(S.G1) + fabs(S.G2)), sumlog = grp_sum(S.sumlog);
    const double emax = grp_max(S.emax);
    if (live && i == 0) {
      PL(W.LS, 0, k, N) = th0, PL(W.LS, 1, k, N) = cost, PL(W.LS, 2, k, N) = sumlog;
      PL(W.RS, RS_rd, k, N) = rd, PL(W.RS, RS_rp, k, N) = rp, PL(W.RS, RS_cmax, k, N) = cmax;
      PL(W.RS, RS_cmin, k, N) = cmin, PL(W.RS, RS_smult, k, N) = sm, PL(W.RS, RS_cost, k, N) = cost;
      PL(W.RS, RS_emax, k, N) = emax;
    }
  }
  // ---- [A | B | b] (row i) and Hc Y (row i) from one pass over Y
  double HYr[11];
#pragma unroll
  for (int c = 0; c < 11; c++) {
    double sa = c < 8 ? (i == c ? 2.0 : 0.0) : (c < 10 ? (i == 6 + c - 8 ? hdt : 0.0) : S.G1);
    double sh = 0.0;
#pragma unroll
    for (int l = 0; l < 8; l++) {
      const double y = L.r1[l * 11 + c];
      sa += S.E1r[l] * y, sh += S.Hcr[l] * y;
    }
    S.ABr[c] = 2.0 * sa, HYr[c] = sh;
  }
  WAVE_SYNC();
#pragma unroll
  for (int c = 0; c < 11; c++) L.r2[i * 11 + c] = HYr[c];
  double gc0all[8], gc1all[8];
  share8(L, i, S.gc0, gc0all);
  share8(L, i, S.gc1, gc1all);
  WAVE_SYNC();
  // ---- condensed Hessian / gradient: row i of [Ac | Bc]^T Hc [Ac | Bc | bc]; lanes 0, 1 also do row 8 + i (the inputs)
  double T[10], q0 = 0.0, q1 = 0.0;
#pragma unroll
  for (int j = 0; j < 10; j++) T[j] = 0.0;
#pragma unroll
  for (int l = 0; l < 8; l++) {
    const double y = L.r1[l * 11 + i];
#pragma unroll
    for (int j = 0; j < 10; j++) T[j] += y * L.r2[l * 11 + j];
    q0 += y * (L.r2[l * 11 + 10] + gc0all[l]), q1 += y * gc1all[l];
  }
  double Tu[10], r0 = 0.0, r1 = 0.0;
#pragma unroll
  for (int j = 0; j < 10; j++) Tu[j] = 0.0;
  if (i < 2) {
#pragma unroll
    for (int l = 0; l < 8; l++) {
      const double y = L.r1[l * 11 + 8 + i];
#pragma unroll
      for (int j = 0; j < 10; j++) Tu[j] += y * L.r2[l * 11 + j];
      r0 += y * (L.r2[l * 11 + 10] + gc0all[l]), r1 += y * gc1all[l];
    }
  }
  if (!live) return;
#pragma unroll
  for (int j = 0; j < 8; j++) PG(W.QP, QP_A + i * 8 + j, k, QP_NF) = S.ABr[j];
  PG(W.QP, QP_B + i * 2 + 0, k, QP_NF) = S.ABr[8], PG(W.QP, QP_B + i * 2 + 1, k, QP_NF) = S.ABr[9];
  PG(W.QP, QP_b + i, k, QP_NF) = S.ABr[10];
#pragma unroll
  for (int j = 0; j < 8; j++)
    if (j <= i) PG(W.QP, QP_Q + sidx(i, j), k, QP_NF) = T[j], PG(W.QP, QP_Qx + sidx(i, j), k + 1, QP_NF) = S.Hxr[j];
  PG(W.QP, QP_q0 + i, k, QP_NF) = q0, PG(W.QP, QP_q1 + i, k, QP_NF) = q1;
  PG(W.QP, QP_qx0 + i, k + 1, QP_NF) = S.gx0, PG(W.QP, QP_qx1 + i, k + 1, QP_NF) = S.gx1;
  if (i < 2) {
#pragma unroll
    for (int j = 0; j < 8; j++) PG(W.QP, QP_S + i * 8 + j, k, QP_NF) = Tu[j];
    PG(W.QP, QP_R + sidx(i, 0), k, QP_NF) = Tu[8] + (i == 0 ? S.Du : 0.0);
    if (i == 1) PG(W.QP, QP_R + sidx(1, 1), k, QP_NF) = Tu[9] + S.Du;
    PG(W.QP, QP_r0 + i, k, QP_NF) = r0 + S.gub0, PG(W.QP, QP_r1 + i, k, QP_NF) = r1 + S.gub1;
  }
}

// One wavefront = 8 slots (instances act[8 grp .. 8 grp + 7] of interval k) x 8 lanes.
__global__ void __launch_bounds__(64) k_eval8(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  __shared__ E8Lds lds[8];
  const int lane = threadIdx.x, g = lane & 7, i = lane >> 3;
  const int G8 = la.n_pad >> 3;
  const int k = blockIdx.x / G8, j = (blockIdx.x % G8) * 8 + g;
  const bool valid = j < la.nact[0];
  const int b = la.act[valid ? j : 0];
  const int* si = W.si;
  const bool live = valid && !si[(size_t)SI_DONE * W.Bp + b] &&
                    (la.force_eval || (!si[(size_t)SI_RETRY * W.Bp + b] && !si[(size_t)SI_SKIP_EVAL * W.Bp + b]));  // else: the blocks of the last launch are still valid
  if (!__any(live)) return;
  d_eval8(K, W, lds[g], i, k, b, live);
}

// ------------------------------------------------------------------------------------------ k_expand8
__device__ __forceinline__ void d_expand8(const Consts& K, const Work& W, E8Lds& L, const int i, const int k, const int b,
                                          const bool live) {
  const int N = W.N;
  const double mu = W.st[(size_t)ST_MU * W.Bp + b], eps = W.st[(size_t)ST_EPS * W.Bp + b];
  const double tau = W.st[(size_t)ST_TAU * W.Bp + b], rho = W.st[(size_t)ST_RHO * W.Bp + b];
  Lin8 S;
  lin8<false, true>(K, W, L, i, k, b, eps, S);
  double dxk[8], dxp[8], du[2];
#pragma unroll
  for (int j = 0; j < 8; j++) dxk[j] = PL(W.dX, j, k, N + 1), dxp[j] = PL(W.dX, j, k + 1, N + 1);
  du[0] = PL(W.dU, 0, k, N), du[1] = PL(W.dU, 1, k, N);
  double dc_i = S.Yr[10] + S.Yr[8] * du[0] + S.Yr[9] * du[1];
#pragma unroll
  for (int j = 0; j < 8; j++) dc_i += S.Yr[j] * dxk[j];
  // costate pi_{k+1} = P_{k+1} dx_{k+1} + Pxv_{k+1} du_k + p_{k+1}
  double pi_i = PG(W.RC, RC_pp + i, k + 1, RC_NF) + PG(W.RC, RC_Pxv + i * 2, k + 1, RC_NF) * du[0] +
                PG(W.RC, RC_Pxv + i * 2 + 1, k + 1, RC_NF) * du[1];
#pragma unroll
  for (int j = 0; j < 8; j++) pi_i += PG(W.RC, RC_P + sidx(i, j), k + 1, RC_NF) * dxp[j];
  double dc[8];
  share8(L, i, dc_i, dc);
  // new collocation multipliers:  M8^T l2 = -(Hc dc + gc) - 2 E1^T pi ;  l1 = 2 (E2^T l2 + pi)
  double v[8];
#pragma unroll
  for (int a = 0; a < 8; a++) v[a] = S.E1r[a] * pi_i;
  const double e1tpi = tsum8(L, i, v);
  double vi = S.gc0 + mu * S.gc1 + 2.0 * e1tpi;
#pragma unroll
  for (int j = 0; j < 8; j++) vi += S.Hcr[j] * dc[j];
  vi = -vi;
#pragma unroll
  for (int a = 0; a < 8; a++) v[a] = S.Mir[a] * vi;
  const double l2_i = tsum8(L, i, v);  // (M8^-1)^T v
#pragma unroll
  for (int a = 0; a < 8; a++) v[a] = S.E2r[a] * l2_i;
  const double l1_i = 2.0 * (pi_i + tsum8(L, i, v));
  // slack / multiplier steps, fraction to the boundary, directional derivative of the barrier objective
  const double dxp_i = sel8(dxp, i);
  double r_pri = 0.0, a_dua = 1.0, gphid = S.gcost * dxp_i;  // r_pri = max(-dt / t), see d_expand
  if (i < 2) {
    const double ui = i == 0 ? S.u[0] : S.u[1], dui = i == 0 ? du[0] : du[1];
    const double v0 = k ? PL(W.U, i, k - 1, N) : W.uprev[(size_t)i * W.Bp + b];
    const double dv0 = k ? PL(W.dU, i, k - 1, N) : 0.0;
    gphid += 2.0 * K.p.r_du[i] * (ui - v0) * (dui - dv0);
  }
  for_each_bound<BoundsAny>(K.p, [&](int m, int kind, int j, double sg, double val) {
    if (j != i) return;
    const double xv = kind == 0 ? (j == 0 ? S.u[0] : S.u[1]) : (kind == 1 ? S.c_i : S.xp_i);
    const double dv = kind == 0 ? (j == 0 ? du[0] : du[1]) : (kind == 1 ? dc_i : dxp_i);
    const double t = PL(W.T, m, k, N), nu = PL(W.NU, m, k, N), it = 1.0 / t;
    const double dtt = -(sg * (xv - val) + t) - sg * dv;
    const double dn = (mu - nu * dtt) * it - nu;
    if (live) PL(W.dT, m, k, N) = dtt, PL(W.dNU, m, k, N) = dn;
    r_pri = fmax(r_pri, -dtt * it);
    if (dn < 0.0) a_dua = fmin(a_dua, -tau * nu / dn);
    gphid -= mu * dtt * it;
  });
  if (i < 3) {  // lane q handles track constraint q
    const int m = S.m_nl + i;
    if (S.nl) {
      const double gvq = i == 0 ? S.gv[0] : (i == 1 ? S.gv[1] : S.gv[2]);
      const double gsq = i == 0 ? S.gs[0] : (i == 1 ? S.gs[1] : S.gs[2]);
      const double gnq = i == 0 ? S.gn[0] : (i == 1 ? S.gn[1] : S.gn[2]);
      const double gmq = i == 0 ? S.gm[0] : (i == 1 ? S.gm[1] : S.gm[2]);
      const double t = PL(W.T, m, k, N), nu = PL(W.NU, m, k, N), it = 1.0 / t;
      const double gd = gsq * dxp[0] + gnq * dxp[1] + gmq * dxp[2];
      if (rho > 0.0) {
        double dtt, dn, dee;
        track_soft_step(rho, mu, tau, gvq, gd, t, nu, PL(W.T, m + 3, k, N), dtt, dn, dee, r_pri, a_dua, gphid);
        if (live) PL(W.dT, m, k, N) = dtt, PL(W.dNU, m, k, N) = dn, PL(W.dT, m + 3, k, N) = dee;
      } else {
        const double dtt = -(gvq + t) - gd;
        const double dn = (mu - nu * dtt) * it - nu;
        if (live) PL(W.dT, m, k, N) = dtt, PL(W.dNU, m, k, N) = dn;
        r_pri = fmax(r_pri, -dtt * it);
        if (dn < 0.0) a_dua = fmin(a_dua, -tau * nu / dn);
        gphid -= mu * dtt * it;
      }
    } else if (live) {
      PL(W.dT, m, k, N) = 0.0, PL(W.dNU, m, k, N) = 0.0, PL(W.dT, m + 3, k, N) = 0.0;
    }
  }
  r_pri = grp_max(r_pri), a_dua = grp_min(a_dua), gphid = grp_sum(gphid);
  const double a_pri = r_pri > tau ? tau / r_pri : 1.0;
  if (!live) return;
  PL(W.dC, i, k, N) = dc_i;
  PL(W.nL1, i, k, N) = l1_i, PL(W.nL2, i, k, N) = l2_i;
  if (i == 0) PL(W.SP, SP_apri, k, N) = a_pri, PL(W.SP, SP_adua, k, N) = a_dua, PL(W.SP, SP_gphid, k, N) = gphid;
}

__global__ void __launch_bounds__(64) k_expand8(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  __shared__ E8Lds lds[8];
  const int lane = threadIdx.x, g = lane & 7, i = lane >> 3;
  const int G8 = la.n_pad >> 3;
  const int k = blockIdx.x / G8, j = (blockIdx.x % G8) * 8 + g;
  const bool valid = j < la.nact[0];
  const int b = la.act[valid ? j : 0];
  const int* si = W.si;
  const bool live = valid && !si[(size_t)SI_DONE * W.Bp + b] && si[(size_t)SI_STEP * W.Bp + b];  // else: no step this launch
  if (!__any(live)) return;
  d_expand8(K, W, lds[g], i, k, b, live);
}

}  // namespace ltompc
