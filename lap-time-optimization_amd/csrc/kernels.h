// kernels.h — HIP kernels of the batched interior-point NLP solver (gfx950).
//
// One make_step (reference src/mpc.py:142, do_mpc MPC.make_step -> IPOPT) = up to max_iter interior-point
// iterations, each a fixed sequence of kernels over the whole batch:
//
//   k_eval    thread = (interval k, instance b)   derivatives of dynamics / cost / constraints at the Radau point
//                                                 and the next node, KKT residual partials, elimination of the
//                                                 collocation variables  -> stage QP blocks (A,B,b,Q,S,R,q,r)
//   k_riccati thread = instance b                 KKT error, termination, barrier update, Riccati backward sweep
//                                                 (with inertia-correcting regularisation) and forward rollout
//   k_expand  thread = (k, b)                     collocation steps, collocation multipliers, slack / inequality
//                                                 multiplier steps, fraction-to-boundary partial minima
//   k_linesearch thread = (k, b)                  filter measures (theta, cost, sum log t) for all step candidates
//   k_pick    thread = b                          filter acceptance test, step length, filter / stall bookkeeping
//   k_update  thread = (k, b)                     z += alpha dz
//
// HBM layout: every per-(k,b) quantity is a plane [field][k][Bp] with the instance index fastest, so that the
// 64 lanes of a wavefront (consecutive b, same k) read/write 512 contiguous bytes per field.
#pragma once
#include "model.h"

namespace ltompc {

constexpr int FILTER_MAX = 16;
constexpr double DW_KEEP = 1e-5;  // regularisation below this is dropped to exactly 0
constexpr int MAX_LS = 12;

// fields of the stage-QP buffer written by k_eval and read by k_riccati
enum : int {
  QP_A = 0,            // 64  A_k   (dx+ = A dx + B du + b)
  QP_B = 64,           // 16
  QP_b = 80,           // 8
  QP_Q = 88,           // 36  condensed-collocation part of the (x_k,x_k) block
  QP_S = 124,          // 16  (u_k, x_k)
  QP_R = 140,          // 3   (u_k, u_k) incl. input-bound barrier
  QP_q0 = 143,         // 8   gradient = q0 + mu * q1
  QP_q1 = 151,         // 8
  QP_r0 = 159,         // 2
  QP_r1 = 161,         // 2
  QP_Qx = 163,         // 36  node block of x_k (cost + constraints + bounds + lambda2-weighted dynamics): written by
                       //     interval k-1 into THIS block, so that stage k of the Riccati sweep reads block k only;
                       //     block N holds the terminal node only, the node part of block 0 is never written (zeros)
  QP_qx0 = 199,        // 8
  QP_qx1 = 207,        // 8
  QP_NF = 215
};
// fields of the Riccati buffer written by k_riccati and read by k_expand; stage index 0..N
enum : int { RC_K = 0, RC_Kv = 16, RC_kff = 20, RC_P = 22, RC_Pxv = 58, RC_pp = 74, RC_NF = 82 };
// residual partials written by k_eval (per k,b)
enum : int { RS_rd = 0, RS_rp, RS_cmax, RS_cmin, RS_smult, RS_cost, RS_NF };
// step partials written by k_expand
enum : int { SP_apri = 0, SP_adua, SP_gphid, SP_NF };
// per-instance double state
enum : int {
  ST_MU = 0, ST_EPS, ST_EPS_NEXT, ST_DW_LAST, ST_FORCE_REG, ST_ALPHA, ST_ADUA, ST_E0, ST_OBJ, ST_TAU,
  ST_THETA0, ST_THMAX, ST_THMIN, ST_DW, ST_DW_TRY,
  ST_C00,  // lterm(x_0) for the current ST_EPS: a constant of the solve between two changes of the table smoothing
  ST_NF
};
// per-instance int state
// SI_LSMORE: the full step was rejected by the filter test, the remaining step candidates have to be evaluated.
// SI_RETRY: the last Riccati sweep failed the inertia test; the next launch repeats it with ST_DW_TRY (no new
// evaluation).  SI_SKIP_EVAL: the iterate did not move (failed line search), k_eval's output is still valid.
enum : int { SI_STATUS = 0, SI_ITERS, SI_NACC, SI_NTINY, SI_NFILT, SI_DONE, SI_STEP, SI_NREG, SI_NLSFAIL, SI_RETRY, SI_TRIES,
             SI_SKIP_EVAL, SI_LSMORE, SI_PREV, SI_NF };  // SI_PREV: status of the previous make_step (k_load_x0)

struct Work {
  int N, B, Bp;
  // iterate
  gptr<double> X, C, U, L1, L2, T, NU;
  // steps
  gptr<double> dX, dC, dU, nL1, nL2, dT, dNU;
  // buffers
  gptr<double> QP, RC, RS, SP, LS;
  gptr<double> x0, uprev;  // [8][Bp], [2][Bp]
  gptr<double> st;     // [ST_NF][Bp]
  gptr<double> filt;   // [2*FILTER_MAX][Bp]
  gptr<int> si;        // [SI_NF][Bp]
  gptr<int> active;    // [max_iter+2] number of unfinished instances after iteration i
  gptr<double> DBG;  // [8][N][Bp] scratch planes for debugging
  gptr<int> ls_list, ls_count;  // instances whose full step was rejected in this iteration (phase 1 of the line search)
};

// What changes from launch to launch (kernel argument; Work and Consts are read from device memory).  Compaction of the
// unfinished instances: thread j of a launch works on instance act[j], j < nact[0] <= the launch width.  The list is
// sorted (stable compaction), so while nothing has finished it is the identity and accesses coalesce.
struct Launch {
  gptr<const int> act;
  gptr<const int> nact;
  int n_pad;  // launch width rounded up to a multiple of 64
};

struct Consts {
  ltompc_params p;
  ltompc_options o;
  Tables T;
  Bounds bd;
};

#define PL(base, f, k, NK) ((base)[((size_t)(f) * (NK) + (k)) * W.Bp + b])
// Stage-QP and Riccati buffers: [k][b / 8][field][b % 8].  A wavefront of k_riccati8 (8 instances x 8 lanes, lane
// (g,i) touching field f0 + i of instance g) then reads/writes 512 contiguous bytes per instruction, and the
// thread-per-(k,b) kernels still move whole 64-byte sectors (8 consecutive instances of one field).
#define PG(base, f, k, NF) ((base)[(((size_t)(k) * (W.Bp >> 3) + (b >> 3)) * (NF) + (f)) * 8 + (b & 7)])

// ------------------------------------------------------------------------------------------ small dense LA
__device__ __forceinline__ double sym_get(const double* H, int i, int j) { return H[sidx(i, j)]; }


// Visits the inequalities of a slot in their storage order (input bounds, Radau-point bounds, node bounds; per
// variable lower then upper, only the bounds that are set).  `f(m, kind, i, sg, val)` gets the running index m,
// kind 0/1/2 = u / c / x+, and the variable index i as a value that is a compile-time constant after unrolling,
// so that per-variable arrays stay in registers (a run-time index would force them into scratch memory).
template <typename F>
__device__ __forceinline__ int for_each_bound(const ltompc_params& p, F&& f) {
  int m = 0;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    if (p.u_lb[i] > -LTOMPC_NO_BOUND) f(m++, 0, i, -1.0, p.u_lb[i]);
    if (p.u_ub[i] < LTOMPC_NO_BOUND) f(m++, 0, i, 1.0, p.u_ub[i]);
  }
#pragma unroll
  for (int kind = 1; kind <= 2; kind++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (p.x_lb[i] > -LTOMPC_NO_BOUND) f(m++, kind, i, -1.0, p.x_lb[i]);
      if (p.x_ub[i] < LTOMPC_NO_BOUND) f(m++, kind, i, 1.0, p.x_ub[i]);
    }
  }
  return m;  // index of the first track constraint
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
  double rp_ineq, cmax, cmin, smult; // WITH_DUAL: max |h + t|, max / min t nu, sum |nu| over the slot's inequalities
  double th_ineq, sumlog;            // WITH_DUAL: sum |h + t|, sum log t (filter measures of the current point)
  double cost;
  int m_nl;                          // storage index of gL
  bool nl;
};

template <bool WITH_DUAL>
__device__ __forceinline__ void linearise_slot(const Consts& K, const Work& W, int k, int b, double eps, Slot& S) {
  const int N = W.N;
  const double hdt = K.o.t_step;
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
  S.rp_ineq = 0.0, S.cmax = 0.0, S.cmin = 1e300, S.smult = 0.0, S.th_ineq = 0.0, S.sumlog = 0.0;
  double lprod = 1.0;  // sum of log t = log of products of 8 slacks (3 logarithms per slot, see d_linesearch)
  const int m_nl = for_each_bound(K.p, [&](int m, int kind, int j, double sg, double val) {
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
    double hss[3], hmm[3];
    cons_eval(K.p, K.T, eps, S.xp, S.gv, S.gs, S.gn, S.gm, hss, hmm);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      int mm = m_nl + q;
      double t = PL(W.T, mm, k, N), nu = PL(W.NU, mm, k, N), it = 1.0 / t;
      double Sg = nu * it, s0 = nu * (S.gv[q] + t) * it;
      double g3[3] = {S.gs[q], S.gn[q], S.gm[q]};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        S.gxp0[a] += s0 * g3[a], S.gxp1[a] += it * g3[a];
        if (WITH_DUAL) S.dxd[a] += nu * g3[a];
#pragma unroll
        for (int c = 0; c <= a; c++) S.Hxp[sidx(a, c)] += Sg * g3[a] * g3[c];
      }
      S.Hxp[sidx(0, 0)] += nu * hss[q];
      S.Hxp[sidx(2, 2)] += nu * hmm[q];
      if (WITH_DUAL) {
        S.rp_ineq = fmax(S.rp_ineq, fabs(S.gv[q] + t));
        S.cmax = fmax(S.cmax, t * nu), S.cmin = fmin(S.cmin, t * nu), S.smult += fabs(nu);
        S.th_ineq += fabs(S.gv[q] + t), lprod *= t;
        if ((mm & 7) == 7) S.sumlog += log(lprod), lprod = 1.0;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 3; q++) S.gv[q] = -1.0, S.gs[q] = S.gn[q] = S.gm[q] = 0.0;
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

__device__ __forceinline__ bool condense_slot(const Consts& K, const Slot& S, M8Blocks& M, double* Y, double* AB) {
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
  const bool ok = inv33(maa, M.iaa) & inv33(mbb, M.ibb);
#pragma unroll
  for (int q = 0; q < 88; q++) Y[q] = 0.0, AB[q] = 0.0;
  condense_column<0>(K, S, M, Y, AB), condense_column<1>(K, S, M, Y, AB), condense_column<2>(K, S, M, Y, AB);
  condense_column<3>(K, S, M, Y, AB), condense_column<4>(K, S, M, Y, AB), condense_column<5>(K, S, M, Y, AB);
  condense_column<6>(K, S, M, Y, AB), condense_column<7>(K, S, M, Y, AB), condense_column<8>(K, S, M, Y, AB);
  condense_column<9>(K, S, M, Y, AB), condense_column<10>(K, S, M, Y, AB);
  return ok;
}

// ------------------------------------------------------------------------------------------ k_init
// Cold: do_mpc set_initial_guess (every state slot = x0, inputs 0, multipliers 0).  Warm: keep the previous
// primal/dual solution un-shifted (do_mpc), node 0 := new x0.  Slacks t = max(-h, bound_push), nu = mu/t.
__global__ void k_init(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, int cold) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int b = tid % W.Bp, k = tid / W.Bp;
  const int N = W.N;
  if (k >= N || b >= W.B) return;
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
  const bool after_failure = !cold && K.o.warm_reset_on_fail && prev != LTOMPC_STATUS_SOLVED && prev != LTOMPC_STATUS_ACCEPTABLE;
  if (after_failure) {
#pragma unroll
    for (int i = 0; i < 8; i++) PL(W.L1, i, k, N) = 0.0, PL(W.L2, i, k, N) = 0.0;
  }
  double xp[8], c[8], u[2];
#pragma unroll
  for (int i = 0; i < 8; i++) xp[i] = cold ? x0[i] : PL(W.X, i, k + 1, N + 1), c[i] = cold ? x0[i] : PL(W.C, i, k, N);
  u[0] = cold ? 0.0 : PL(W.U, 0, k, N), u[1] = cold ? 0.0 : PL(W.U, 1, k, N);
  const double mu = (!cold && !after_failure && K.o.mu_init_warm > 0) ? K.o.mu_init_warm : K.o.mu_init;
  const double eps = (K.o.smooth_scale > 0 || K.o.smooth_eps_min > 0) ? fmax(K.o.smooth_eps_min, K.o.smooth_scale * mu) : 0.0;
  // (flat visitor, no nested by-reference lambdas: see d_expand)
  const int m = for_each_bound(K.p, [&](int mm, int kind, int j, double sg, double val) {
    const double xv = kind == 0 ? u[j] : (kind == 1 ? c[j] : xp[j]);
    const double hv = sg * (xv - val);
    const double t = -hv > K.o.bound_push ? -hv : K.o.bound_push;
    PL(W.T, mm, k, N) = t, PL(W.NU, mm, k, N) = mu / t;
  });
  double gv[3] = {-1.0, -1.0, -1.0};
  if (k + 1 <= N - 1) cons_eval(K.p, K.T, eps, xp, gv, nullptr, nullptr, nullptr, nullptr, nullptr);
  for (int q = 0; q < 3; q++) {
    const double t = -gv[q] > K.o.bound_push ? -gv[q] : K.o.bound_push;
    PL(W.T, m + q, k, N) = t, PL(W.NU, m + q, k, N) = mu / t;
  }
  if (k == 0) {
    double* st = W.st;
    st[(size_t)ST_MU * W.Bp + b] = mu, st[(size_t)ST_EPS * W.Bp + b] = eps, st[(size_t)ST_EPS_NEXT * W.Bp + b] = eps;
    st[(size_t)ST_DW_LAST * W.Bp + b] = 0.0, st[(size_t)ST_FORCE_REG * W.Bp + b] = 0.0;
    st[(size_t)ST_ALPHA * W.Bp + b] = 0.0, st[(size_t)ST_ADUA * W.Bp + b] = 0.0;
    st[(size_t)ST_E0 * W.Bp + b] = 1e300, st[(size_t)ST_OBJ * W.Bp + b] = 0.0, st[(size_t)ST_TAU * W.Bp + b] = 0.99;
    st[(size_t)ST_THETA0 * W.Bp + b] = -1.0, st[(size_t)ST_THMAX * W.Bp + b] = 0.0, st[(size_t)ST_THMIN * W.Bp + b] = 0.0;
    st[(size_t)ST_DW * W.Bp + b] = 0.0, st[(size_t)ST_DW_TRY * W.Bp + b] = 0.0;
    st[(size_t)ST_C00 * W.Bp + b] = cost_eval(K.p, K.T, eps, x0, false, nullptr, nullptr);
    for (int i = 0; i < SI_NF; i++)
      if (i != SI_PREV) W.si[(size_t)i * W.Bp + b] = 0;
    W.si[(size_t)SI_STATUS * W.Bp + b] = LTOMPC_STATUS_MAX_ITER;
  }
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

// ------------------------------------------------------------------------------------------ k_eval
__device__ __forceinline__ void d_eval(const Consts& K, const Work& W, const int k, const int b) {
  const int N = W.N;
  if (W.si[(size_t)SI_DONE * W.Bp + b]) return;
  if (W.si[(size_t)SI_RETRY * W.Bp + b] || W.si[(size_t)SI_SKIP_EVAL * W.Bp + b]) return;  // blocks of the last launch are still valid
  const double hdt = K.o.t_step;
  const double eps = W.st[(size_t)ST_EPS * W.Bp + b];
  Slot S;
  linearise_slot<true>(K, W, k, b, eps, S);
  // ---- residual partials (IPOPT's E_mu ingredients) ----
  {
    double l1[8], l2[8], rd = 0.0, rp = 0.0, sm = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) l1[i] = PL(W.L1, i, k, N), l2[i] = PL(W.L2, i, k, N);
#pragma unroll
    for (int a = 0; a < 8; a++) {
      double rcx = S.dcd[a] + 4.5 * l2[a];
      double rxp = S.dxd[a] - 0.5 * l1[a];
#pragma unroll
      for (int i = 0; i <= ge_(a); i++)  // column a of E1 / E2: rows of the groups up to a's
        if (a >= elo_(i) && a <= ehi_(i)) rcx += S.E1[i * 8 + a] * l1[i], rxp += S.E2[i * 8 + a] * l2[i];
      if (k + 1 < N) rxp += 2.0 * PL(W.L1, a, k + 1, N) - 2.0 * PL(W.L2, a, k + 1, N);
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
#pragma unroll
  for (int i = 0; i < 36; i++) PG(W.QP, QP_Qx + i, k + 1, QP_NF) = S.Hxp[i];
#pragma unroll
  for (int i = 0; i < 8; i++) PG(W.QP, QP_qx0 + i, k + 1, QP_NF) = S.gxp0[i], PG(W.QP, QP_qx1 + i, k + 1, QP_NF) = S.gxp1[i];
}

__global__ void __launch_bounds__(64) k_eval(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int j = tid % la.n_pad, k = tid / la.n_pad;
  if (k >= W.N || j >= la.nact[0]) return;
  d_eval(K, W, k, la.act[j]);
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
#define STD(f) st[(size_t)(f) * W.Bp + b]
#define STI(f) si[(size_t)(f) * W.Bp + b]
  if (STI(SI_DONE)) return;
  const ltompc_options& o = K.o;
  // ---- reduce residual partials, KKT error, termination (IPOPT eq. (5),(6)) ----
  double rd = 0.0, rp = 0.0, cmax = 0.0, cmin = 1e300, smult = 0.0;
  double obj;
  obj = STD(ST_C00);  // lterm(x_0), kept by k_init / d_pick
  for (int k = 0; k < N; k++) {
    rd = fmax(rd, PL(W.RS, RS_rd, k, N)), rp = fmax(rp, PL(W.RS, RS_rp, k, N));
    cmax = fmax(cmax, PL(W.RS, RS_cmax, k, N)), cmin = fmin(cmin, PL(W.RS, RS_cmin, k, N));
    smult += PL(W.RS, RS_smult, k, N), obj += PL(W.RS, RS_cost, k, N);
  }
  const int n_mult = N * (2 * NX + K.bd.ni) - 3;  // multipliers counted (last slot has no nl constraints)
  double mu = STD(ST_MU);
  double s_d = fmax(o.s_max, smult / n_mult) / o.s_max;
  double E0 = fmax(fmax(rd / s_d, rp), cmax / s_d);
  double rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
  double Emu = fmax(fmax(rd / s_d, rp), rcmu / s_d);
  STD(ST_E0) = E0, STD(ST_OBJ) = obj;
  int iters = STI(SI_ITERS);
  int term = -1;
  if (!isfinite(E0)) term = LTOMPC_STATUS_NUMERICAL;
  else if (E0 <= o.tol) term = LTOMPC_STATUS_SOLVED;
  else {
    if (E0 <= o.acceptable_tol) {
      int na = STI(SI_NACC) + 1;
      STI(SI_NACC) = na;
      if (na >= o.acceptable_iter) term = LTOMPC_STATUS_ACCEPTABLE;
    } else STI(SI_NACC) = 0;
    if (term < 0 && iters >= o.max_iter) term = LTOMPC_STATUS_MAX_ITER;
  }
  if (term >= 0) {
    STI(SI_STATUS) = term, STI(SI_DONE) = 1;
    return;
  }
  atomicAdd(&W.active[it_index], 1);
  // ---- monotone barrier update (IPOPT eq. (7)) ----
  bool mu_changed = false;
  while (Emu <= o.kappa_eps * mu && mu > o.mu_min) {
    mu = fmax(o.mu_min, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
    mu_changed = true;
    rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
    Emu = fmax(fmax(rd / s_d, rp), rcmu / s_d);
  }
  if (mu_changed) {
    STD(ST_MU) = mu;
    STD(ST_EPS_NEXT) = (o.smooth_scale > 0 || o.smooth_eps_min > 0) ? fmax(o.smooth_eps_min, o.smooth_scale * mu) : 0.0;
    STI(SI_NFILT) = 0, STD(ST_THETA0) = -1.0;
  }
  STD(ST_TAU) = fmax(o.tau_min, 1.0 - mu);
  // ---- backward sweep, retried with Hessian regularisation until every Huu is positive definite ----
  const double r2[2] = {2.0 * K.p.r_du[0], 2.0 * K.p.r_du[1]};
  double delta_w = STD(ST_FORCE_REG);
  const double dw_last = STD(ST_DW_LAST);
  if (delta_w == 0.0 && dw_last > DW_KEEP) delta_w = dw_last / 3.0;  // see DESIGN.md §3 (deviation from Algorithm IC)
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
      double Hi[4] = {Huu[3] / det, -Huu[1] / det, -Huu[2] / det, Huu[0] / det};
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
    if (delta_w == 0.0) delta_w = dw_last == 0.0 ? o.delta_w_first : fmax(1e-20, dw_last / 3.0);
    else delta_w *= (dw_last == 0.0 ? 100.0 : 8.0);
    STI(SI_NREG) += 1;
    if (++tries > 40 || delta_w > 1e20) {
      numerical = true;
      break;
    }
  }
  if (numerical) {
    STI(SI_STATUS) = LTOMPC_STATUS_NUMERICAL, STI(SI_DONE) = 1;
    return;
  }
  if (delta_w > 0.0) STD(ST_DW_LAST) = delta_w > DW_KEEP ? delta_w : 0.0;
  else if (dw_last <= DW_KEEP) STD(ST_DW_LAST) = 0.0;
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
#undef STD
#undef STI

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

__device__ __forceinline__ double grp_max(double v) {  // over the 8 lanes of an instance (lane stride 8)
  v = fmax(v, __shfl_xor(v, 8)), v = fmax(v, __shfl_xor(v, 16)), v = fmax(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ double grp_sum(double v) {
  v += __shfl_xor(v, 8), v += __shfl_xor(v, 16), v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double grp_min(double v) {
  v = fmin(v, __shfl_xor(v, 8)), v = fmin(v, __shfl_xor(v, 16)), v = fmin(v, __shfl_xor(v, 32));
  return v;
}

// In a one-wavefront workgroup LDS instructions execute in program order, so exchanging data through LDS needs
// no s_barrier and, unlike __syncthreads(), must not drain the outstanding global loads/stores (vmcnt): only the
// compiler has to keep the LDS accesses in order.
#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

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
#define STD(f) st[(size_t)(f) * W.Bp + b]
#define STI(f) si[(size_t)(f) * W.Bp + b]
  const ltompc_options& o = K.o;
  bool live = valid && !STI(SI_DONE);
  if (!__any(live)) return;
  // One sweep per launch: an instance whose sweep fails the inertia test repeats it in the NEXT launch with a larger
  // delta_w (its blocks stay in HBM, k_eval skips it) instead of looping here, so that a launch never takes longer
  // than one sweep however hard the worst instance of the batch is.
  const bool retry = live && STI(SI_RETRY);
  // ---- residual partials: lane i reduces k = i, i+8, ...; the sum over k is done in the order k = 0..N-1 by
  //      every lane (identical to the serial kernel, so that both produce the same bits)
  double rd = 0.0, rp = 0.0, cmax = 0.0, cmin = 1e300;
  for (int k = i; k < N; k += 8) {
    rd = fmax(rd, PL(W.RS, RS_rd, k, N)), rp = fmax(rp, PL(W.RS, RS_rp, k, N));
    cmax = fmax(cmax, PL(W.RS, RS_cmax, k, N)), cmin = fmin(cmin, PL(W.RS, RS_cmin, k, N));
  }
  rd = grp_max(rd), rp = grp_max(rp), cmax = grp_max(cmax), cmin = grp_min(cmin);
  double smult = 0.0, obj;
  obj = STD(ST_C00);  // lterm(x_0), kept by k_init / d_pick
  for (int k = 0; k < N; k++) smult += PL(W.RS, RS_smult, k, N), obj += PL(W.RS, RS_cost, k, N);
  const int n_mult = N * (2 * NX + K.bd.ni) - 3;
  double mu = STD(ST_MU);
  double s_d = fmax(o.s_max, smult / n_mult) / o.s_max;
  double E0 = fmax(fmax(rd / s_d, rp), cmax / s_d);
  double rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
  double Emu = fmax(fmax(rd / s_d, rp), rcmu / s_d);
  int term = -1;
  if (live && !retry) {
    int iters = STI(SI_ITERS);
    if (!isfinite(E0)) term = LTOMPC_STATUS_NUMERICAL;
    else if (E0 <= o.tol) term = LTOMPC_STATUS_SOLVED;
    else {
      int na = (E0 <= o.acceptable_tol) ? STI(SI_NACC) + 1 : 0;
      if (i == 0) STI(SI_NACC) = na;
      if (na >= o.acceptable_iter && E0 <= o.acceptable_tol) term = LTOMPC_STATUS_ACCEPTABLE;
      if (term < 0 && iters >= o.max_iter) term = LTOMPC_STATUS_MAX_ITER;
    }
    if (i == 0) {
      STD(ST_E0) = E0, STD(ST_OBJ) = obj;
      if (term >= 0) STI(SI_STATUS) = term, STI(SI_DONE) = 1;
    }
    if (term >= 0) live = false;
  }
  if (live && i == 0 && active_slot >= 0) atomicAdd(&W.active[active_slot], 1);
  if (!__any(live)) return;
  // ---- monotone barrier update
  bool mu_changed = false;
  while (live && !retry && Emu <= o.kappa_eps * mu && mu > o.mu_min) {
    mu = fmax(o.mu_min, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
    mu_changed = true;
    rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
    Emu = fmax(fmax(rd / s_d, rp), rcmu / s_d);
  }
  if (live && !retry && i == 0) {
    if (mu_changed) {
      STD(ST_MU) = mu;
      STD(ST_EPS_NEXT) = (o.smooth_scale > 0 || o.smooth_eps_min > 0) ? fmax(o.smooth_eps_min, o.smooth_scale * mu) : 0.0;
      STI(SI_NFILT) = 0, STD(ST_THETA0) = -1.0;
    }
    STD(ST_TAU) = fmax(o.tau_min, 1.0 - mu);
  }
  // ---- backward sweep (whole wave in lock-step; an instance whose Huu fails retries with a larger delta_w,
  //      the others recompute the same numbers)
  const double r2[2] = {2.0 * K.p.r_du[0], 2.0 * K.p.r_du[1]};
  double delta_w = STD(ST_FORCE_REG);
  const double dw_last = STD(ST_DW_LAST);
  if (delta_w == 0.0 && dw_last > DW_KEEP) delta_w = dw_last / 3.0;  // see DESIGN.md §3 (deviation from Algorithm IC)
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
      // 1. row i of P A, P B, P b + p
      double PAr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, PBr[2] = {0, 0}, Pbi = ppi;
#pragma unroll
      for (int l = 0; l < 8; l++) {
#pragma unroll
        for (int j = 0; j < 8; j++) PAr[j] += Prow[l] * LA(l * 8 + j);
        PBr[0] += Prow[l] * LB(l * 2), PBr[1] += Prow[l] * LB(l * 2 + 1);
        Pbi += Prow[l] * Lb(l);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) L.PA[i * 8 + j][g] = PAr[j];
      L.PB[i * 2][g] = PBr[0], L.PB[i * 2 + 1][g] = PBr[1], L.Pb[i][g] = Pbi;
      WAVE_SYNC();
      // 2. row i of Hxx = Q + A^T (P A), of Hux^T, and gx_i
      double Hxx[8], Hxu[2], gx = qi;
#pragma unroll
      for (int j = 0; j < 8; j++) Hxx[j] = Qrow[j];
      Hxu[0] = Scol[0], Hxu[1] = Scol[1];
#pragma unroll
      for (int l = 0; l < 8; l++) {
        double ali = LA(l * 8 + i);
#pragma unroll
        for (int j = 0; j < 8; j++) Hxx[j] += ali * L.PA[l * 8 + j][g];
        double pali = L.PA[l * 8 + i][g];
        Hxu[0] += LB(l * 2) * pali + L.Pxv[l * 2][g] * ali;
        Hxu[1] += LB(l * 2 + 1) * pali + L.Pxv[l * 2 + 1][g] * ali;
        gx += ali * L.Pb[l][g];
      }
      // (the next stage block is requested here: its 27 registers per lane are live for half a stage only)
      {
        const double* src = &PG(W.QP, i, kn, QP_NF);
#pragma unroll
        for (int j = 0; j < NPF; j++) pf[j] = src[j * 64];
      }
      // 3. Huu, gu (same numbers in the 8 lanes of an instance)
      double Huu[4], gu[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int d = 0; d < 2; d++) {
          double s = Rm[sidx(c, d)] + Pvv[c * 2 + d];
#pragma unroll
          for (int l = 0; l < 8; l++)
            s += LB(l * 2 + c) * L.PB[l * 2 + d][g] + LB(l * 2 + c) * L.Pxv[l * 2 + d][g] + L.Pxv[l * 2 + c][g] * LB(l * 2 + d);
          Huu[c * 2 + d] = s;
        }
        Huu[c * 2 + c] += r2[c] + delta_w;
        double s = rr[c] + r2[c] * (uk[c] - vk[c]) + pv[c];
#pragma unroll
        for (int l = 0; l < 8; l++) s += LB(l * 2 + c) * L.Pb[l][g] + L.Pxv[l * 2 + c][g] * Lb(l);
        gu[c] = s;
      }
#undef LA
#undef LB
#undef Lb
      double det = Huu[0] * Huu[3] - Huu[1] * Huu[2];
      bool bad = !(Huu[0] > 0.0) || !(det > 1e-14 * Huu[0] * Huu[3]) || !isfinite(det);
      if (bad && live) ok = false;
      if (bad) det = 1.0, Huu[0] = Huu[3] = 1.0, Huu[1] = Huu[2] = 0.0;  // keep the lock-step arithmetic finite
      double Hi[4] = {Huu[3] / det, -Huu[1] / det, -Huu[2] / det, Huu[0] / det};
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
#pragma unroll
      for (int j = 0; j < 8; j++) Prow[j] = (j == i) ? Pn[j] : 0.5 * (Pn[j] + L.PA[j * 8 + i][g]);
      L.Pxv[i * 2][g] = pxv[0], L.Pxv[i * 2 + 1][g] = pxv[1];  // read after the next stage's first barrier
      // the stage block fetched above goes to the other buffer (last read one stage ago), BEFORE this stage's stores
      // are issued: waiting for the loads then does not wait for the stores
#pragma unroll
      for (int j = 0; j < NPF; j++) L.sb[(k & 1) ^ 1][(i + 8 * j) * 8 + g] = pf[j];
      uk[0] = vk[0], uk[1] = vk[1];  // u_{k-1} is the v of stage k
      vk[0] = k > 1 ? vn0 : up0, vk[1] = k > 1 ? vn1 : up1;
      if (live) {
        PG(W.RC, RC_K + i, k, RC_NF) = Kc[0], PG(W.RC, RC_K + 8 + i, k, RC_NF) = Kc[1];
        if (i < 4) PG(W.RC, RC_Kv + i, k, RC_NF) = Kv[i];
        if (i < 2) PG(W.RC, RC_kff + i, k, RC_NF) = kff[i];
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
      if (delta_w == 0.0) delta_w = dw_last == 0.0 ? o.delta_w_first : fmax(1e-20, dw_last / 3.0);
      else delta_w *= (dw_last == 0.0 ? 100.0 : 8.0);
      if (++tries > 40 || delta_w > 1e20) numerical = true;
      if (i == 0) STI(SI_NREG) += 1;
    }
    const bool again = failed && !numerical && sweep + 1 < max_sweeps;
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
    STD(ST_DW_LAST) = delta_w > DW_KEEP ? delta_w : 0.0;
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
#undef STD
#undef STI
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
#define STD(f) st[(size_t)(f) * W.Bp + b]
#define STI(f) si[(size_t)(f) * W.Bp + b]
  const ltompc_options& o = K.o;
  // LTOMPC_DBG: shader-clock cycles of block 0 per section (head, staging, backward sweeps, forward), summed over launches
  const bool rprof = W.DBG != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  long long rt0 = rprof ? clock64() : 0;
#define RTOCK(q) if (rprof) { const long long t1 = clock64(); W.DBG[q] += (double)(t1 - rt0); rt0 = t1; }
  bool live = valid && !STI(SI_DONE);
  if (!__any(live)) return;
  // One sweep per launch: an instance whose sweep fails the inertia test repeats it in the NEXT launch with a larger
  // delta_w (its blocks stay in HBM, k_eval skips it) instead of looping here, so that a launch never takes longer
  // than one sweep however hard the worst instance of the batch is.
  const bool retry = live && STI(SI_RETRY);
  // ---- residual partials: lane i reduces k = i, i+8, ...; the sum over k is done in the order k = 0..N-1 by
  //      every lane (identical to the serial kernel, so that both produce the same bits)
  double rd = 0.0, rp = 0.0, cmax = 0.0, cmin = 1e300;
  for (int k = i; k < N; k += 8) {
    rd = fmax(rd, PL(W.RS, RS_rd, k, N)), rp = fmax(rp, PL(W.RS, RS_rp, k, N));
    cmax = fmax(cmax, PL(W.RS, RS_cmax, k, N)), cmin = fmin(cmin, PL(W.RS, RS_cmin, k, N));
  }
  rd = grp_max(rd), rp = grp_max(rp), cmax = grp_max(cmax), cmin = grp_min(cmin);
  double smult = 0.0, obj;
  obj = STD(ST_C00);  // lterm(x_0), kept by k_init / d_pick
  for (int k = 0; k < N; k++) smult += PL(W.RS, RS_smult, k, N), obj += PL(W.RS, RS_cost, k, N);
  const int n_mult = N * (2 * NX + K.bd.ni) - 3;
  double mu = STD(ST_MU);
  double s_d = fmax(o.s_max, smult / n_mult) / o.s_max;
  double E0 = fmax(fmax(rd / s_d, rp), cmax / s_d);
  double rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
  double Emu = fmax(fmax(rd / s_d, rp), rcmu / s_d);
  int term = -1;
  if (live && !retry) {
    int iters = STI(SI_ITERS);
    if (!isfinite(E0)) term = LTOMPC_STATUS_NUMERICAL;
    else if (E0 <= o.tol) term = LTOMPC_STATUS_SOLVED;
    else {
      int na = (E0 <= o.acceptable_tol) ? STI(SI_NACC) + 1 : 0;
      if (i == 0) STI(SI_NACC) = na;
      if (na >= o.acceptable_iter && E0 <= o.acceptable_tol) term = LTOMPC_STATUS_ACCEPTABLE;
      if (term < 0 && iters >= o.max_iter) term = LTOMPC_STATUS_MAX_ITER;
    }
    if (i == 0) {
      STD(ST_E0) = E0, STD(ST_OBJ) = obj;
      if (term >= 0) STI(SI_STATUS) = term, STI(SI_DONE) = 1;
    }
    if (term >= 0) live = false;
  }
  if (live && i == 0 && active_slot >= 0) atomicAdd(&W.active[active_slot], 1);
  if (!__any(live)) return;
  // ---- monotone barrier update
  bool mu_changed = false;
  while (live && !retry && Emu <= o.kappa_eps * mu && mu > o.mu_min) {
    mu = fmax(o.mu_min, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
    mu_changed = true;
    rcmu = fmax(fabs(cmax - mu), fabs(cmin - mu));
    Emu = fmax(fmax(rd / s_d, rp), rcmu / s_d);
  }
  if (live && !retry && i == 0) {
    if (mu_changed) {
      STD(ST_MU) = mu;
      STD(ST_EPS_NEXT) = (o.smooth_scale > 0 || o.smooth_eps_min > 0) ? fmax(o.smooth_eps_min, o.smooth_scale * mu) : 0.0;
      STI(SI_NFILT) = 0, STD(ST_THETA0) = -1.0;
    }
    STD(ST_TAU) = fmax(o.tau_min, 1.0 - mu);
  }
  // ---- backward sweep (whole wave in lock-step; an instance whose Huu fails retries with a larger delta_w,
  //      the others recompute the same numbers)
  const double r2[2] = {2.0 * K.p.r_du[0], 2.0 * K.p.r_du[1]};
  double delta_w = STD(ST_FORCE_REG);
  const double dw_last = STD(ST_DW_LAST);
  if (delta_w == 0.0 && dw_last > DW_KEEP) delta_w = dw_last / 3.0;  // see DESIGN.md §3 (deviation from Algorithm IC)
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
    const int total = N * QP_NF;
    for (int base = lane; base < total; base += 64 * 8) {  // 8 independent loads in flight per lane
      double v[8];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int idx = base + 64 * r, ic = idx < total ? idx : total - 1;
        const int kq = ic / QP_NF, fq = ic - kq * QP_NF;
        v[r] = PG(W.QP, fq, kq, QP_NF);
      }
#pragma unroll
      for (int r = 0; r < 8; r++)
        if (base + 64 * r < total) S.q[base + 64 * r] = v[r];
    }
    for (int idx = lane; idx < N * 2; idx += 64) S.u[idx] = PL(W.U, idx & 1, idx >> 1, N);
  }
  WAVE_SYNC();
  RTOCK(1);
  // (read once: a global load inside the stage loop would wait, on vmcnt, for the RC stores of the previous stage)
  const double up0 = W.uprev[b], up1 = W.uprev[(size_t)W.Bp + b];
  mu = __shfl(mu, 8 * i);  // the column lanes (g > 0) do real work here: give them the live lane's barrier parameter
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
      // 2. element (i, g) of Hxx, row i of Hux^T, gx_i
      double Ai[8], PAg[8], PAi[8], B0[8], B1[8], X0[8], X1[8], Pbv[8];
#pragma unroll
      for (int l = 0; l < 8; l++) {
        Ai[l] = qk[QP_A + l * 8 + i], PAg[l] = L.PA[l * 8 + g], PAi[l] = L.PA[l * 8 + i];
        B0[l] = qk[QP_B + l * 2], B1[l] = qk[QP_B + l * 2 + 1], X0[l] = L.Pxv[l * 2], X1[l] = L.Pxv[l * 2 + 1];
        Pbv[l] = L.Pb[l];
      }
      double hxx = cur.q_elem, Hxu[2] = {cur.S[0], cur.S[1]}, gx = cur.q;
#pragma unroll
      for (int l = 0; l < 8; l++) {
        double ali = Ai[l];
        hxx += ali * PAg[l];
        double pali = PAi[l];
        Hxu[0] += B0[l] * pali + X0[l] * ali;
        Hxu[1] += B1[l] * pali + X1[l] * ali;
        gx += ali * Pbv[l];
      }
      // 3. Huu, gu: one element per lane (g = 0..3: Huu[g>>1][g&1], g = 4, 5: gu[g-4]), gathered with wave shuffles.
      //    Both sums are formed by every lane with selected operands (no divergent branches, no run-time indices into
      //    register arrays: those would live in scratch, and a scratch reload waits for the RC stores of the stage before)
      double he;
      {
        const bool c1 = (g >> 1) & 1, d1 = g & 1;
        const double rm = (c1 && d1) ? Rm[2] : ((c1 || d1) ? Rm[1] : Rm[0]);     // Rm[sidx(c, d)]
        const double pvv = c1 ? (d1 ? Pvv[3] : Pvv[2]) : (d1 ? Pvv[1] : Pvv[0]);  // Pvv[c * 2 + d]
        double PBd[8];
#pragma unroll
        for (int l = 0; l < 8; l++) PBd[l] = L.PB[l * 2 + (g & 1)];
        double s = rm + pvv;
#pragma unroll
        for (int l = 0; l < 8; l++) {
          const double Bc = c1 ? B1[l] : B0[l], Bd = d1 ? B1[l] : B0[l], Xc = c1 ? X1[l] : X0[l], Xd = d1 ? X1[l] : X0[l];
          s += Bc * PBd[l] + Bc * Xd + Xc * Bd;
        }
        double heH = s;
        if (c1 == d1) heH += (c1 ? r2[1] : r2[0]) + delta_w;
        // gu[c], c = g & 1
        double sg = (d1 ? rr[1] : rr[0]) + (d1 ? r2[1] : r2[0]) * ((d1 ? uk[1] : uk[0]) - (d1 ? vk[1] : vk[0])) + (d1 ? pv[1] : pv[0]);
#pragma unroll
        for (int l = 0; l < 8; l++) {
          const double Bc = d1 ? B1[l] : B0[l], Xc = d1 ? X1[l] : X0[l];
          sg += Bc * Pbv[l] + Xc * bl[l];
        }
        he = g < 4 ? heH : (g < 6 ? sg : 0.0);
      }
      double Huu[4], gu[2];
#pragma unroll
      for (int q = 0; q < 4; q++) Huu[q] = __shfl(he, q + 8 * i);
      gu[0] = __shfl(he, 4 + 8 * i), gu[1] = __shfl(he, 5 + 8 * i);
      double det = Huu[0] * Huu[3] - Huu[1] * Huu[2];
      bool bad = !(Huu[0] > 0.0) || !(det > 1e-14 * Huu[0] * Huu[3]) || !isfinite(det);
      if (bad && live) ok = false;
      if (bad) det = 1.0, Huu[0] = Huu[3] = 1.0, Huu[1] = Huu[2] = 0.0;
      double Hi[4] = {Huu[3] / det, -Huu[1] / det, -Huu[2] / det, Huu[0] / det};
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
#pragma unroll
      for (int j = 0; j < 8; j++) Prow[j] = (j == i) ? L.P[i * 8 + j] : 0.5 * (L.P[i * 8 + j] + L.P[j * 8 + i]);
      L.Pxv[i * 2] = pxv[0], L.Pxv[i * 2 + 1] = pxv[1];
      if (g == 0) {  // the gains stay in LDS for the forward rollout (same numbers in all column lanes)
        S.kk[k * 22 + i] = Kc[0], S.kk[k * 22 + 8 + i] = Kc[1];
        if (i < 4) S.kk[k * 22 + 16 + i] = Kv[i];
        if (i < 2) S.kk[k * 22 + 20 + i] = kff[i];
      }
      if (live) {
        PG(W.RC, RC_K + i, k, RC_NF) = Kc[0], PG(W.RC, RC_K + 8 + i, k, RC_NF) = Kc[1];
        if (i < 4) PG(W.RC, RC_Kv + i, k, RC_NF) = Kv[i];
        if (i < 2) PG(W.RC, RC_kff + i, k, RC_NF) = kff[i];
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
      if (delta_w == 0.0) delta_w = dw_last == 0.0 ? o.delta_w_first : fmax(1e-20, dw_last / 3.0);
      else delta_w *= (dw_last == 0.0 ? 100.0 : 8.0);
      if (++tries > 40 || delta_w > 1e20) numerical = true;
      if (i == 0) STI(SI_NREG) += 1;
    }
    const bool again = failed && !numerical && sweep + 1 < max_sweeps;
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
    STD(ST_DW_LAST) = delta_w > DW_KEEP ? delta_w : 0.0;
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
#undef STD
#undef STI
}

__global__ void __launch_bounds__(64) k_riccati8(Consts K, Work W, Launch la, int it_index, int max_sweeps) {
  __shared__ RicLds L;
  const int lane = threadIdx.x, g = lane & 7, i = lane >> 3;
  const int jj = blockIdx.x * 8 + g;
  const bool valid = jj < la.nact[0];
  d_riccati8(K, W, L, g, i, la.act[valid ? jj : 0], valid, it_index, max_sweeps);
}

// One wavefront per instance (narrow launches: once few instances are left, a launch is as long as one wavefront's
// sweep, and 8 instances per wavefront make that sweep ~3x longer than it has to be).  Dynamic LDS: ric1_lds_bytes(N).
__global__ void __launch_bounds__(64) k_riccati1(Consts K, Work W, Launch la, int it_index) {
  extern __shared__ double lds1[];
  const int lane = threadIdx.x, g = lane & 7, i = lane >> 3;
  if ((int)blockIdx.x >= la.nact[0]) return;
  const int N = W.N;
  StageLds S{lds1, lds1 + (size_t)N * QP_NF, lds1 + (size_t)N * (QP_NF + 2)};
  Ric1Lds& L = *reinterpret_cast<Ric1Lds*>(lds1 + (size_t)N * (QP_NF + 24));
  d_riccati1(K, W, L, S, g, i, la.act[blockIdx.x], g == 0, it_index, 1);
}

// ------------------------------------------------------------------------------------------ k_expand
__device__ __forceinline__ void d_expand(const Consts& K, const Work& W, const int k, const int b) {
  const int N = W.N;
  if (W.si[(size_t)SI_DONE * W.Bp + b] || !W.si[(size_t)SI_STEP * W.Bp + b]) return;  // no step this launch
  const double mu = W.st[(size_t)ST_MU * W.Bp + b], eps = W.st[(size_t)ST_EPS * W.Bp + b];
  const double tau = W.st[(size_t)ST_TAU * W.Bp + b];
  Slot S;
  linearise_slot<false>(K, W, k, b, eps, S);
  M8Blocks M8;
  double Y[88], AB[88];
  condense_slot(K, S, M8, Y, AB);
  double dxk[8], dxp[8], du[2], dc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) dxk[i] = PL(W.dX, i, k, N + 1), dxp[i] = PL(W.dX, i, k + 1, N + 1);
  du[0] = PL(W.dU, 0, k, N), du[1] = PL(W.dU, 1, k, N);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double s = Y[i * 11 + 10] + Y[i * 11 + 8] * du[0] + Y[i * 11 + 9] * du[1];
#pragma unroll
    for (int j = gs_(i); j < 8; j++)  // row i of Ac: block upper triangular, its (delta, T) block is the identity
      if (i < 6 || j == i) s += Y[i * 11 + j] * dxk[j];
    dc[i] = s;
    PL(W.dC, i, k, N) = s;
  }
  // costate pi_{k+1} = P_{k+1} dx_{k+1} + Pxv_{k+1} du_k + p_{k+1}
  double pi[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double s = PG(W.RC, RC_pp + i, k + 1, RC_NF) + PG(W.RC, RC_Pxv + i * 2, k + 1, RC_NF) * du[0] +
               PG(W.RC, RC_Pxv + i * 2 + 1, k + 1, RC_NF) * du[1];
#pragma unroll
    for (int j = 0; j < 8; j++) s += PG(W.RC, RC_P + sidx(i, j), k + 1, RC_NF) * dxp[j];
    pi[i] = s;
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
  // (one flat visitor: an earlier version with a second, nested by-reference lambda produced run-to-run varying
  //  values of gphid for the last interval on ROCm 7.2 / gfx950, a code-generation problem that instrumenting stores
  //  made disappear; tests/test_gpu_parity.py::test_full_size_batch_properties guards against its return)
  for_each_bound(K.p, [&](int m, int kind, int j, double sg, double val) {
    const double xv = kind == 0 ? S.u[j] : (kind == 1 ? S.c[j] : S.xp[j]);
    const double dv = kind == 0 ? du[j] : (kind == 1 ? dc[j] : dxp[j]);
    const double t = PL(W.T, m, k, N), nu = PL(W.NU, m, k, N), it = 1.0 / t;
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
      const double t = PL(W.T, m, k, N), nu = PL(W.NU, m, k, N), it = 1.0 / t;
      const double dtt = -(S.gv[q] + t) - (S.gs[q] * dxp[0] + S.gn[q] * dxp[1] + S.gm[q] * dxp[2]);
      const double dn = (mu - nu * dtt) * it - nu;
      PL(W.dT, m, k, N) = dtt, PL(W.dNU, m, k, N) = dn;
      r_pri = fmax(r_pri, -dtt * it);
      if (dn < 0.0) a_dua = fmin(a_dua, -tau * nu / dn);
      gphid -= mu * dtt * it;
    } else {
      PL(W.dT, m, k, N) = 0.0, PL(W.dNU, m, k, N) = 0.0;
    }
  }
  const double a_pri = r_pri > tau ? tau / r_pri : 1.0;
  PL(W.SP, SP_apri, k, N) = a_pri, PL(W.SP, SP_adua, k, N) = a_dua, PL(W.SP, SP_gphid, k, N) = gphid;
}

__global__ void __launch_bounds__(64) k_expand(Consts K, Work W, Launch la) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int j = tid % la.n_pad, k = tid / la.n_pad;
  if (k >= W.N || j >= la.nact[0]) return;
  d_expand(K, W, k, la.act[j]);
}

// ------------------------------------------------------------------------------------------ k_linesearch
// candidate 0 is the current point (alpha = 0); candidate l >= 1 has alpha = a_pri * 2^-(l-1).
// LS plane layout: [3 * (n_ls + 1)][N][Bp] : theta, cost, sum log t per candidate.
// Two phases (97% of all iterations accept the full step): phase 0 evaluates the current point and the first candidate
// for every instance; phase 1 evaluates the remaining candidates for the instances whose first candidate was rejected.
// Filter measures (theta, cost, sum log t) of the step candidates l_begin..l_end of interval k of instance b;
// candidate l >= 1 has alpha = a_pri * 2^-(l-1) (l = 0, the current point, is written by k_eval).
__device__ __forceinline__ void d_linesearch(const Consts& K, const Work& W, const int k, const int b, const int l_begin,
                                             const int l_end) {
  const int N = W.N;
  const double hdt = K.o.t_step;
  if (W.si[(size_t)SI_DONE * W.Bp + b] || !W.si[(size_t)SI_STEP * W.Bp + b]) return;  // no step this launch
  const double eps = W.st[(size_t)ST_EPS * W.Bp + b];
  double a_pri = 1.0;
  for (int kk = 0; kk < N; kk++) a_pri = fmin(a_pri, PL(W.SP, SP_apri, kk, N));
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
    const int m = for_each_bound(K.p, [&](int mm, int kind, int jj, double sg, double val) {
      const double xv = kind == 0 ? tu[jj] : (kind == 1 ? tc[jj] : txp[jj]);
      const double t = PL(W.T, mm, k, N) + alpha * PL(W.dT, mm, k, N);
      th += fabs(sg * (xv - val) + t), pr *= t;
      if ((mm & 7) == 7) sl += log(pr), pr = 1.0;
    });
    if (nl) {
      double gv[3];
      cons_eval(K.p, K.T, eps, txp, gv, nullptr, nullptr, nullptr, nullptr, nullptr);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        double t = PL(W.T, m + q, k, N) + alpha * PL(W.dT, m + q, k, N);
        th += fabs(gv[q] + t), pr *= t;
        if (((m + q) & 7) == 7) sl += log(pr), pr = 1.0;
      }
    }
    sl += log(pr);
    PL(W.LS, 3 * l + 0, k, N) = th, PL(W.LS, 3 * l + 1, k, N) = co, PL(W.LS, 3 * l + 2, k, N) = sl;
  }
}

__global__ void __launch_bounds__(64, 2) k_linesearch(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la, int phase, int jw) {
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
  for (int j = j0; j < count; j += jw) d_linesearch(K, W, k, phase == 0 ? la.act[j] : W.ls_list[j], l, l);
}

// ------------------------------------------------------------------------------------------ k_pick
// Filter line search of Waechter & Biegler 2006 (no second-order correction, no restoration phase).
#define STD(f) st[(size_t)(f) * W.Bp + b]
#define STI(f) si[(size_t)(f) * W.Bp + b]
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
  for (int k = i; k < N; k += 8) {
    a_pri = fmin(a_pri, PL(W.SP, SP_apri, k, N)), a_dua = fmin(a_dua, PL(W.SP, SP_adua, k, N));
    gphid += PL(W.SP, SP_gphid, k, N);
  }
  a_pri = grp_min(a_pri), a_dua = grp_min(a_dua), gphid = grp_sum(gphid);
  // lterm(x_0) is a constant of the solve; kept so that phi matches the oracle's barrier objective
  const double c00 = STD(ST_C00);
  auto measures = [&](int l, double& th, double& ph) {
    double t = 0.0, c = 0.0, s = 0.0;
    for (int k = i; k < N; k += 8) t += PL(W.LS, 3 * l + 0, k, N), c += PL(W.LS, 3 * l + 1, k, N), s += PL(W.LS, 3 * l + 2, k, N);
    t = grp_sum(t), c = grp_sum(c), s = grp_sum(s);
    th = t, ph = (c00 + c) - mu * s;
  };
  double th0, ph0;
  measures(0, th0, ph0);
  double theta0 = STD(ST_THETA0);
  int nfilt = STI(SI_NFILT);
  double theta_max = STD(ST_THMAX), theta_min = STD(ST_THMIN);
  if (theta0 < 0.0) {
    theta0 = th0, theta_max = 1e4 * fmax(1.0, theta0), theta_min = 1e-4 * fmax(1.0, theta0);
    if (i == 0) STD(ST_THETA0) = theta0, STD(ST_THMAX) = theta_max, STD(ST_THMIN) = theta_min;
    nfilt = 0;
  }
  const double g_th = 1e-5, g_ph = 1e-8, eta_ph = 1e-8, s_th = 1.1, s_ph = 2.3, dlt = 1.0;
  bool accepted = false;
  double alpha = a_pri;
  const int n_ls = o.n_linesearch;
  const int n_try = (phase == 0) ? 1 : n_ls;  // phase 1 repeats the test of candidate 0 (same outcome) and goes on
  for (int l = 0; l < n_try; l++, alpha *= 0.5) {
    double th, ph;
    measures(l + 1, th, ph);
    if (!isfinite(th) || !isfinite(ph) || th > theta_max) continue;
    bool in_filter = false;
    for (int f = 0; f < nfilt; f++)
      if (th >= W.filt[(size_t)(2 * f) * W.Bp + b] && ph >= W.filt[(size_t)(2 * f + 1) * W.Bp + b]) {
        in_filter = true;
        break;
      }
    if (in_filter) continue;
    bool sw = (gphid < 0.0) && (alpha * pow(-gphid, s_ph) > dlt * pow(th0, s_th));
    bool armijo = ph <= ph0 + eta_ph * alpha * gphid;
    bool ok;
    if (th0 <= theta_min && sw) ok = armijo;
    else ok = (th <= (1.0 - g_th) * th0) || (ph <= ph0 - g_ph * th0);
    if (!ok) continue;
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
        W.filt[(size_t)(2 * nfilt + 1) * W.Bp + b] = ph0 - g_ph * th0;
      }
      nfilt++;
    }
    accepted = true;
    break;
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
  bool take = true, give_up = false;
  if (!accepted) {
    const int nf = STI(SI_NLSFAIL) + 1;
    STI(SI_NLSFAIL) = nf;
    double fr = STD(ST_FORCE_REG);
    if (o.max_ls_fail > 0 && nf >= o.max_ls_fail) {
      give_up = true, take = false;
    } else if (fr < 1e4) {
      STD(ST_FORCE_REG) = fr == 0.0 ? 1e-2 : fr * 100.0;
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
    if (o.stall_iter > 0 && nt >= o.stall_iter) {
      STI(SI_STATUS) = LTOMPC_STATUS_STALLED, STI(SI_DONE) = 1;
      take = false;
    }
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
  STI(SI_SKIP_EVAL) = (!take && !eps_switched) ? 1 : 0;  // the iterate did not move: the stage blocks stay valid
}

__global__ void __launch_bounds__(64) k_pick(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la, int phase) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  const int lane = threadIdx.x, g = lane & 7, i = lane >> 3;
  const int j = blockIdx.x * 8 + g;
  if (j >= (phase == 0 ? la.nact[0] : W.ls_count[0])) return;
  d_pick(K, W, phase == 0 ? la.act[j] : W.ls_list[j], i, phase, true);
}

// ------------------------------------------------------------------------------------------ k_update
__device__ __forceinline__ void d_update(const Consts& K, const Work& W, const int k, const int b) {
  const int N = W.N;
  if (!W.si[(size_t)SI_STEP * W.Bp + b] || W.si[(size_t)SI_DONE * W.Bp + b]) return;
  const double alpha = W.st[(size_t)ST_ALPHA * W.Bp + b], a_dua = W.st[(size_t)ST_ADUA * W.Bp + b];
  const double mu = W.st[(size_t)ST_MU * W.Bp + b];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    PL(W.X, i, k + 1, N + 1) += alpha * PL(W.dX, i, k + 1, N + 1);
    PL(W.C, i, k, N) += alpha * PL(W.dC, i, k, N);
    double l1 = PL(W.L1, i, k, N), l2 = PL(W.L2, i, k, N);
    PL(W.L1, i, k, N) = l1 + alpha * (PL(W.nL1, i, k, N) - l1);
    PL(W.L2, i, k, N) = l2 + alpha * (PL(W.nL2, i, k, N) - l2);
  }
  PL(W.U, 0, k, N) += alpha * PL(W.dU, 0, k, N), PL(W.U, 1, k, N) += alpha * PL(W.dU, 1, k, N);
  const int ni = K.bd.ni, nact = (k + 1 <= N - 1) ? ni : ni - 3;
  for (int m = 0; m < nact; m++) {
    double t = PL(W.T, m, k, N) + alpha * PL(W.dT, m, k, N);
    double nu = PL(W.NU, m, k, N) + a_dua * PL(W.dNU, m, k, N);
    double lo = mu / (1e10 * t), hi = 1e10 * mu / t;  // IPOPT eq. (16)
    PL(W.T, m, k, N) = t, PL(W.NU, m, k, N) = nu < lo ? lo : (nu > hi ? hi : nu);
  }
}

__global__ void __launch_bounds__(64) k_update(const Consts* __restrict__ Kp, const Work* __restrict__ Wp, Launch la) {
  const Consts& K = *Kp;  // K and W live in device memory: fields are fetched where they are used instead of
  const Work& W = *Wp;    // occupying (spilled) SGPRs for the whole kernel
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid == 0) W.ls_count[0] = 0;  // both line-search phases of this iteration are over
  int j = tid % la.n_pad, k = tid / la.n_pad;
  if (k >= W.N || j >= la.nact[0]) return;
  d_update(K, W, k, la.act[j]);
}


// ------------------------------------------------------------------------------------------ k_step1
// Narrow launches: the whole step selection of ONE instance per workgroup (both line-search phases, the filter test
// and the update), i.e. five dependent launches of 10..30 us each in one.  Same device functions, same numbers.
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
  for (int idx = tid; idx < N * K.o.n_linesearch; idx += 320) d_linesearch(K, W, idx % N, b, 1 + idx / N, 1 + idx / N);
  __syncthreads();
  if (tid < 64 && (tid & 7) == 0) d_pick(K, W, b, tid >> 3, 0, false);
  __syncthreads();
  if (si[(size_t)SI_LSMORE * W.Bp + b]) {  // block-uniform (written before the barrier)
    if (tid < 64 && (tid & 7) == 0) d_pick(K, W, b, tid >> 3, 1, false);
    __syncthreads();
  }
  for (int kk = tid; kk < N; kk += 320) d_update(K, W, kk, b);
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

// ------------------------------------------------------------------------------------------ I/O helpers
// row-major (B x 8) user buffer -> [8][Bp] planes
__global__ void k_load_x0(Work W, const double* __restrict__ x0_rm) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
#pragma unroll
  for (int i = 0; i < 8; i++) W.x0[(size_t)i * W.Bp + b] = x0_rm[(size_t)b * 8 + i];
  W.si[(size_t)SI_PREV * W.Bp + b] = W.si[(size_t)SI_STATUS * W.Bp + b];  // k_init resets the rest
}
__global__ void k_zero_uprev(Work W) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
  W.uprev[b] = 0.0, W.uprev[(size_t)W.Bp + b] = 0.0;
}
// u0 = U[:,0,:] -> row-major (B x 2) and u_prev := u0 (do_mpc: _u_prev = last returned u0)
__global__ void k_store_u0(Work W, double* __restrict__ u0_rm) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W.B) return;
  const int N = W.N;
  double a = PL(W.U, 0, 0, N), c = PL(W.U, 1, 0, N);
  if (u0_rm) u0_rm[(size_t)b * 2] = a, u0_rm[(size_t)b * 2 + 1] = c;
  W.uprev[b] = a, W.uprev[(size_t)W.Bp + b] = c;
}

// plant: classical RK4 with n_sub sub-steps, zero-order-hold input (do_mpc Simulator / CVODES stand-in)
__global__ void k_plant(Consts K, int B, const double* __restrict__ x, const double* __restrict__ u, double dt,
                        int n_sub, double* __restrict__ xn) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double y[8], uu[2] = {u[(size_t)b * 2], u[(size_t)b * 2 + 1]};
#pragma unroll
  for (int i = 0; i < 8; i++) y[i] = x[(size_t)b * 8 + i];
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

#undef STD
#undef STI
}  // namespace ltompc
