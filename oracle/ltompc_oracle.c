/*
 * ltompc_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's make_step path:
 *   model            src/mpc/model.py:101-117,124-128,152-183  (rhs, slip angles, Pacejka, slip cost)
 *   constraints      src/mpc/model.py:70-84 ; src/mpc/controller.py:57-103
 *   objective        src/mpc/controller.py:36-55 ; rterm weights src/mpc.py:104
 *   tables           src/path.py:96-101 ; src/mpc/track.py:30-42      (piece-wise linear, linear extrapolation)
 *   transcription    do_mpc 4.6.5 defaults: Radau-IIA deg-2 collocation, nl_cons at nodes 0..N-1,
 *                    bounds on nodes + collocation points, rterm on Delta-u   
 *   NLP solver       IPOPT-style primal-dual interior point (monotone mu, fraction-to-boundary, filter
 *                    line search, Hessian regularisation), restated from Waechter & Biegler 2006; the
 *                    real third-party stack (casadi 3.6.6 / do_mpc 4.6.5 / IPOPT+MUMPS) is not installable
 *                    offline, so there is no do_mpc number to compare with: make_step parity is pinned only
 *                    by the anchors in tests/ (see DESIGN.md "oracle pinning").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Deliberately different from the HIP product so that the two check each other:
 *   - every derivative here comes from forward-mode second-order jets (generic AD), the product uses
 *     hand-derived analytic derivatives;
 *   - the per-interval collocation system is solved as one 16x16 LU with partial pivoting, the product
 *     block-eliminates it to 8x8.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/ltompc.h"

#define NX 8
#define NU 2
#define NH 36
#define MAXXB 16
#define NNL 3 /* gL, gR+ , gR- : see cons_jet */
#define NEL 2 /* friction-ellipse constraints of the two axles (model.py:86-99), present when params.ell_penalty > 0 */
#define NNLT (NNL + NEL)
#define MAXI (4 + 2 * MAXXB + NNLT)

/* ------------------------------------------------------------------ jets */
typedef struct {
  double v, g[NX], h[NH];
} jet;
static inline int hidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

static jet jconst(double c) {
  jet r;
  memset(&r, 0, sizeof r);
  r.v = c;
  return r;
}
static jet jvar(double v, int i) {
  jet r = jconst(v);
  r.g[i] = 1.0;
  return r;
}
static jet jadd(jet a, jet b) {
  jet r;
  r.v = a.v + b.v;
  for (int i = 0; i < NX; i++) r.g[i] = a.g[i] + b.g[i];
  for (int i = 0; i < NH; i++) r.h[i] = a.h[i] + b.h[i];
  return r;
}
static jet jsub(jet a, jet b) {
  jet r;
  r.v = a.v - b.v;
  for (int i = 0; i < NX; i++) r.g[i] = a.g[i] - b.g[i];
  for (int i = 0; i < NH; i++) r.h[i] = a.h[i] - b.h[i];
  return r;
}
static jet jscale(jet a, double s) {
  jet r;
  r.v = a.v * s;
  for (int i = 0; i < NX; i++) r.g[i] = a.g[i] * s;
  for (int i = 0; i < NH; i++) r.h[i] = a.h[i] * s;
  return r;
}
static jet jadds(jet a, double s) {
  a.v += s;
  return a;
}
static jet jmul(jet a, jet b) {
  jet r;
  r.v = a.v * b.v;
  for (int i = 0; i < NX; i++) r.g[i] = a.v * b.g[i] + b.v * a.g[i];
  for (int i = 0; i < NX; i++)
    for (int j = 0; j <= i; j++)
      r.h[hidx(i, j)] = a.v * b.h[hidx(i, j)] + b.v * a.h[hidx(i, j)] + a.g[i] * b.g[j] + a.g[j] * b.g[i];
  return r;
}
static jet junary(jet a, double f0, double f1, double f2) {
  jet r;
  r.v = f0;
  for (int i = 0; i < NX; i++) r.g[i] = f1 * a.g[i];
  for (int i = 0; i < NX; i++)
    for (int j = 0; j <= i; j++) r.h[hidx(i, j)] = f1 * a.h[hidx(i, j)] + f2 * a.g[i] * a.g[j];
  return r;
}
static jet jrecip(jet a) { return junary(a, 1.0 / a.v, -1.0 / (a.v * a.v), 2.0 / (a.v * a.v * a.v)); }
static jet jdiv(jet a, jet b) { return jmul(a, jrecip(b)); }
static jet jsin(jet a) { return junary(a, sin(a.v), cos(a.v), -sin(a.v)); }
static jet jcos(jet a) { return junary(a, cos(a.v), -sin(a.v), -cos(a.v)); }
static jet jatan(jet a) {
  double d = 1.0 + a.v * a.v;
  return junary(a, atan(a.v), 1.0 / d, -2.0 * a.v / (d * d));
}
/* sin(sign(mu)*mu) (model.py:77-78).  eps = 0: CasADi's derivative convention sign' = 0.
 * eps > 0: |mu| -> sqrt(mu^2 + eps^2) (C-infinity, error <= eps). */
static jet jsinabs(jet a, double eps) {
  if (eps > 0.0) {
    double r = sqrt(a.v * a.v + eps * eps), q = a.v / r;
    return junary(a, sin(r), cos(r) * q, -sin(r) * q * q + cos(r) * eps * eps / (r * r * r));
  }
  double sg = (a.v > 0) - (a.v < 0);
  return junary(a, sin(fabs(a.v)), sg * cos(a.v), -sg * sg * sin(fabs(a.v)));
}
static double sinabs_val(double mu, double eps) { return eps > 0.0 ? sin(sqrt(mu * mu + eps * eps)) : sin(fabs(mu)); }
static jet jatan2(jet y, jet x) {
  double q = y.v * y.v + x.v * x.v;
  double ty = x.v / q, tx = -y.v / q;
  double tyy = -2.0 * x.v * y.v / (q * q), txx = -tyy, txy = (y.v * y.v - x.v * x.v) / (q * q);
  jet r;
  r.v = atan2(y.v, x.v);
  for (int i = 0; i < NX; i++) r.g[i] = ty * y.g[i] + tx * x.g[i];
  for (int i = 0; i < NX; i++)
    for (int j = 0; j <= i; j++)
      r.h[hidx(i, j)] = ty * y.h[hidx(i, j)] + tx * x.h[hidx(i, j)] + tyy * y.g[i] * y.g[j] +
                        txx * x.g[i] * x.g[j] + txy * (y.g[i] * x.g[j] + y.g[j] * x.g[i]);
  return r;
}

/* ------------------------------------------------------------------ tables */
typedef struct {
  int n;
  const double *s_kappa, *kappa, *s_arc, *n_left, *n_right, *v_ref;
  double eps_s;  /* table-kink smoothing length [m]  (0 = exact PWL)        */
  double eps_mu; /* |mu| smoothing [rad]             (0 = exact sin|mu|)    */
  double period; /* options.periodic_tables: span of the grids, tables evaluated at s modulo it; else 0 */
} tables_t;

static tables_t tables_view(const double* packed, int n) {
  tables_t t;
  t.n = n;
  t.s_kappa = packed;
  t.kappa = packed + n;
  t.s_arc = packed + 2 * n;
  t.n_left = packed + 3 * n;
  t.n_right = packed + 4 * n;
  t.v_ref = packed + 5 * n;
  t.eps_s = 0.0, t.eps_mu = 0.0, t.period = 0.0;
  return t;
}
/* CasADi linear interpolant: interval i with grid[i] <= s < grid[i+1], clamped to [0, n-2] => linear
 * extrapolation outside the grid. */
static int lut_interval(const double* grid, int n, double s) {
  double inv = (double)(n - 1) / (grid[n - 1] - grid[0]);
  double fi = (s - grid[0]) * inv;
  int i = !(fi > 0) ? 0 : (fi >= n - 2 ? n - 2 : (int)fi); /* (NaN -> 0) */
  while (i > 0 && s < grid[i]) i--;
  while (i < n - 2 && s >= grid[i + 1]) i++;
  return i;
}
/* Table value, slope and second derivative.
 * eps = 0: the reference's exact piece-wise-linear table (kinks at the knots).
 * eps > 0: every interior knot's kink (J/2)|z|, z = s - g_i, J = slope jump, is replaced inside |z| < W_i
 * (W_i = half the shorter adjacent interval, so windows never overlap) by (J/2) k(z),
 *   k(z) = sqrt(z^2 + eps^2) + a z^2 + b,   a, b such that k(W) = W, k'(W) = 1,
 * i.e. a compact-support pseudo-Huber patch: C1 everywhere, identical to the PWL table outside the windows,
 * -> |z| as eps -> 0 (max deviation (J/2) eps at the knot).  See DESIGN.md "non-smoothness". */
static void lut_eval2(const double* grid, const double* y, int n, double s, double eps, double period, double* val,
                      double* slope, double* curv) {
  if (period > 0.0) s -= period * floor((s - grid[0]) / period); /* closed track (options.periodic_tables) */
  int i = lut_interval(grid, n, s);
  double d = grid[i + 1] - grid[i];
  double sl = (y[i + 1] - y[i]) / d;
  *val = y[i] + sl * (s - grid[i]);
  *slope = sl;
  *curv = 0.0;
  if (eps > 0.0) {
    int kn = -1;
    double z = 0, W = 0, sg = 0;
    if (i > 0) {
      double Wk = 0.5 * fmin(grid[i] - grid[i - 1], d), zz = s - grid[i];
      if (zz >= 0 && zz < Wk) kn = i, z = zz, W = Wk, sg = 1.0;
    }
    if (kn < 0 && i + 1 < n - 1) {
      double Wk = 0.5 * fmin(d, grid[i + 2] - grid[i + 1]), zz = s - grid[i + 1];
      if (zz < 0 && -zz < Wk) kn = i + 1, z = zz, W = Wk, sg = -1.0;
    }
    if (kn > 0) {
      double sa = (y[kn] - y[kn - 1]) / (grid[kn] - grid[kn - 1]);
      double sb = (y[kn + 1] - y[kn]) / (grid[kn + 1] - grid[kn]);
      double J = sb - sa, R = sqrt(z * z + eps * eps), RW = sqrt(W * W + eps * eps);
      double a = (1.0 - W / RW) / (2.0 * W), b = W - RW - a * W * W;
      *val += 0.5 * J * (R + a * z * z + b - sg * z);
      *slope += 0.5 * J * (z / R + 2.0 * a * z - sg);
      *curv = 0.5 * J * (eps * eps / (R * R * R) + 2.0 * a);
    }
  }
}
static jet jlut(const double* grid, const double* y, int n, jet s, double eps, double period) {
  double v, sl, cv;
  lut_eval2(grid, y, n, s.v, eps, period, &v, &sl, &cv);
  return junary(s, v, sl, cv);
}
static double lut_val(const double* grid, const double* y, int n, double s, double eps, double period) {
  double v, sl, cv;
  lut_eval2(grid, y, n, s, eps, period, &v, &sl, &cv);
  return v;
}

/* ------------------------------------------------------------------ model (jets over the 8 states) */
typedef struct {
  jet f[6]; /* rows 0..5 of the rhs; rows 6,7 are u[0], u[1] */
} rhs_jets;

static void slip_forces_jet(const ltompc_params* p, jet vx, jet vy, jet r, jet delta, jet* Fyf, jet* Fyr,
                            jet* af_out, jet* ar_out) {
  /* model.py:101-104 */
  jet af = jsub(jatan2(jadd(vy, jscale(r, p->length_f)), vx), delta);
  jet ar = jatan2(jsub(vy, jscale(r, p->length_r)), vx);
  /* model.py:106-114 */
  double L = p->length_f + p->length_r;
  double Fnf = p->length_r * p->mass * p->gravity / L;
  double Fnr = p->length_f * p->mass * p->gravity / L;
  *Fyf = jscale(jsin(jscale(jatan(jscale(af, p->B_f)), p->C_f)), -Fnf * p->D_f);
  *Fyr = jscale(jsin(jscale(jatan(jscale(ar, p->B_r)), p->C_r)), -Fnr * p->D_r);
  if (af_out) *af_out = af;
  if (ar_out) *ar_out = ar;
}

static void rhs_jet(const ltompc_params* p, const tables_t* T, const double* x, rhs_jets* out) {
  jet s = jvar(x[0], 0), n = jvar(x[1], 1), mu = jvar(x[2], 2), vx = jvar(x[3], 3), vy = jvar(x[4], 4),
      r = jvar(x[5], 5), de = jvar(x[6], 6), th = jvar(x[7], 7);
  jet kap = jlut(T->s_kappa, T->kappa, T->n, s, T->eps_s, T->period);                                 /* model.py:66-67 */
  jet cm = jcos(mu), sm = jsin(mu);
  jet sdot = jdiv(jsub(jmul(vx, cm), jmul(vy, sm)), jsub(jconst(1.0), jmul(n, kap))); /* model.py:152 */
  jet Fyf, Fyr;
  slip_forces_jet(p, vx, vy, r, de, &Fyf, &Fyr, NULL, NULL);
  jet Fx = jsub(jadds(jscale(th, p->C_m), -p->Cr_0), jscale(jmul(vx, vx), p->Cr_2)); /* model.py:160 */
  out->f[0] = sdot;                                                                /* :166 */
  out->f[1] = jadd(jmul(vx, sm), jmul(vy, cm));                                    /* :167-169 */
  out->f[2] = jsub(r, jmul(kap, sdot));                                            /* :170-172 */
  out->f[3] = jscale(jadd(jsub(Fx, jmul(Fyf, jsin(de))), jscale(jmul(vy, r), p->mass)), 1.0 / p->mass);
  out->f[4] = jscale(jsub(jadd(Fyr, jmul(Fyf, jcos(de))), jscale(jmul(vx, r), p->mass)), 1.0 / p->mass);
  /* model.py:162-164: Mtv = ptv (rt - r), rt = tan(delta) vx / (l_f + l_r); the reference sets Mtv = 0 (ptv = 0 here) */
  jet Mtv = jscale(jsub(jscale(jmul(jdiv(jsin(de), jcos(de)), vx), 1.0 / (p->length_f + p->length_r)), r), p->ptv);
  out->f[5] = jscale(jadd(jsub(jscale(jmul(Fyf, jcos(de)), p->length_f), jscale(Fyr, p->length_r)), Mtv), 1.0 / p->inertia_z);
}

/* value-only rhs (plant, line search) */
static void rhs_val(const ltompc_params* p, const tables_t* T, const double* x, const double* u, double* f) {
  double kap = lut_val(T->s_kappa, T->kappa, T->n, x[0], T->eps_s, T->period);
  double n = x[1], mu = x[2], vx = x[3], vy = x[4], r = x[5], de = x[6], th = x[7];
  double sdot = (vx * cos(mu) - vy * sin(mu)) / (1.0 - n * kap);
  double af = atan2(vy + p->length_f * r, vx) - de;
  double ar = atan2(vy - p->length_r * r, vx);
  double L = p->length_f + p->length_r;
  double Fnf = p->length_r * p->mass * p->gravity / L, Fnr = p->length_f * p->mass * p->gravity / L;
  double Fyf = -Fnf * p->D_f * sin(p->C_f * atan(p->B_f * af));
  double Fyr = -Fnr * p->D_r * sin(p->C_r * atan(p->B_r * ar));
  double Fx = p->C_m * th - p->Cr_0 - p->Cr_2 * vx * vx;
  f[0] = sdot;
  f[1] = vx * sin(mu) + vy * cos(mu);
  f[2] = r - kap * sdot;
  f[3] = (Fx - Fyf * sin(de) + p->mass * vy * r) / p->mass;
  f[4] = (Fyr + Fyf * cos(de) - p->mass * vx * r) / p->mass;
  f[5] = (Fyf * p->length_f * cos(de) - Fyr * p->length_r + p->ptv * (tan(de) * vx / L - r)) / p->inertia_z;
  f[6] = u[0]; /* model.py:183: rhs('steering_angle') = steering_angle_change */
  f[7] = u[1]; /* model.py:182 */
}

/* stage cost at a node: lterm (terminal = 0) or mterm (terminal = 1); controller.py:51-53, model.py:124-128 */
static jet cost_jet(const ltompc_params* p, const tables_t* T, const double* x, int terminal) {
  jet s = jvar(x[0], 0), n = jvar(x[1], 1), mu = jvar(x[2], 2), vx = jvar(x[3], 3), vy = jvar(x[4], 4),
      de = jvar(x[6], 6);
  jet m = jadd(jadd(jscale(jmul(n, n), p->q_n), jscale(jmul(mu, mu), p->q_mu)), jscale(jmul(vy, vy), p->q_vy));
  if (terminal) return m;
  jet vref = jlut(T->s_arc, T->v_ref, T->n, s, T->eps_s, T->period);
  jet e = jsub(vx, jscale(vref, p->vref_scale));
  jet bdyn = jatan(jdiv(vy, vx));
  jet bkin = jatan(jscale(de, p->length_r / (p->length_f + p->length_r)));
  jet db = jsub(bdyn, bkin);
  return jadd(jadd(m, jscale(jmul(e, e), p->q_v)), jscale(jmul(db, db), p->q_B));
}
static double cost_val(const ltompc_params* p, const tables_t* T, const double* x, int terminal) {
  double m = p->q_n * x[1] * x[1] + p->q_mu * x[2] * x[2] + p->q_vy * x[4] * x[4];
  if (terminal) return m;
  double vref = lut_val(T->s_arc, T->v_ref, T->n, x[0], T->eps_s, T->period);
  double e = x[3] - p->vref_scale * vref;
  double db = atan(x[4] / x[3]) - atan(x[6] * p->length_r / (p->length_f + p->length_r));
  return m + p->q_v * e * e + p->q_B * db * db;
}
/* track-boundary constraints g <= 0 (model.py:70-84):
 *   left :  n - (L/2) sin|mu| + (W/2) cos mu - N_L(s) <= 0
 *   right: -n + (L/2) sin|mu| + (W/2) cos mu - N_R(s) <= 0
 * The right one has a convex kink at mu = 0, exactly where the cost wants mu; since sin|mu| = max(sin mu, -sin mu)
 * on |mu| <= pi/2 (the mu bounds, controller.py:80,88) it is EXACTLY equivalent to the two smooth constraints
 *   gR+ = -n + (L/2) sin mu + (W/2) cos mu - N_R(s) <= 0,   gR- = -n - (L/2) sin mu + (W/2) cos mu - N_R(s) <= 0
 * (same feasible set, same minimisers; the reference multiplier is nu(gR+) + nu(gR-)).  The left one has a
 * concave kink (a local max of the Lagrangian in mu, never a minimiser) and keeps sin(sign(mu) mu) with CasADi's
 * derivative convention. */
static void cons_jet(const ltompc_params* p, const tables_t* T, const double* x, jet* g) {
  jet s = jvar(x[0], 0), n = jvar(x[1], 1), mu = jvar(x[2], 2);
  double len = p->length_f + p->length_r, wid = p->width;
  jet NL = jlut(T->s_arc, T->n_left, T->n, s, T->eps_s, T->period), NR = jlut(T->s_arc, T->n_right, T->n, s, T->eps_s, T->period);
  jet sa = jscale(jsinabs(mu, T->eps_mu), 0.5 * len), sp = jscale(jsin(mu), 0.5 * len), cw = jscale(jcos(mu), 0.5 * wid);
  g[0] = jsub(jadd(jsub(n, sa), cw), NL);
  g[1] = jsub(jadd(jsub(sp, n), cw), NR);
  g[2] = jsub(jsub(jsub(cw, sp), n), NR);
}
static void cons_val(const ltompc_params* p, const tables_t* T, const double* x, double* g) {
  double NL = lut_val(T->s_arc, T->n_left, T->n, x[0], T->eps_s, T->period), NR = lut_val(T->s_arc, T->n_right, T->n, x[0], T->eps_s, T->period);
  double len = p->length_f + p->length_r, wid = p->width;
  double sa = 0.5 * len * sinabs_val(x[2], T->eps_mu), sp = 0.5 * len * sin(x[2]), cw = 0.5 * wid * cos(x[2]);
  g[0] = x[1] - sa + cw - NL;
  g[1] = -x[1] + sp + cw - NR;
  g[2] = -x[1] - sp + cw - NR;
}

/* Friction-ellipse constraints of the two axles (model.py:86-99 get_traction_ellipse_constraint; registered as SOFT nl
 * constraints in lines the reference has commented out, controller.py:72-74):
 *   long = rho 0.5 C_m T ,  ellipse_a = long^2 + F_y,a^2 - (alpha D_a)^2 <= 0 ,  a = front, rear.
 * Here normalised by the radius, g_a = (long^2 + F_y,a^2) / D_a^2 - 1 with D_a = params.ell_D_f / ell_D_r (= alpha D_a; with the
 * reference's D = 1.0 this IS its expression; a physical radius is F_N D), rho = params.ell_rho.  Always soft: elastic
 * variable with penalty params.ell_penalty (do_mpc: penalty_term_cons). */
static void ell_jet(const ltompc_params* p, const double* x, jet* g) {
  jet vx = jvar(x[3], 3), vy = jvar(x[4], 4), r = jvar(x[5], 5), de = jvar(x[6], 6), th = jvar(x[7], 7);
  jet Fyf, Fyr;
  slip_forces_jet(p, vx, vy, r, de, &Fyf, &Fyr, NULL, NULL);
  jet lng = jscale(th, p->ell_rho * 0.5 * p->C_m);
  jet l2 = jmul(lng, lng);
  g[0] = jadds(jscale(jadd(l2, jmul(Fyf, Fyf)), 1.0 / (p->ell_D_f * p->ell_D_f)), -1.0);
  g[1] = jadds(jscale(jadd(l2, jmul(Fyr, Fyr)), 1.0 / (p->ell_D_r * p->ell_D_r)), -1.0);
}
static void ell_val(const ltompc_params* p, const double* x, double* g) {
  double vx = x[3], vy = x[4], r = x[5], de = x[6], th = x[7];
  double af = atan2(vy + p->length_f * r, vx) - de, ar = atan2(vy - p->length_r * r, vx);
  double L = p->length_f + p->length_r;
  double Fnf = p->length_r * p->mass * p->gravity / L, Fnr = p->length_f * p->mass * p->gravity / L;
  double Fyf = -Fnf * p->D_f * sin(p->C_f * atan(p->B_f * af)), Fyr = -Fnr * p->D_r * sin(p->C_r * atan(p->B_r * ar));
  double lng = p->ell_rho * 0.5 * p->C_m * th;
  g[0] = (lng * lng + Fyf * Fyf) / (p->ell_D_f * p->ell_D_f) - 1.0;
  g[1] = (lng * lng + Fyr * Fyr) / (p->ell_D_r * p->ell_D_r) - 1.0;
}

/* ------------------------------------------------------------------ small dense LA */
/* LU with partial pivoting, n <= 16, row-major a[n*n]; returns 0 ok */
static int lu_factor(double* a, int n, int* piv) {
  for (int k = 0; k < n; k++) {
    int pr = k;
    double mx = fabs(a[k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (fabs(a[i * n + k]) > mx) mx = fabs(a[i * n + k]), pr = i;
    if (mx == 0.0 || !isfinite(mx)) return -1;
    piv[k] = pr;
    if (pr != k)
      for (int j = 0; j < n; j++) {
        double t = a[k * n + j];
        a[k * n + j] = a[pr * n + j];
        a[pr * n + j] = t;
      }
    for (int i = k + 1; i < n; i++) {
      double l = a[i * n + k] / a[k * n + k];
      a[i * n + k] = l;
      for (int j = k + 1; j < n; j++) a[i * n + j] -= l * a[k * n + j];
    }
  }
  return 0;
}
static void lu_solve(const double* a, int n, const int* piv, double* b) { /* A x = b */
  for (int k = 0; k < n; k++) /* rows were swapped whole (LAPACK style): permute b first */
    if (piv[k] != k) {
      double t = b[k];
      b[k] = b[piv[k]];
      b[piv[k]] = t;
    }
  for (int k = 0; k < n; k++)
    for (int i = k + 1; i < n; i++) b[i] -= a[i * n + k] * b[k];
  for (int i = n - 1; i >= 0; i--) {
    for (int j = i + 1; j < n; j++) b[i] -= a[i * n + j] * b[j];
    b[i] /= a[i * n + i];
  }
}
static void lu_solve_t(const double* a, int n, const int* piv, double* b) { /* A^T x = b, A = P^T L U */
  /* U^T y = b */
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < i; j++) b[i] -= a[j * n + i] * b[j];
    b[i] /= a[i * n + i];
  }
  /* L^T z = y */
  for (int i = n - 1; i >= 0; i--)
    for (int j = i + 1; j < n; j++) b[i] -= a[j * n + i] * b[j];
  /* x = P^T z : undo row swaps in reverse */
  for (int k = n - 1; k >= 0; k--)
    if (piv[k] != k) {
      double t = b[k];
      b[k] = b[piv[k]];
      b[piv[k]] = t;
    }
}

/* ------------------------------------------------------------------ NLP bookkeeping */
typedef struct {
  int n_xb;               /* finite state bounds */
  int xb_idx[MAXXB];      /* state index */
  double xb_sgn[MAXXB];   /* -1: lower (h = lb - x), +1: upper (h = x - ub) */
  double xb_val[MAXXB];
  int n_ub;
  int ub_idx[4];
  double ub_sgn[4], ub_val[4];
  int nel; /* 0, or NEL when the friction-ellipse constraints are switched on */
  int nnl; /* nonlinear constraints per node = NNL + nel */
  int ni;  /* inequalities per slot = n_ub + 2 n_xb + nnl */
} bounds_t;

static void build_bounds(const ltompc_params* p, bounds_t* b) {
  b->n_xb = 0;
  for (int i = 0; i < NX; i++) { /* order: per state, lower then upper */
    if (p->x_lb[i] > -LTOMPC_NO_BOUND) {
      b->xb_idx[b->n_xb] = i, b->xb_sgn[b->n_xb] = -1, b->xb_val[b->n_xb] = p->x_lb[i];
      b->n_xb++;
    }
    if (p->x_ub[i] < LTOMPC_NO_BOUND) {
      b->xb_idx[b->n_xb] = i, b->xb_sgn[b->n_xb] = +1, b->xb_val[b->n_xb] = p->x_ub[i];
      b->n_xb++;
    }
  }
  b->n_ub = 0;
  for (int i = 0; i < NU; i++) {
    if (p->u_lb[i] > -LTOMPC_NO_BOUND) {
      b->ub_idx[b->n_ub] = i, b->ub_sgn[b->n_ub] = -1, b->ub_val[b->n_ub] = p->u_lb[i];
      b->n_ub++;
    }
    if (p->u_ub[i] < LTOMPC_NO_BOUND) {
      b->ub_idx[b->n_ub] = i, b->ub_sgn[b->n_ub] = +1, b->ub_val[b->n_ub] = p->u_ub[i];
      b->n_ub++;
    }
  }
  b->nel = p->ell_penalty > 0 ? NEL : 0;
  b->nnl = NNL + b->nel;
  b->ni = b->n_ub + 2 * b->n_xb + b->nnl;
}
static inline double bound_h(double sgn, double val, double x) { return sgn < 0 ? val - x : x - val; }

/* Radau-IIA(2) collocation in do_mpc's form:
 *   G1 = h f(c,u) + 2 x - 1.5 c - 0.5 x+ = 0 ;  G2 = h f(x+,u) - 2 x + 4.5 c - 2.5 x+ = 0            */
typedef struct {
  int N;
  double *x, *c, *u, *l1, *l2; /* (N+1)*8, N*8, N*2, N*8, N*8 */
  double *t, *nu;              /* N*MAXI */
  double *e;                   /* N*NNLT: elastic variables of the softened nonlinear constraints */
  double rho;                  /* penalty of the track constraints' elastic variables (0: hard) */
  double pen[NNLT];            /* penalty per nonlinear constraint: rho x 3, params.ell_penalty x 2 (it_set_rho) */
} iterate_t;
static void it_set_rho(iterate_t* it, double rho, const ltompc_params* p) {
  it->rho = rho;
  for (int q = 0; q < NNL; q++) it->pen[q] = rho;
  for (int q = NNL; q < NNLT; q++) it->pen[q] = p->ell_penalty;
}

static iterate_t it_alloc(int N) {
  iterate_t it;
  it.N = N;
  it.x = calloc((size_t)(N + 1) * NX, sizeof(double));
  it.c = calloc((size_t)N * NX, sizeof(double));
  it.u = calloc((size_t)N * NU, sizeof(double));
  it.l1 = calloc((size_t)N * NX, sizeof(double));
  it.l2 = calloc((size_t)N * NX, sizeof(double));
  it.t = calloc((size_t)N * MAXI, sizeof(double));
  it.nu = calloc((size_t)N * MAXI, sizeof(double));
  it.e = calloc((size_t)N * NNLT, sizeof(double));
  it.rho = 0;
  memset(it.pen, 0, sizeof it.pen);
  return it;
}
static void it_free(iterate_t* it) {
  free(it->x), free(it->c), free(it->u), free(it->l1), free(it->l2), free(it->t), free(it->nu), free(it->e);
}

/* inequality values of slot k (u_k, c_k, x_{k+1}); order: u bounds, c bounds, x+ bounds, gL, gR.
 * nl constraints exist at nodes 1..N-1 only (node 0 is fixed data, node N is not checked by do_mpc). */
static void slot_ineq(const ltompc_params* p, const tables_t* T, const bounds_t* bd, int N, int k,
                      const double* u, const double* c, const double* xp, const double* e, const double* pen, double* h, int* active) {
  int m = 0;
  for (int i = 0; i < bd->n_ub; i++, m++) h[m] = bound_h(bd->ub_sgn[i], bd->ub_val[i], u[bd->ub_idx[i]]), active[m] = 1;
  for (int i = 0; i < bd->n_xb; i++, m++) h[m] = bound_h(bd->xb_sgn[i], bd->xb_val[i], c[bd->xb_idx[i]]), active[m] = 1;
  for (int i = 0; i < bd->n_xb; i++, m++) h[m] = bound_h(bd->xb_sgn[i], bd->xb_val[i], xp[bd->xb_idx[i]]), active[m] = 1;
  int nl = (k + 1 <= N - 1);
  double g[NNLT] = {-1, -1, -1, -1, -1};
  if (nl) {
    cons_val(p, T, xp, g);
    if (bd->nel) ell_val(p, xp, g + NNL);
  }
  for (int q = 0; q < bd->nnl; q++) h[m] = g[q] - (e && nl && pen[q] > 0 ? e[q] : 0.0), active[m++] = nl; /* soft: h = g - e */
}

/* constraint residuals of an iterate: collocation equations G1, G2 (N x 8 each) and inequality rows r = h + t
 * (N x MAXI; soft track constraints: h = g - e); also the filter measures theta = ||.||_1, cost, sum of logs. */
static void eval_residuals(const ltompc_params* p, const ltompc_options* o, const tables_t* T, const bounds_t* bd,
                           const iterate_t* it, const double* uprev, double* G1, double* G2, double* R, double* theta,
                           double* cost, double* sumlog) {
  int N = it->N;
  double th = 0, co = 0, sl = 0, hdt = o->t_step;
  co += cost_val(p, T, it->x, 0); /* lterm(x_0): constant, kept so that J matches the NLP objective */
  for (int k = 0; k < N; k++) {
    const double *xk = it->x + k * NX, *xp = it->x + (k + 1) * NX, *c = it->c + k * NX, *u = it->u + k * NU;
    const double* v = k ? it->u + (k - 1) * NU : uprev;
    double f1[NX], f2[NX];
    rhs_val(p, T, c, u, f1);
    rhs_val(p, T, xp, u, f2);
    for (int i = 0; i < NX; i++) {
      double a = hdt * f1[i] + 2 * xk[i] - 1.5 * c[i] - 0.5 * xp[i];
      double b = hdt * f2[i] - 2 * xk[i] + 4.5 * c[i] - 2.5 * xp[i];
      th += fabs(a);
      th += fabs(b);
      if (G1) G1[k * NX + i] = a, G2[k * NX + i] = b;
    }
    co += cost_val(p, T, xp, k == N - 1);
    for (int i = 0; i < NU; i++) co += p->r_du[i] * (u[i] - v[i]) * (u[i] - v[i]);
    double h[MAXI];
    int act[MAXI];
    slot_ineq(p, T, bd, N, k, u, c, xp, it->e + k * NNLT, it->pen, h, act);
    for (int m = 0; m < bd->ni; m++) {
      if (R) R[k * MAXI + m] = 0.0;
      if (act[m]) {
        double t = it->t[k * MAXI + m];
        th += fabs(h[m] + t);
        if (R) R[k * MAXI + m] = h[m] + t;
        sl += log(t);
        const int q = m - (bd->ni - bd->nnl);
        if (q >= 0 && it->pen[q] > 0) {
          double e = it->e[k * NNLT + q];
          sl += log(e), co += it->pen[q] * e;
        }
      }
    }
  }
  *theta = th, *cost = co, *sumlog = sl;
}

/* ------------------------------------------------------------------ per-slot linearisation */
typedef struct {
  double E1[64], E2[64], G1[8], G2[8];
  double Hc[64], gc[8];     /* QP Hessian / gradient for c_k (barrier terms included)          */
  double Hxp[64], gxp[8];   /* ... for node x_{k+1}                                             */
  double Du[2], gub[2];     /* u_k: diagonal barrier Hessian and barrier gradient               */
  double dc_dual[8], dxp_dual[8], du_dual[2]; /* parts of grad_z Lagrangian not involving lambda */
  double gcost[8];          /* grad of node cost at x_{k+1}                                     */
  double gnl[NNLT][8];      /* grad of gL, gR+, gR- (and the two ellipse constraints) at x_{k+1}       */
  double h[MAXI];
  int act[MAXI];
  double cost;
  /* right-hand side of the Newton system: the constraint residuals the step has to remove.  At the iterate they are
   * G1, G2, h + t; a second-order correction replaces them by alpha c(x) + c(x + alpha d) (same matrix). */
  double rG1[8], rG2[8], rI[MAXI];
} slot_lin;

/* Gradient blocks of the stage QP for the current right-hand-side residuals (L->rI) and barrier parameter.
 * Barrier: sigma = (mu + nu r) / t per inequality with residual r = h + t.  Softened track constraint (rho > 0):
 * g - e + t = 0, e >= 0 with multiplier z = rho - nu; eliminating (dt, de, dnu) gives
 * dnu = Sg (grad g . dx + geff + mu / nu - mu / z), geff = r + e - t (= g at the iterate). */
static void slot_gradients(const bounds_t* bd, const iterate_t* it, int k, double mu, slot_lin* L) {
  const double *t = it->t + k * MAXI, *nu = it->nu + k * MAXI;
  for (int a = 0; a < NX; a++) L->gc[a] = 0.0, L->gxp[a] = L->gcost[a];
  L->gub[0] = L->gub[1] = 0.0;
  int m = 0;
  for (int i = 0; i < bd->n_ub; i++, m++) L->gub[bd->ub_idx[i]] += bd->ub_sgn[i] * ((mu + nu[m] * L->rI[m]) / t[m]);
  for (int i = 0; i < bd->n_xb; i++, m++) L->gc[bd->xb_idx[i]] += bd->xb_sgn[i] * ((mu + nu[m] * L->rI[m]) / t[m]);
  for (int i = 0; i < bd->n_xb; i++, m++) L->gxp[bd->xb_idx[i]] += bd->xb_sgn[i] * ((mu + nu[m] * L->rI[m]) / t[m]);
  if (L->act[m])
    for (int q = 0; q < bd->nnl; q++) {
      int mm = m + q;
      double sg = (mu + nu[mm] * L->rI[mm]) / t[mm];
      if (it->pen[q] > 0) {
        double e = it->e[k * NNLT + q], z = it->pen[q] - nu[mm];
        double Sg = 1.0 / (t[mm] / nu[mm] + e / z);
        /* (geff from the SAME numbers the step recovery uses: a difference of one ulp of the O(1) terms of g is
         *  amplified by Sg ~ nu / t ~ 1e9 into the dual residual) */
        sg = (nu[mm] + Sg * (L->rI[mm] + e - t[mm])) + mu * (Sg * (1.0 / nu[mm] - 1.0 / z));
      }
      for (int a = 0; a < NX; a++) L->gxp[a] += sg * L->gnl[q][a];
    }
}

static void linearise_slot(const ltompc_params* p, const ltompc_options* o, const tables_t* T, const bounds_t* bd,
                           const iterate_t* it, int k, double mu, slot_lin* L) {
  int N = it->N;
  double hdt = o->t_step;
  const double *xk = it->x + k * NX, *xp = it->x + (k + 1) * NX, *c = it->c + k * NX, *u = it->u + k * NU;
  const double *l1 = it->l1 + k * NX, *l2 = it->l2 + k * NX;
  const double *t = it->t + k * MAXI, *nu = it->nu + k * MAXI;
  rhs_jets F1, F2;
  rhs_jet(p, T, c, &F1);
  rhs_jet(p, T, xp, &F2);
  memset(L->E1, 0, sizeof L->E1), memset(L->E2, 0, sizeof L->E2);
  memset(L->Hc, 0, sizeof L->Hc), memset(L->Hxp, 0, sizeof L->Hxp);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < NX; j++) L->E1[i * 8 + j] = hdt * F1.f[i].g[j], L->E2[i * 8 + j] = hdt * F2.f[i].g[j];
  for (int i = 0; i < NX; i++) L->E1[i * 8 + i] -= 1.5, L->E2[i * 8 + i] -= 2.5;
  for (int i = 0; i < NX; i++) {
    double f1 = i < 6 ? F1.f[i].v : u[i - 6], f2 = i < 6 ? F2.f[i].v : u[i - 6];
    L->G1[i] = hdt * f1 + 2 * xk[i] - 1.5 * c[i] - 0.5 * xp[i];
    L->G2[i] = hdt * f2 - 2 * xk[i] + 4.5 * c[i] - 2.5 * xp[i];
    L->rG1[i] = L->G1[i], L->rG2[i] = L->G2[i];
  }
  for (int i = 0; i < 6; i++)
    for (int a = 0; a < NX; a++)
      for (int b = 0; b < NX; b++) {
        L->Hc[a * 8 + b] += hdt * l1[i] * F1.f[i].h[hidx(a, b)];
        L->Hxp[a * 8 + b] += hdt * l2[i] * F2.f[i].h[hidx(a, b)];
      }
  jet cj = cost_jet(p, T, xp, k == N - 1);
  L->cost = cj.v;
  for (int a = 0; a < NX; a++) {
    L->gcost[a] = cj.g[a];
    L->dxp_dual[a] = cj.g[a];
    L->dc_dual[a] = 0;
    for (int b = 0; b < NX; b++) L->Hxp[a * 8 + b] += cj.h[hidx(a, b)];
  }
  L->Du[0] = L->Du[1] = 0, L->du_dual[0] = L->du_dual[1] = 0;
  slot_ineq(p, T, bd, N, k, u, c, xp, it->e + k * NNLT, it->pen, L->h, L->act);
  for (int m = 0; m < bd->ni; m++) L->rI[m] = L->act[m] ? L->h[m] + t[m] : 0.0;
  /* barrier contributions to the Hessian: Sigma = nu/t */
  int m = 0;
  for (int i = 0; i < bd->n_ub; i++, m++) {
    int j = bd->ub_idx[i];
    L->Du[j] += nu[m] / t[m], L->du_dual[j] += bd->ub_sgn[i] * nu[m];
  }
  for (int i = 0; i < bd->n_xb; i++, m++) {
    int j = bd->xb_idx[i];
    L->Hc[j * 8 + j] += nu[m] / t[m], L->dc_dual[j] += bd->xb_sgn[i] * nu[m];
  }
  for (int i = 0; i < bd->n_xb; i++, m++) {
    int j = bd->xb_idx[i];
    L->Hxp[j * 8 + j] += nu[m] / t[m], L->dxp_dual[j] += bd->xb_sgn[i] * nu[m];
  }
  memset(L->gnl, 0, sizeof L->gnl);
  if (L->act[m]) {
    jet g[NNLT];
    cons_jet(p, T, xp, g);
    if (bd->nel) ell_jet(p, xp, g + NNL);
    for (int q = 0; q < bd->nnl; q++) {
      int mm = m + q;
      double Sg = nu[mm] / t[mm];
      if (it->pen[q] > 0) { /* softened: g - e + t = 0, e >= 0 with multiplier z = rho - nu */
        double e = it->e[k * NNLT + q], z = it->pen[q] - nu[mm];
        Sg = 1.0 / (t[mm] / nu[mm] + e / z);
      }
      for (int a = 0; a < NX; a++) {
        L->gnl[q][a] = g[q].g[a];
        L->dxp_dual[a] += nu[mm] * g[q].g[a];
        for (int b = 0; b < NX; b++) L->Hxp[a * 8 + b] += nu[mm] * g[q].h[hidx(a, b)] + Sg * g[q].g[a] * g[q].g[b];
      }
    }
  }
  slot_gradients(bd, it, k, mu, L);
}

/* ------------------------------------------------------------------ the interior-point solve */
typedef struct {
  double Ac[64], Bc[16], bc[8], A[64], B[16], b[8];
  double M[256];
  int piv[16];
  double Q[64], S[16], R[4], q[8], r[2]; /* condensed-c + u-barrier contributions in (x_k,u_k) */
  double K[16], Kv[4], kff[2];           /* du = kff + K dx + Kv dv */
  double P[64], Pxv[16], Pvv[4], pp[8], pv[2];
  double Hux[16], Hi[4];                 /* kept for right-hand-side-only sweeps (second-order correction) */
} stage_ws;

typedef struct {
  int status, iters;
  int status_solver; /* the solver's own termination status, before the node-0 rule (options.node0_check): what decides how the next
                        solve is warm-started (warm_reset_on_fail, resto_sticky) */
  double kkt, obj, mu;
  int n_reg, n_lsfail, n_soc, n_resto, n_fallback, n_shift;
  double viol; /* largest elastic variable at termination (0 on the hard constraints); g0 when node 0 decided the status */
  double g0;   /* largest track constraint at the measured state (options.node0_check) */
  double rho_end; /* penalty of the elastic variables at termination (0: hard constraints) */
  int trig; /* diagnostics (experiment): bit 0 / 1: the early-stall rule fired before / after the shifted restart */
  int mu_stay_max; /* diagnostics: longest run of iterations without a decrease of the barrier parameter */
  double rd_max; /* diagnostics: largest dual infeasibility (in the units of the penalty scale) any iterate of the solve had */
} solve_stats;

#define FILTER_MAX 16 /* (as on the device: the oldest pair leaves a full filter) */
#define ELASTIC_CP_VIOL 0.1 /* [m] violation of a track constraint above which its elastic variable starts on the central path */
#define DW_KEEP 1e-5
#define RHO_UNIT 1000.0 /* penalty of the elastic variables at which the solve runs unscaled (IPOPT's restoration penalty), see S in solve_one */

/* everything one solve works on */
typedef struct {
  const ltompc_params* p;
  const ltompc_options* o;
  tables_t* T;
  bounds_t bd;
  int N, ni;
  double hdt;
  const double* uprev;
  iterate_t it, tr;
  slot_lin* L;
  stage_ws* W;
  double *dx, *dc, *du, *nl1, *nl2, *dt, *dnu, *de;
  double mu, delta_w;
} ipws;

/* matrix part of the elimination of the collocation point: [dc; dx+] = M^-1 (-[G1;G2] - [2;-2] dx - [Bu;Bu] du),
 * and the projection of the c-block of the QP onto (x_k, u_k) */
static int condense_matrices(ipws* s, int k) {
  stage_ws* w = &s->W[k];
  slot_lin* L = s->L;
  double* M = w->M;
  memset(M, 0, sizeof w->M);
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      M[i * 16 + j] = L[k].E1[i * 8 + j];
      M[(8 + i) * 16 + 8 + j] = L[k].E2[i * 8 + j];
    }
  for (int i = 0; i < 8; i++) M[i * 16 + 8 + i] = -0.5, M[(8 + i) * 16 + i] = 4.5;
  if (lu_factor(M, 16, w->piv)) return -1;
  for (int col = 0; col < 10; col++) {
    double rhs[16];
    for (int i = 0; i < 8; i++) {
      if (col < 8) rhs[i] = (i == col) ? -2.0 : 0.0, rhs[8 + i] = (i == col) ? 2.0 : 0.0;
      else rhs[i] = rhs[8 + i] = (i == 6 + (col - 8)) ? -s->hdt : 0.0;
    }
    lu_solve(M, 16, w->piv, rhs);
    for (int i = 0; i < 8; i++) {
      if (col < 8) w->Ac[i * 8 + col] = rhs[i], w->A[i * 8 + col] = rhs[8 + i];
      else w->Bc[i * 2 + col - 8] = rhs[i], w->B[i * 2 + col - 8] = rhs[8 + i];
    }
  }
  double HA[64], HB[16];
  for (int i = 0; i < 8; i++) {
    for (int j = 0; j < 8; j++) {
      double v = 0;
      for (int l = 0; l < 8; l++) v += L[k].Hc[i * 8 + l] * w->Ac[l * 8 + j];
      HA[i * 8 + j] = v;
    }
    for (int j = 0; j < 2; j++) {
      double v = 0;
      for (int l = 0; l < 8; l++) v += L[k].Hc[i * 8 + l] * w->Bc[l * 2 + j];
      HB[i * 2 + j] = v;
    }
  }
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      double v = 0;
      for (int l = 0; l < 8; l++) v += w->Ac[l * 8 + i] * HA[l * 8 + j];
      w->Q[i * 8 + j] = v;
    }
  for (int i = 0; i < 2; i++) {
    for (int j = 0; j < 8; j++) {
      double v = 0;
      for (int l = 0; l < 8; l++) v += w->Bc[l * 2 + i] * HA[l * 8 + j];
      w->S[i * 8 + j] = v;
    }
    for (int j = 0; j < 2; j++) {
      double v = 0;
      for (int l = 0; l < 8; l++) v += w->Bc[l * 2 + i] * HB[l * 2 + j];
      w->R[i * 2 + j] = v;
    }
    w->R[i * 2 + i] += L[k].Du[i];
  }
  /* node terms of x_k come from slot k-1 (x_0 is fixed: dx_0 = 0, nothing to add) */
  if (k > 0)
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) w->Q[i * 8 + j] += L[k - 1].Hxp[i * 8 + j];
  return 0;
}
/* right-hand-side part: bc, b from the residuals (rG1, rG2), q, r from the gradients */
static void condense_rhs(ipws* s, int k) {
  stage_ws* w = &s->W[k];
  slot_lin* L = s->L;
  double rhs[16], Hb[8];
  for (int i = 0; i < 8; i++) rhs[i] = -L[k].rG1[i], rhs[8 + i] = -L[k].rG2[i];
  lu_solve(w->M, 16, w->piv, rhs);
  for (int i = 0; i < 8; i++) w->bc[i] = rhs[i], w->b[i] = rhs[8 + i];
  for (int i = 0; i < 8; i++) {
    double v = L[k].gc[i];
    for (int l = 0; l < 8; l++) v += L[k].Hc[i * 8 + l] * w->bc[l];
    Hb[i] = v;
  }
  for (int i = 0; i < 8; i++) {
    double v = 0;
    for (int l = 0; l < 8; l++) v += w->Ac[l * 8 + i] * Hb[l];
    w->q[i] = v;
  }
  for (int i = 0; i < 2; i++) {
    double v = L[k].gub[i];
    for (int l = 0; l < 8; l++) v += w->Bc[l * 2 + i] * Hb[l];
    w->r[i] = v;
  }
  if (k > 0)
    for (int i = 0; i < 8; i++) w->q[i] += L[k - 1].gxp[i];
}

/* Riccati sweep on the state (x_k, v_k = u_{k-1}).  rhs_only = 0: matrices and vectors, returns 0 when some Huu is
 * not positive definite (inertia test).  rhs_only = 1: the vectors kff, pp, pv again for new b, q, r (the factors K,
 * Kv, P, Pxv, Hux, Huu^-1 of the last full sweep are kept): what a second-order correction costs. */
static int riccati_sweep(ipws* s, double delta_w, int rhs_only) {
  const int N = s->N;
  const ltompc_params* p = s->p;
  const iterate_t* it = &s->it;
  slot_lin* L = s->L;
  double P[64], Pxv[16], Pvv[4], pp[8], pv[2];
  for (int i = 0; i < 64; i++) P[i] = L[N - 1].Hxp[i];
  for (int i = 0; i < 8; i++) P[i * 8 + i] += delta_w, pp[i] = L[N - 1].gxp[i];
  memset(Pxv, 0, sizeof Pxv), memset(Pvv, 0, sizeof Pvv), memset(pv, 0, sizeof pv);
  for (int k = N - 1; k >= 0; k--) {
    stage_ws* w = &s->W[k];
    const double* v = k ? it->u + (k - 1) * NU : s->uprev;
    double r2[2] = {2 * p->r_du[0], 2 * p->r_du[1]};
    double Pb[8]; /* P b + p */
    for (int i = 0; i < 8; i++) {
      double v2 = pp[i];
      for (int l = 0; l < 8; l++) v2 += P[i * 8 + l] * w->b[l];
      Pb[i] = v2;
    }
    double gu[2], gx[8];
    for (int i = 0; i < 2; i++) {
      double v2 = w->r[i] + r2[i] * (it->u[k * NU + i] - v[i]) + pv[i];
      for (int l = 0; l < 8; l++) v2 += w->B[l * 2 + i] * Pb[l] + Pxv[l * 2 + i] * w->b[l];
      gu[i] = v2;
    }
    for (int i = 0; i < 8; i++) {
      double v2 = w->q[i];
      for (int l = 0; l < 8; l++) v2 += w->A[l * 8 + i] * Pb[l];
      gx[i] = v2;
    }
    if (!rhs_only) {
      double PA[64], PB[16];
      for (int i = 0; i < 8; i++) {
        for (int j = 0; j < 8; j++) {
          double v2 = 0;
          for (int l = 0; l < 8; l++) v2 += P[i * 8 + l] * w->A[l * 8 + j];
          PA[i * 8 + j] = v2;
        }
        for (int j = 0; j < 2; j++) {
          double v2 = 0;
          for (int l = 0; l < 8; l++) v2 += P[i * 8 + l] * w->B[l * 2 + j];
          PB[i * 2 + j] = v2;
        }
      }
      double Huu[4], Hxx[64];
      for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 2; j++) {
          double v2 = w->R[i * 2 + j] + Pvv[i * 2 + j];
          for (int l = 0; l < 8; l++) v2 += w->B[l * 2 + i] * PB[l * 2 + j] + w->B[l * 2 + i] * Pxv[l * 2 + j] + Pxv[l * 2 + i] * w->B[l * 2 + j];
          Huu[i * 2 + j] = v2;
        }
        Huu[i * 2 + i] += r2[i] + delta_w;
        for (int j = 0; j < 8; j++) {
          double v2 = w->S[i * 8 + j];
          for (int l = 0; l < 8; l++) v2 += w->B[l * 2 + i] * PA[l * 8 + j] + Pxv[l * 2 + i] * w->A[l * 8 + j];
          w->Hux[i * 8 + j] = v2;
        }
      }
      for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
          double v2 = w->Q[i * 8 + j] + (i == j ? delta_w : 0.0);
          for (int l = 0; l < 8; l++) v2 += w->A[l * 8 + i] * PA[l * 8 + j];
          Hxx[i * 8 + j] = v2;
        }
      /* Huu must be positive definite (inertia test of the reduced Hessian) */
      double det = Huu[0] * Huu[3] - Huu[1] * Huu[2];
      if (!(Huu[0] > 0) || !(det > 1e-14 * Huu[0] * Huu[3]) || !isfinite(det)) return 0;
      w->Hi[0] = Huu[3] / det, w->Hi[1] = -Huu[1] / det, w->Hi[2] = -Huu[2] / det, w->Hi[3] = Huu[0] / det;
      for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 8; j++) w->K[i * 8 + j] = -(w->Hi[i * 2 + 0] * w->Hux[0 * 8 + j] + w->Hi[i * 2 + 1] * w->Hux[1 * 8 + j]);
        for (int j = 0; j < 2; j++) w->Kv[i * 2 + j] = w->Hi[i * 2 + j] * r2[j]; /* -Huu^-1 Huv, Huv = -diag(r2) */
      }
      for (int i = 0; i < 8; i++) {
        for (int j = 0; j < 8; j++) w->P[i * 8 + j] = Hxx[i * 8 + j] + w->Hux[0 * 8 + i] * w->K[0 * 8 + j] + w->Hux[1 * 8 + i] * w->K[1 * 8 + j];
        for (int j = 0; j < 2; j++) w->Pxv[i * 2 + j] = w->Hux[0 * 8 + i] * w->Kv[0 * 2 + j] + w->Hux[1 * 8 + i] * w->Kv[1 * 2 + j];
      }
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) w->Pvv[i * 2 + j] = (i == j ? r2[i] : 0.0) - r2[i] * w->Kv[i * 2 + j];
      for (int i = 0; i < 8; i++) /* symmetrise P */
        for (int j = 0; j < i; j++) w->P[i * 8 + j] = w->P[j * 8 + i] = 0.5 * (w->P[i * 8 + j] + w->P[j * 8 + i]);
    }
    for (int i = 0; i < 2; i++) w->kff[i] = -(w->Hi[i * 2 + 0] * gu[0] + w->Hi[i * 2 + 1] * gu[1]);
    /* cost-to-go of (x_k, v_k) */
    double gv[2] = {-r2[0] * (it->u[k * NU + 0] - v[0]), -r2[1] * (it->u[k * NU + 1] - v[1])};
    for (int i = 0; i < 8; i++) w->pp[i] = gx[i] + w->Hux[0 * 8 + i] * w->kff[0] + w->Hux[1 * 8 + i] * w->kff[1];
    for (int i = 0; i < 2; i++) w->pv[i] = gv[i] - r2[i] * w->kff[i];
    memcpy(P, w->P, sizeof P), memcpy(Pxv, w->Pxv, sizeof Pxv), memcpy(Pvv, w->Pvv, sizeof Pvv);
    memcpy(pp, w->pp, sizeof pp), memcpy(pv, w->pv, sizeof pv);
  }
  return 1;
}

/* forward sweep + recovery of dc and the collocation multipliers */
static void forward_recover(ipws* s, double delta_w) {
  const int N = s->N;
  slot_lin* L = s->L;
  double *dx = s->dx, *dc = s->dc, *du = s->du;
  memset(dx, 0, sizeof(double) * NX);
  double dv[2] = {0, 0};
  for (int k = 0; k < N; k++) {
    stage_ws* w = &s->W[k];
    for (int i = 0; i < 2; i++) {
      double v = w->kff[i] + w->Kv[i * 2 + 0] * dv[0] + w->Kv[i * 2 + 1] * dv[1];
      for (int j = 0; j < 8; j++) v += w->K[i * 8 + j] * dx[k * NX + j];
      du[k * NU + i] = v;
    }
    for (int i = 0; i < 8; i++) {
      double v = w->b[i] + w->B[i * 2] * du[k * NU] + w->B[i * 2 + 1] * du[k * NU + 1];
      double sc = w->bc[i] + w->Bc[i * 2] * du[k * NU] + w->Bc[i * 2 + 1] * du[k * NU + 1];
      for (int j = 0; j < 8; j++) v += w->A[i * 8 + j] * dx[k * NX + j], sc += w->Ac[i * 8 + j] * dx[k * NX + j];
      dx[(k + 1) * NX + i] = v, dc[k * NX + i] = sc;
    }
    dv[0] = du[k * NU], dv[1] = du[k * NU + 1];
  }
  for (int k = 0; k < N; k++) {
    stage_ws* w = &s->W[k];
    /* costate pi_{k+1} = dV_{k+1}/dx_{k+1} ; V_N = terminal node block */
    double pi[8], rhs[16];
    for (int i = 0; i < 8; i++) {
      double v;
      if (k + 1 < N) {
        stage_ws* wn = &s->W[k + 1];
        v = wn->pp[i] + wn->Pxv[i * 2] * du[k * NU] + wn->Pxv[i * 2 + 1] * du[k * NU + 1];
        for (int j = 0; j < 8; j++) v += wn->P[i * 8 + j] * dx[(k + 1) * NX + j];
      } else {
        v = L[N - 1].gxp[i] + delta_w * dx[N * NX + i];
        for (int j = 0; j < 8; j++) v += L[N - 1].Hxp[i * 8 + j] * dx[N * NX + j];
      }
      pi[i] = v;
    }
    for (int i = 0; i < 8; i++) {
      double v = L[k].gc[i];
      for (int j = 0; j < 8; j++) v += L[k].Hc[i * 8 + j] * dc[k * NX + j];
      rhs[i] = -v, rhs[8 + i] = -pi[i];
    }
    lu_solve_t(w->M, 16, w->piv, rhs);
    memcpy(s->nl1 + k * NX, rhs, sizeof(double) * 8), memcpy(s->nl2 + k * NX, rhs + 8, sizeof(double) * 8);
  }
}

/* slack / inequality-multiplier steps for the residuals L->rI, fraction to the boundary, directional derivative of
 * the barrier objective */
static void slack_steps(ipws* s, double tau, double* a_pri_out, double* a_dua_out, double* gphi_out) {
  const int N = s->N, ni = s->ni;
  const bounds_t* bd = &s->bd;
  const ltompc_params* p = s->p;
  iterate_t* it = &s->it;
  slot_lin* L = s->L;
  const double mu = s->mu;
  double *dx = s->dx, *dc = s->dc, *du = s->du, *dt = s->dt, *dnu = s->dnu, *de = s->de;
  double a_pri = 1.0, a_dua = 1.0, gphi_d = 0.0;
  for (int k = 0; k < N; k++) {
    int m = 0;
    const double* v = k ? it->u + (k - 1) * NU : s->uprev;
    const double* dvk = k ? du + (k - 1) * NU : NULL;
    for (int i = 0; i < NU; i++) {
      double ddu = du[k * NU + i] - (dvk ? dvk[i] : 0.0);
      gphi_d += 2 * p->r_du[i] * (it->u[k * NU + i] - v[i]) * ddu;
    }
    for (int a = 0; a < NX; a++) gphi_d += L[k].gcost[a] * dx[(k + 1) * NX + a];
    for (int i = 0; i < ni; i++, m++) {
      if (!L[k].act[m]) {
        dt[k * MAXI + m] = dnu[k * MAXI + m] = 0;
        if (m >= ni - bd->nnl) de[k * NNLT + m - (ni - bd->nnl)] = 0;
        continue;
      }
      double gd;
      if (m < bd->n_ub) gd = bd->ub_sgn[m] * du[k * NU + bd->ub_idx[m]];
      else if (m < bd->n_ub + bd->n_xb) gd = bd->xb_sgn[m - bd->n_ub] * dc[k * NX + bd->xb_idx[m - bd->n_ub]];
      else if (m < bd->n_ub + 2 * bd->n_xb) gd = bd->xb_sgn[m - bd->n_ub - bd->n_xb] * dx[(k + 1) * NX + bd->xb_idx[m - bd->n_ub - bd->n_xb]];
      else {
        int q = m - bd->n_ub - 2 * bd->n_xb;
        gd = 0;
        for (int a = 0; a < NX; a++) gd += L[k].gnl[q][a] * dx[(k + 1) * NX + a];
      }
      double t = it->t[k * MAXI + m], nu = it->nu[k * MAXI + m];
      double dtt = -L[k].rI[m] - gd;
      double dn = (mu - nu * dtt) / t - nu;
      const int q = m - (ni - bd->nnl);
      if (q >= 0) de[k * NNLT + q] = 0;
      if (q >= 0 && it->pen[q] > 0) {
        const double rho = it->pen[q];
        double e = it->e[k * NNLT + q], z = rho - nu;
        double Sg = 1.0 / (t / nu + e / z);
        dn = Sg * (gd + (L[k].rI[m] + e - t) + mu / nu - mu / z);
        dtt = mu / nu - t - (t / nu) * dn;
        double dee = mu / z - e + (e / z) * dn;
        de[k * NNLT + q] = dee;
        if (dee < 0) a_pri = fmin(a_pri, -tau * e / dee);
        if (dn > 0) a_dua = fmin(a_dua, tau * z / dn);
        gphi_d += rho * dee - mu * dee / e;
      }
      dt[k * MAXI + m] = dtt, dnu[k * MAXI + m] = dn;
      if (dtt < 0 && -tau * t / dtt < a_pri && getenv("ORACLE_TRACE2")) fprintf(stderr, "     limit k=%d m=%d t=%.3e dt=%.3e h=%.3e gd=%.3e\n", k, m, t, dtt, L[k].h[m], gd);
      if (dtt < 0) a_pri = fmin(a_pri, -tau * t / dtt);
      if (dn < 0) a_dua = fmin(a_dua, -tau * nu / dn);
      gphi_d -= mu * dtt / t;
    }
  }
  *a_pri_out = a_pri, *a_dua_out = a_dua, *gphi_out = gphi_d;
}

/* trial point it + alpha * (dx, dc, du, dt, de) into tr */
static void make_trial(ipws* s, double alpha) {
  const int N = s->N, ni = s->ni;
  iterate_t *it = &s->it, *tr = &s->tr;
  for (int k = 0; k <= N; k++)
    for (int i = 0; i < NX; i++) tr->x[k * NX + i] = it->x[k * NX + i] + alpha * s->dx[k * NX + i];
  for (int k = 0; k < N; k++) {
    for (int i = 0; i < NX; i++) tr->c[k * NX + i] = it->c[k * NX + i] + alpha * s->dc[k * NX + i];
    for (int i = 0; i < NU; i++) tr->u[k * NU + i] = it->u[k * NU + i] + alpha * s->du[k * NU + i];
    for (int m = 0; m < ni; m++) tr->t[k * MAXI + m] = it->t[k * MAXI + m] + alpha * s->dt[k * MAXI + m];
    for (int q = 0; q < NNLT; q++) tr->e[k * NNLT + q] = it->e[k * NNLT + q] + alpha * s->de[k * NNLT + q];
  }
}

/* slacks, inequality multipliers (and elastic variables) of every slot from the primal point: t = max(-h, bound_push),
 * nu = mu / t; softened track constraints: see the comment inside */
static void init_slacks(ipws* s) {
  const int N = s->N, ni = s->ni;
  iterate_t* it = &s->it;
  const ltompc_options* o = s->o;
  const double mu = s->mu;
  const int nnl = s->bd.nnl;
  for (int k = 0; k < N; k++) {
    double h[MAXI];
    int act[MAXI];
    slot_ineq(s->p, s->T, &s->bd, N, k, it->u + k * NU, it->c + k * NX, it->x + (k + 1) * NX, NULL, it->pen, h, act);
    for (int m = 0; m < ni; m++) {
      const int q = m - (ni - nnl);
      if (q >= 0) it->e[k * NNLT + q] = 0.0;
      if (q >= 0 && it->pen[q] > 0) {
        const double rho = it->pen[q];
        /* softened track constraint: slack and multiplier as for the hard one (t = max(-g, bound_push), nu = mu / t,
         * so that a violated constraint starts as an INFEASIBILITY of g - e + t = 0 that the Newton steps remove, not
         * as a large elastic variable with nu ~ rho that the barrier lets go of only slowly), the elastic variable
         * on the central path of its own pair: e (rho - nu) = mu.  t >= 2 mu / rho keeps nu <= rho / 2. */
        double t = fmax(fmax(-h[m], o->bound_push), 2 * mu / rho), nu = mu / t;
        it->t[k * MAXI + m] = t, it->nu[k * MAXI + m] = nu, it->e[k * NNLT + q] = mu / (rho - nu);
        if (h[m] > ELASTIC_CP_VIOL) {
          /* grossly violated (here the Newton steps would have to shrink t by rho / nu ~ 1e4 at 1 % of a step per
           * iteration): on the central path of the elastic pair instead, e - t = g, t nu = mu, e (rho - nu) = mu */
          double g = h[m], bq = rho * g - 2 * mu;
          t = (-bq + sqrt(bq * bq + 4 * rho * mu * g)) / (2 * rho);
          it->t[k * MAXI + m] = t, it->nu[k * MAXI + m] = mu / t, it->e[k * NNLT + q] = g + t;
        }
        continue;
      }
      double t = -h[m] > o->bound_push ? -h[m] : o->bound_push;
      it->t[k * MAXI + m] = t;
      it->nu[k * MAXI + m] = mu / t;
    }
  }
}

/* warm: 0 cold start (do_mpc set_initial_guess) | 1 from the previous solution | 2 from its primal point only (option
 * warm_reset_on_fail after a solve that did not converge: multipliers 0, barrier at mu_init).  start_elastic: option
 * resto_sticky, the solve starts in the restoration phase's elastic mode. */
static int solve_one(const ltompc_params* p, const ltompc_options* o, const tables_t* T0, int N, const double* x0,
                     const double* uprev, int warm, int start_elastic, double* X, double* C, double* U, double* L1, double* L2,
                     double* Tout, double* NUout, solve_stats* st) {
  ipws WS;
  ipws* s = &WS;
  memset(s, 0, sizeof WS);
  s->p = p, s->o = o, s->N = N, s->hdt = o->t_step, s->uprev = uprev;
  build_bounds(p, &s->bd);
  const bounds_t* bd = &s->bd;
  const int ni = s->ni = bd->ni;
  const double hdt = o->t_step;
  tables_t Tl = *T0;
  tables_t* T = &Tl;
  s->T = T;
#define SMOOTHING(mu_) ((o->smooth_scale > 0 || o->smooth_eps_min > 0) ? fmax(o->smooth_eps_min, o->smooth_scale * (mu_)) : 0.0)
  Tl.eps_mu = 0.0;
  s->it = it_alloc(N), s->tr = it_alloc(N);
  iterate_t *it = &s->it, *tr = &s->tr;
  it_set_rho(it, o->soft_rho, p), it_set_rho(tr, o->soft_rho, p);
  s->de = calloc((size_t)N * NNLT, sizeof(double));
  s->L = malloc(sizeof(slot_lin) * (size_t)N);
  s->W = malloc(sizeof(stage_ws) * (size_t)N);
  slot_lin* L = s->L;
  s->dx = calloc((size_t)(N + 1) * NX, sizeof(double));
  s->dc = calloc((size_t)N * NX, sizeof(double));
  s->du = calloc((size_t)N * NU, sizeof(double));
  s->nl1 = calloc((size_t)N * NX, sizeof(double));
  s->nl2 = calloc((size_t)N * NX, sizeof(double));
  s->dt = calloc((size_t)N * MAXI, sizeof(double));
  s->dnu = calloc((size_t)N * MAXI, sizeof(double));
  double *dx = s->dx, *dc = s->dc, *du = s->du, *nl1 = s->nl1, *nl2 = s->nl2, *dt = s->dt, *dnu = s->dnu, *de = s->de;
  /* second-order correction: residuals of a trial point */
  double* sG1 = calloc((size_t)N * NX, sizeof(double));
  double* sG2 = calloc((size_t)N * NX, sizeof(double));
  double* sR = calloc((size_t)N * MAXI, sizeof(double));

  /* ---- initial point: do_mpc set_initial_guess (all slots = x0, u = 0) or the previous solution ---- */
  if (!warm) {
    for (int k = 0; k <= N; k++) memcpy(it->x + k * NX, x0, sizeof(double) * NX);
    for (int k = 0; k < N; k++) memcpy(it->c + k * NX, x0, sizeof(double) * NX);
  } else {
    memcpy(it->x, X, sizeof(double) * (size_t)(N + 1) * NX);
    memcpy(it->c, C, sizeof(double) * (size_t)N * NX);
    memcpy(it->u, U, sizeof(double) * (size_t)N * NU);
    memcpy(it->l1, L1, sizeof(double) * (size_t)N * NX);
    memcpy(it->l2, L2, sizeof(double) * (size_t)N * NX);
    if (o->warm_shift) { /* option: shift the previous solution by one interval (last interval repeated) */
      for (int k = 0; k + 1 < N; k++) {
        memcpy(it->x + k * NX, X + (k + 1) * NX, sizeof(double) * NX);
        memcpy(it->c + k * NX, C + (k + 1) * NX, sizeof(double) * NX);
        memcpy(it->u + k * NU, U + (k + 1) * NU, sizeof(double) * NU);
        memcpy(it->l1 + k * NX, L1 + (k + 1) * NX, sizeof(double) * NX);
        memcpy(it->l2 + k * NX, L2 + (k + 1) * NX, sizeof(double) * NX);
      }
      memcpy(it->x + (N - 1) * NX, X + N * NX, sizeof(double) * NX);
    }
    memcpy(it->x, x0, sizeof(double) * NX); /* node 0 is the measured state */
    if (warm == 2) memset(it->l1, 0, sizeof(double) * (size_t)N * NX), memset(it->l2, 0, sizeof(double) * (size_t)N * NX);
  }
  /* the solve's own starting point (options.resto_shift_retry) */
  double* Xw = malloc(sizeof(double) * ((size_t)(N + 1) * NX + (size_t)N * NX + (size_t)N * NU));
  double *Cw = Xw + (size_t)(N + 1) * NX, *Uw = Cw + (size_t)N * NX;
  memcpy(Xw, it->x, sizeof(double) * (size_t)(N + 1) * NX), memcpy(Cw, it->c, sizeof(double) * (size_t)N * NX);
  memcpy(Uw, it->u, sizeof(double) * (size_t)N * NU);
  double mu = (warm == 1 && o->mu_init_warm > 0) ? o->mu_init_warm : o->mu_init;
  Tl.eps_s = SMOOTHING(mu);
  mu *= (o->soft_rho > RHO_UNIT ? o->soft_rho / RHO_UNIT : 1.0); /* (penalty scale S below: soft_rho > RHO_UNIT runs in scaled units) */
  s->mu = mu;
  init_slacks(s);
  st->n_reg = 0, st->n_lsfail = 0, st->n_soc = 0, st->n_resto = 0, st->n_fallback = 0, st->n_shift = 0, st->rd_max = 0.0, st->mu_stay_max = 0, st->trig = 0;

  /* filter */
  double filt_th[FILTER_MAX], filt_ph[FILTER_MAX];
  int nfilt = 0;
  double theta0 = -1, theta_max = 0, theta_min = 0;
  double delta_w_last = 0.0;
  int status = LTOMPC_STATUS_MAX_ITER, iter = 0, n_acc = 0;
  double E0 = INFINITY, obj = 0;
  double force_reg = 0.0;
  /* restoration (elastic mode): 0 = not entered, 1 = solving the elastic problem, 2 = back on the hard constraints */
  int resto = 0;
  /* Penalty scale of the restoration phase (escalation, include/ltompc.h resto_rho_max): the elastic problem is
   * f + S resto_rho violation, and everything that has the units of the objective - KKT tolerances, barrier parameter,
   * regularisation, the objective side of the filter - is taken in the units of  f / S + resto_rho violation,  the problem
   * the solver already handles at S = 1 (at S = 1e4 the unscaled tolerance 1e-8 on gradients of size 1e7 is below the
   * rounding floor).  S = 1 outside an escalated restoration: x / 1.0 and x * 1.0 are exact, nothing changes there. */
#define PEN_SCALE(rho_) ((rho_) > RHO_UNIT ? (rho_) / RHO_UNIT : 1.0)
  double S = PEN_SCALE(it->rho); /* (soft_rho) */
  const int resto_allowed = o->resto_rho > 0 && !(o->soft_rho > 0);
  if (start_elastic && resto_allowed && warm) { /* (slacks and elastic variables below: init_slacks with rho = resto_rho) */
    /* start_elastic = 2 (options.infeasible_sticky): the solve before this one ended INFEASIBLE, at the largest penalty; this one
     * starts where that one ended - in the escalated elastic problem, from its primal point and its multipliers */
    const double rho_start = start_elastic == 2 ? fmax(o->resto_rho, o->resto_rho_max) : o->resto_rho;
    resto = 1, st->n_resto = 1;
    it_set_rho(it, rho_start, p), it_set_rho(tr, rho_start, p);
    S = PEN_SCALE(it->rho);
    mu = s->mu = mu * S; /* (the barrier parameter chosen above is in scaled units; soft_rho = 0 here) */
    init_slacks(s);
  }

  double eps_next = Tl.eps_s;
  int n_tiny = 0;
  /* options.warm_fallback_iter: iterations since the barrier parameter last decreased, in a solve that started at mu_init_warm */
  int shift_retried = 0, stuck = 0;
  int since_mu = 0, fallback_armed = (warm == 1 && o->mu_init_warm > 0 && o->warm_fallback_iter > 0);
  for (iter = 0;; iter++) {
    /* table smoothing follows the barrier parameter with a lag of one iteration (so that one linearisation
     * serves the whole iteration, also when mu is reduced in it); the filter restarts when it changes */
    if (eps_next != Tl.eps_s) Tl.eps_s = eps_next, nfilt = 0, theta0 = -1;
    /* ---- linearise ---- */
    for (int k = 0; k < N; k++) linearise_slot(p, o, T, bd, it, k, mu, &L[k]);
    /* ---- KKT error (IPOPT eq. (5)/(6)) ---- */
    double rd = 0, rp = 0, rc_mu = 0, rc_0 = 0, sum_mult = 0, e_max = 0;
    int rd_k = -1, rd_v = -1;
    int n_mult = 0;
    obj = cost_val(p, T, it->x, 0);
    for (int k = 0; k < N; k++) {
      const double *l1 = it->l1 + k * NX, *l2 = it->l2 + k * NX;
      const double* v = k ? it->u + (k - 1) * NU : uprev;
      obj += L[k].cost;
      if (L[k].act[ni - 1])
        for (int q = 0; q < bd->nnl; q++)
          if (it->pen[q] > 0) {
            obj += it->pen[q] * it->e[k * NNLT + q];
            if (q < NNL) e_max = fmax(e_max, it->e[k * NNLT + q]); /* (the restoration phase is about the track constraints) */
          }
      for (int a = 0; a < NX; a++) {
        double rcx = L[k].dc_dual[a] + 4.5 * l2[a];
        double rxp = L[k].dxp_dual[a] - 0.5 * l1[a];
        for (int i = 0; i < NX; i++) rcx += L[k].E1[i * 8 + a] * l1[i], rxp += L[k].E2[i * 8 + a] * l2[i];
        if (k + 1 < N) rxp += 2 * it->l1[(k + 1) * NX + a] - 2 * it->l2[(k + 1) * NX + a];
        if (fabs(rcx) > rd) rd_k = k, rd_v = 100 + a;
        if (fabs(rxp) > rd && fabs(rxp) > fabs(rcx)) rd_k = k, rd_v = a;
        rd = fmax(rd, fmax(fabs(rcx), fabs(rxp)));
        rp = fmax(rp, fmax(fabs(L[k].G1[a]), fabs(L[k].G2[a])));
        sum_mult += fabs(l1[a]) + fabs(l2[a]);
      }
      n_mult += 2 * NX;
      for (int i = 0; i < NU; i++) {
        double uk = it->u[k * NU + i];
        obj += p->r_du[i] * (uk - v[i]) * (uk - v[i]);
        double ru = L[k].du_dual[i] + 2 * p->r_du[i] * (uk - v[i]) + hdt * (l1[6 + i] + l2[6 + i]);
        if (k + 1 < N) ru -= 2 * p->r_du[i] * (it->u[(k + 1) * NU + i] - uk);
        rd = fmax(rd, fabs(ru));
      }
      for (int m = 0; m < ni; m++)
        if (L[k].act[m]) {
          double t = it->t[k * MAXI + m], nu = it->nu[k * MAXI + m];
          rp = fmax(rp, fabs(L[k].h[m] + t));
          rc_mu = fmax(rc_mu, fabs(t * nu - mu));
          rc_0 = fmax(rc_0, fabs(t * nu));
          sum_mult += fabs(nu), n_mult++;
          const int q = m - (ni - bd->nnl);
          if (q >= 0 && it->pen[q] > 0) {
            double ez = it->e[k * NNLT + q] * (it->pen[q] - nu);
            rc_mu = fmax(rc_mu, fabs(ez - mu)), rc_0 = fmax(rc_0, fabs(ez));
            sum_mult += fabs(it->pen[q] - nu), n_mult++;
          }
        }
    }
    if (rd / S > st->rd_max) st->rd_max = rd / S;
    double s_d = fmax(o->s_max, sum_mult / S / n_mult) / o->s_max;
    E0 = fmax(fmax(rd / S / s_d, rp), rc_0 / S / s_d);
    double Emu = fmax(fmax(rd / S / s_d, rp), rc_mu / S / s_d);
    int term = -1;
    if (!isfinite(E0)) term = LTOMPC_STATUS_NUMERICAL;
    else if (E0 <= o->tol) term = LTOMPC_STATUS_SOLVED;
    else {
      /* (an escalated elastic problem only has to answer "is some elastic variable > 0 at the least violation": the
       *  acceptable level decides that at once, as IPOPT's restoration phase does not iterate to the NLP's tolerance either; at
       *  S = 1e4 the unscaled dual residual sits at the rounding floor, 1e-4 on multipliers of 1e7, i.e. 1e-8 scaled) */
      if (E0 <= o->acceptable_tol) { if (++n_acc >= o->acceptable_iter || (resto == 1 && S > 1.0)) term = LTOMPC_STATUS_ACCEPTABLE; }
      else n_acc = 0;
      if (term < 0 && iter >= o->max_iter) term = LTOMPC_STATUS_MAX_ITER;
    }
    if (resto == 1 && (term == LTOMPC_STATUS_SOLVED || term == LTOMPC_STATUS_ACCEPTABLE)) {
      /* The elastic problem has converged.  All elastic variables at (numerically) zero: its solution is a KKT point
       * of the hard-constrained NLP with the same multipliers (nu < rho); back to the hard constraints, where the
       * termination test is repeated on the hard problem's own KKT error.  Otherwise the violation cannot be removed
       * locally: a stationary point of the infeasibility. */
      const double e_tol = term == LTOMPC_STATUS_SOLVED ? o->tol : o->acceptable_tol;
      if (e_max <= e_tol) {
        resto = 2, it_set_rho(it, 0.0, p), it_set_rho(tr, 0.0, p);
        S = 1.0; /* (the barrier parameter stays where it is, mu_min * S_old or above: the hard problem goes on from there) */
        n_acc = 0, nfilt = 0, theta0 = -1;
        if (getenv("ORACLE_TRACE")) fprintf(stderr, "it %3d elastic problem converged, e_max %.2e: back to the hard constraints\n", iter, e_max);
        iter--; /* (this pass only switched the problem: not an iteration) */
        continue;
      }
      /* Some elastic variable stays > tol: at THIS penalty violating is cheaper than complying (that constraint's multiplier
       * sits at the penalty), which a feasible NLP with multipliers > rho shows as well.  Escalate the penalty of this
       * instance and solve the elastic problem again from the current primal point (re-centred exactly as at the entry of the
       * phase); INFEASIBLE only at the largest penalty: a stationary point of objective / rho_max + violation. */
      if (o->resto_rho_factor > 1.0 && it->rho < o->resto_rho_max) {
        const double rho_new = fmin(it->rho * o->resto_rho_factor, o->resto_rho_max);
        if (getenv("ORACLE_TRACE")) fprintf(stderr, "it %3d elastic problem converged with e_max %.2e at rho %.0e: penalty -> %.0e\n", iter, e_max, it->rho, rho_new);
        it_set_rho(it, rho_new, p), it_set_rho(tr, rho_new, p);
        S = PEN_SCALE(rho_new);
        memset(it->l1, 0, sizeof(double) * (size_t)N * NX), memset(it->l2, 0, sizeof(double) * (size_t)N * NX);
        mu = s->mu = o->mu_init * S;
        Tl.eps_s = eps_next = SMOOTHING(mu / S);
        init_slacks(s);
        nfilt = 0, theta0 = -1, delta_w_last = 0.0, force_reg = 0.0, n_tiny = 0, n_acc = 0, since_mu = 0;
        st->n_resto++;
        iter--; /* (this pass only changed the problem: not an iteration) */
        continue;
      }
      term = LTOMPC_STATUS_INFEASIBLE;
    }
    if (term >= 0) { status = term; break; }
    /* ---- barrier update (monotone, IPOPT eq. (7)); slot derivatives depend on mu only via gradients ---- */
    int mu_changed = 0;
    while (Emu <= o->kappa_eps * (mu / S) && mu / S > o->mu_min) {
      double mn = fmax(o->mu_min, fmin(o->kappa_mu * (mu / S), pow(mu / S, o->theta_mu)));
      mu = mn * S, mu_changed = 1;
      Emu = fmax(fmax(rd / S / s_d, rp), 0.0);
      double rcm = 0;
      for (int k = 0; k < N; k++)
        for (int m = 0; m < ni; m++)
          if (L[k].act[m]) {
            rcm = fmax(rcm, fabs(it->t[k * MAXI + m] * it->nu[k * MAXI + m] - mu));
            const int q = m - (ni - bd->nnl);
            if (q >= 0 && it->pen[q] > 0) rcm = fmax(rcm, fabs(it->e[k * NNLT + q] * (it->pen[q] - it->nu[k * MAXI + m]) - mu));
          }
      Emu = fmax(Emu, rcm / S / s_d);
    }
    s->mu = mu;
    since_mu = mu_changed ? 0 : since_mu + 1;
    if (since_mu > st->mu_stay_max) st->mu_stay_max = since_mu;
    /* options.max_mu_stay (warm-started solves): this many iterations without a decrease of the barrier parameter - the iterates
     * wander or cycle (the filter holds FILTER_MAX pairs; a solve that converges needs at most ~60 on this NLP).  On the hard
     * constraints the recovery steps take over at the end of this iteration, elsewhere the solve ends STALLED. */
    stuck = warm && o->max_mu_stay > 0 && since_mu >= o->max_mu_stay;
    if (stuck && !(resto_allowed && resto == 0)) { status = LTOMPC_STATUS_STALLED; break; }
    if (fallback_armed && since_mu >= o->warm_fallback_iter) {
      /* the solve started at the small barrier parameter of a tuned warm start and is going nowhere: once, start again
       * from the current primal point the way a solve after a failed one starts (multipliers 0, barrier at mu_init) */
      fallback_armed = 0, since_mu = 0;
      memset(it->l1, 0, sizeof(double) * (size_t)N * NX), memset(it->l2, 0, sizeof(double) * (size_t)N * NX);
      mu = s->mu = o->mu_init * S;
      Tl.eps_s = eps_next = SMOOTHING(mu / S);
      init_slacks(s);
      nfilt = 0, theta0 = -1, delta_w_last = 0.0, force_reg = 0.0, n_tiny = 0, n_acc = 0;
      st->n_fallback++;
      if (getenv("ORACLE_TRACE")) fprintf(stderr, "it %3d no barrier update for %d iterations after a tuned warm start: restart at mu_init\n", iter, o->warm_fallback_iter);
      iter--; /* (this pass only re-initialised the point: not an iteration) */
      continue;
    }
    if (mu_changed) {
      eps_next = SMOOTHING(mu / S);
      for (int k = 0; k < N; k++) slot_gradients(bd, it, k, mu, &L[k]);
      nfilt = 0, theta0 = -1; /* filter reset */
    }
    double tau = fmax(o->tau_min, 1.0 - mu / S);

    /* ---- condensing of the collocation point ---- */
    for (int k = 0; k < N; k++) {
      if (condense_matrices(s, k)) { status = LTOMPC_STATUS_NUMERICAL; goto done; }
      condense_rhs(s, k);
    }

    /* ---- Riccati sweep, retried with Hessian regularisation ---- */
    double delta_w = force_reg;
    /* Deviation from IPOPT's Algorithm IC: while the previous iteration needed a regularisation larger than
     * DW_KEEP the first attempt already uses delta_w_last / 3 instead of 0 (every attempt is a full sweep here;
     * this halves the number of sweeps at the same iteration counts).  delta_w decays by 3 per iteration and
     * returns to exactly 0 below DW_KEEP, so the final Newton iterations are unregularised. */
    if (delta_w == 0.0 && delta_w_last > DW_KEEP * S) delta_w = delta_w_last / 3.0;
    int tries = 0;
    while (!riccati_sweep(s, delta_w, 0)) {
      /* IPOPT-style inertia correction schedule (Waechter-Biegler Alg. IC) */
      if (delta_w == 0.0) delta_w = delta_w_last == 0.0 ? o->delta_w_first * S : fmax(1e-20, delta_w_last / 3.0);
      else delta_w *= (delta_w_last == 0.0 ? 100.0 : 8.0);
      st->n_reg++;
      if (++tries > 40 || delta_w > 1e20) {
        status = LTOMPC_STATUS_NUMERICAL;
        goto done;
      }
    }
    if (delta_w > 0) delta_w_last = delta_w;
    if (delta_w_last <= DW_KEEP * S) delta_w_last = 0.0;

    forward_recover(s, delta_w);
    if (getenv("ORACLE_CHECK")) { /* residual of the linear KKT system the sweep is supposed to solve */
      double ra = 0, rb = 0, rcc = 0, rdd = 0, re = 0;
      for (int k = 0; k < N; k++) {
        for (int i = 0; i < 8; i++) {
          double a = L[k].G1[i] - 0.5 * dx[(k + 1) * NX + i] + 2 * dx[k * NX + i] + (i >= 6 ? hdt * du[k * NU + i - 6] : 0);
          double b = L[k].G2[i] + 4.5 * dc[k * NX + i] - 2 * dx[k * NX + i] + (i >= 6 ? hdt * du[k * NU + i - 6] : 0);
          double c = L[k].gc[i] + 4.5 * nl2[k * NX + i];
          double d = L[k].gxp[i] - 0.5 * nl1[k * NX + i] + delta_w * dx[(k + 1) * NX + i];
          if (k + 1 < N) d += 2 * nl1[(k + 1) * NX + i] - 2 * nl2[(k + 1) * NX + i];
          for (int j = 0; j < 8; j++) {
            a += L[k].E1[i * 8 + j] * dc[k * NX + j];
            b += L[k].E2[i * 8 + j] * dx[(k + 1) * NX + j];
            c += L[k].Hc[i * 8 + j] * dc[k * NX + j] + L[k].E1[j * 8 + i] * nl1[k * NX + j];
            d += L[k].Hxp[i * 8 + j] * dx[(k + 1) * NX + j] + L[k].E2[j * 8 + i] * nl2[k * NX + j];
          }
          ra = fmax(ra, fabs(a)), rb = fmax(rb, fabs(b)), rcc = fmax(rcc, fabs(c)), rdd = fmax(rdd, fabs(d));
        }
        const double* v = k ? it->u + (k - 1) * NU : uprev;
        for (int i = 0; i < 2; i++) {
          double r2 = 2 * p->r_du[i];
          double e = (L[k].Du[i] + r2 + delta_w) * du[k * NU + i] + L[k].gub[i] + r2 * (it->u[k * NU + i] - v[i]) + hdt * (nl1[k * NX + 6 + i] + nl2[k * NX + 6 + i]);
          if (k > 0) e -= r2 * du[(k - 1) * NU + i];
          if (k + 1 < N) e += r2 * du[k * NU + i] - r2 * du[(k + 1) * NU + i] - r2 * (it->u[(k + 1) * NU + i] - it->u[k * NU + i]);
          re = fmax(re, fabs(e));
        }
      }
      fprintf(stderr, "   KKT-lin residuals: G1 %.2e G2 %.2e c %.2e x+ %.2e u %.2e\n", ra, rb, rcc, rdd, re);
    }
    double a_pri, a_dua, gphi_d;
    slack_steps(s, tau, &a_pri, &a_dua, &gphi_d);

    /* ---- filter line search over alpha = a_pri * 2^-l, second-order correction after a rejected full step ---- */
    double th0, co0, sl0;
    eval_residuals(p, o, T, bd, it, uprev, NULL, NULL, NULL, &th0, &co0, &sl0);
    double ph0 = co0 - mu * sl0;
    if (theta0 < 0) {
      theta0 = th0;
      theta_max = 1e4 * fmax(1.0, theta0), theta_min = 1e-4 * fmax(1.0, theta0);
      nfilt = 0;
    }
    const double g_th = 1e-5, g_ph = 1e-8, eta_ph = 1e-8, s_th = 1.1, s_ph = 2.3, dlt = 1.0, kappa_soc = 0.99;
    int accepted = 0, used_soc = 0;
    double alpha = a_pri;
    /* acceptance test of IPOPT's filter line search for a trial point with measures (th, ph); a_test: the step size
     * in the switching / Armijo conditions (the uncorrected one also for corrected steps) */
#define ACCEPTABLE(th, ph, a_test, ok_out)                                                                      \
  do {                                                                                                          \
    ok_out = 0;                                                                                                 \
    if (!isfinite(th) || !isfinite(ph) || th > theta_max) break;                                                \
    int in_filter_ = 0;                                                                                         \
    for (int f = 0; f < nfilt; f++)                                                                             \
      if (th >= filt_th[f] && ph >= filt_ph[f]) { in_filter_ = 1; break; }                                      \
    if (in_filter_) break;                                                                                      \
    int sw_ = (gphi_d < 0) && ((a_test) * pow(-gphi_d / S, s_ph) > dlt * pow(th0, s_th));                       \
    int armijo_ = ph <= ph0 + eta_ph * (a_test) * gphi_d;                                                       \
    int ok_;                                                                                                    \
    if (th0 <= theta_min && sw_) ok_ = armijo_;                                                                 \
    else ok_ = (th <= (1 - g_th) * th0) || (ph <= ph0 - g_ph * S * th0);                                        \
    if (!ok_) break;                                                                                            \
    if (!(sw_ && armijo_)) { /* augment filter */                                                               \
      if (nfilt == FILTER_MAX) { memmove(filt_th, filt_th + 1, sizeof(double) * (FILTER_MAX - 1)); memmove(filt_ph, filt_ph + 1, sizeof(double) * (FILTER_MAX - 1)); nfilt--; } \
      filt_th[nfilt] = (1 - g_th) * th0, filt_ph[nfilt] = ph0 - g_ph * S * th0, nfilt++;                        \
    }                                                                                                           \
    ok_out = 1;                                                                                                 \
  } while (0)
    for (int l = 0; l < o->n_linesearch; l++, alpha *= 0.5) {
      make_trial(s, alpha);
      double th, co, sl;
      eval_residuals(p, o, T, bd, tr, uprev, sG1, sG2, sR, &th, &co, &sl);
      double ph = co - mu * sl;
      ACCEPTABLE(th, ph, alpha, accepted);
      if (accepted) {
        if (getenv("ORACLE_LSHIST")) fprintf(stderr, "LS %d\n", l);
        break;
      }
      if (l == 0 && o->max_soc > 0 && isfinite(th) && th >= th0) {
        /* ---- second-order correction (Waechter & Biegler 2006, A-5.5 .. A-5.9): the full step was rejected and did not
         *      reduce the infeasibility.  c_soc = alpha c(x) + c(x + alpha d); same KKT matrix, new right-hand side.
         *      The uncorrected step is kept aside: the backtracking continues with it when the correction fails. */
        const size_t nX = (size_t)(N + 1) * NX, nC = (size_t)N * NX, nU = (size_t)N * NU, nT = (size_t)N * MAXI, nE = (size_t)N * NNL;
        double* keep = malloc(sizeof(double) * (nX + 3 * nC + nU + 2 * nT + nE));
        double *k_dx = keep, *k_dc = k_dx + nX, *k_du = k_dc + nC, *k_l1 = k_du + nU, *k_l2 = k_l1 + nC, *k_dt = k_l2 + nC, *k_dn = k_dt + nT, *k_de = k_dn + nT;
        memcpy(k_dx, dx, sizeof(double) * nX), memcpy(k_dc, dc, sizeof(double) * nC), memcpy(k_du, du, sizeof(double) * nU);
        memcpy(k_l1, nl1, sizeof(double) * nC), memcpy(k_l2, nl2, sizeof(double) * nC);
        memcpy(k_dt, dt, sizeof(double) * nT), memcpy(k_dn, dnu, sizeof(double) * nT), memcpy(k_de, de, sizeof(double) * nE);
        double th_old = th0, a_soc = alpha, a_dua_soc = a_dua;
        int soc_ok = 0;
        for (int c = 0; c < o->max_soc; c++) {
          /* c_soc <- a_soc c_soc + c(trial)  (c_soc = c(x) before the first correction) */
          for (int k = 0; k < N; k++) {
            for (int i = 0; i < NX; i++)
              L[k].rG1[i] = a_soc * L[k].rG1[i] + sG1[k * NX + i], L[k].rG2[i] = a_soc * L[k].rG2[i] + sG2[k * NX + i];
            for (int m = 0; m < ni; m++)
              if (L[k].act[m]) L[k].rI[m] = a_soc * L[k].rI[m] + sR[k * MAXI + m];
            slot_gradients(bd, it, k, mu, &L[k]);
          }
          for (int k = 0; k < N; k++) condense_rhs(s, k);
          riccati_sweep(s, delta_w, 1);
          forward_recover(s, delta_w);
          double gphi_soc;
          slack_steps(s, tau, &a_soc, &a_dua_soc, &gphi_soc);
          make_trial(s, a_soc);
          double th_s, co_s, sl_s;
          eval_residuals(p, o, T, bd, tr, uprev, sG1, sG2, sR, &th_s, &co_s, &sl_s);
          double ph_s = co_s - mu * sl_s;
          st->n_soc++;
          ACCEPTABLE(th_s, ph_s, alpha, soc_ok);
          if (getenv("ORACLE_TRACE")) fprintf(stderr, "     soc %d: a_soc %.4f theta %.3e -> %.3e (start %.3e) phi %.6e ok %d\n", c, a_soc, th, th_s, th0, ph_s, soc_ok);
          if (soc_ok || !isfinite(th_s) || th_s > kappa_soc * th_old) break;
          th_old = th_s;
        }
        if (soc_ok) {
          accepted = 1, used_soc = 1, alpha = a_soc, a_dua = a_dua_soc;
          free(keep);
          break;
        }
        /* back to the uncorrected step and residuals */
        memcpy(dx, k_dx, sizeof(double) * nX), memcpy(dc, k_dc, sizeof(double) * nC), memcpy(du, k_du, sizeof(double) * nU);
        memcpy(nl1, k_l1, sizeof(double) * nC), memcpy(nl2, k_l2, sizeof(double) * nC);
        memcpy(dt, k_dt, sizeof(double) * nT), memcpy(dnu, k_dn, sizeof(double) * nT), memcpy(de, k_de, sizeof(double) * nE);
        free(keep);
        for (int k = 0; k < N; k++) {
          memcpy(L[k].rG1, L[k].G1, sizeof L[k].G1), memcpy(L[k].rG2, L[k].G2, sizeof L[k].G2);
          for (int m = 0; m < ni; m++) L[k].rI[m] = L[k].act[m] ? L[k].h[m] + it->t[k * MAXI + m] : 0.0;
          slot_gradients(bd, it, k, mu, &L[k]);
        }
      }
    }
    (void)used_soc;
    if (!accepted) {
      st->n_lsfail++;
      if (resto_allowed && resto == 0) {
      enter_resto:
        if (warm && !o->warm_shift && o->resto_shift_retry && !shift_retried) {
          /* ---- first remedy: the jam may be the un-shifted warm start's doing (do_mpc re-uses the previous solution as it
           *      is, one interval behind the new measured state; measured on the reference's own loop: two of the seven ticks
           *      that ended INFEASIBLE in round 2 are solved in 14 iterations from the shifted point).  Once per solve: start
           *      again on the hard constraints from the solve's own starting point moved one interval ahead (options.warm_shift's
           *      rule: last interval repeated), multipliers 0, barrier at mu_init.  The restoration phase proper follows if
           *      this start jams too. */
          shift_retried = 1, st->n_shift++;
          for (int k = 0; k + 1 < N; k++) {
            memcpy(it->x + k * NX, Xw + (k + 1) * NX, sizeof(double) * NX);
            memcpy(it->c + k * NX, Cw + (k + 1) * NX, sizeof(double) * NX);
            memcpy(it->u + k * NU, Uw + (k + 1) * NU, sizeof(double) * NU);
          }
          memcpy(it->x + (N - 1) * NX, Xw + N * NX, sizeof(double) * NX);
          memcpy(it->x + N * NX, Xw + N * NX, sizeof(double) * NX);
          memcpy(it->c + (N - 1) * NX, Cw + (N - 1) * NX, sizeof(double) * NX);
          memcpy(it->u + (N - 1) * NU, Uw + (N - 1) * NU, sizeof(double) * NU);
          memcpy(it->x, x0, sizeof(double) * NX);
          memset(it->l1, 0, sizeof(double) * (size_t)N * NX), memset(it->l2, 0, sizeof(double) * (size_t)N * NX);
          mu = s->mu = o->mu_init;
          Tl.eps_s = eps_next = SMOOTHING(mu);
          init_slacks(s);
          nfilt = 0, theta0 = -1, delta_w_last = 0.0, force_reg = 0.0, n_tiny = 0, n_acc = 0, since_mu = 0;
          if (getenv("ORACLE_TRACE")) fprintf(stderr, "it %3d -> start again from the shifted warm start\n", iter);
          continue;
        }
        /* ---- restoration phase, as an elastic mode: the track constraints get elastic variables that
         *      cost resto_rho each; equality multipliers, slacks and the barrier parameter start again at the current
         *      primal point.  (IPOPT: min rho ||c(x)||_1 + zeta/2 ||D(x - x_R)||^2 over all constraints, then back to
         *      the original problem; here the objective stays, so the elastic problem's solution with e = 0 already
         *      is the solution.) */
        resto = 1, st->n_resto++;
        it_set_rho(it, o->resto_rho, p), it_set_rho(tr, o->resto_rho, p);
        S = PEN_SCALE(it->rho);
        memset(it->l1, 0, sizeof(double) * (size_t)N * NX), memset(it->l2, 0, sizeof(double) * (size_t)N * NX);
        mu = s->mu = o->mu_init * S;
        Tl.eps_s = eps_next = SMOOTHING(mu / S);
        init_slacks(s);
        nfilt = 0, theta0 = -1, delta_w_last = 0.0, force_reg = 0.0, n_tiny = 0, n_acc = 0, since_mu = 0;
        if (getenv("ORACLE_TRACE")) fprintf(stderr, "it %3d -> restoration (elastic mode, rho %.0f)\n", iter, o->resto_rho);
        continue;
      }
      /* no (further) restoration: retry this iterate with a (larger) forced regularisation; after a few
       * failures take the smallest step and reset the filter so that the iteration cannot stall */
      if (o->max_ls_fail > 0 && st->n_lsfail >= o->max_ls_fail) { status = LTOMPC_STATUS_STALLED; break; }
      if (force_reg < 1e4 * S) { force_reg = force_reg == 0 ? 1e-2 * S : force_reg * 100; continue; }
      nfilt = 0;
      alpha = a_pri * pow(0.5, o->n_linesearch - 1);
    }
    force_reg = 0.0;
    /* consecutive tiny steps: IPOPT would enter restoration */
    n_tiny = alpha <= 1e-3 ? n_tiny + 1 : 0;
    /* options.dual_inf_max: on the hard constraints a dual infeasibility that has grown beyond any scale of the problem (the
     * multipliers diverge while the line search keeps accepting steps of a percent: 100+ iterations before it finally fails)
     * is treated like that failure at once: shifted restart, then the restoration phase.  Warm-started solves only: from do_mpc's
     * cold start (every node = x0) a converging solve passes through dual infeasibilities of 1e7. */
    const int blowup = stuck || (o->dual_inf_max > 0 && warm && resto_allowed && resto == 0 && rd > o->dual_inf_max);
    if (blowup) st->trig |= (shift_retried ? 2 : 1);
    if (blowup) goto enter_resto;
    if (o->stall_iter > 0 && n_tiny >= o->stall_iter) {
      if (resto_allowed && resto == 0) goto enter_resto;
      status = LTOMPC_STATUS_STALLED;
      break;
    }
    if (getenv("ORACLE_TRACE"))
      fprintf(stderr, "it %3d mu %.2e E0 %.3e rd %.2e rp %.2e rc %.2e th0 %.3e ph0 %.6e a_pri %.3f a_dua %.3f alpha %.4f dw %.1e gphid %.3e acc %d obj %.6f u0 %.8f %.8f rdk %d rdv %d mu_k %.3e s_k %.4f nfilt %d thmin %.2e\n",
              iter, mu, E0, rd, rp, rc_0, th0, ph0, a_pri, a_dua, alpha, delta_w, gphi_d, accepted, obj, it->u[0], it->u[1], rd_k, rd_v, rd_k >= 0 ? it->x[(rd_k + 1) * NX + 2] : 0.0, rd_k >= 0 ? it->x[(rd_k + 1) * NX] : 0.0, nfilt, theta_min);
    /* ---- take the step ---- */
    for (int k = 1; k <= N; k++)
      for (int i = 0; i < NX; i++) it->x[k * NX + i] += alpha * dx[k * NX + i];
    for (int k = 0; k < N; k++) {
      for (int i = 0; i < NX; i++) {
        it->c[k * NX + i] += alpha * dc[k * NX + i];
        it->l1[k * NX + i] += alpha * (nl1[k * NX + i] - it->l1[k * NX + i]);
        it->l2[k * NX + i] += alpha * (nl2[k * NX + i] - it->l2[k * NX + i]);
      }
      for (int i = 0; i < NU; i++) it->u[k * NU + i] += alpha * du[k * NU + i];
      for (int m = 0; m < ni; m++) {
        if (!L[k].act[m]) continue;
        double t = it->t[k * MAXI + m] + alpha * dt[k * MAXI + m];
        double nu = it->nu[k * MAXI + m] + a_dua * dnu[k * MAXI + m];
        /* IPOPT eq. (16): keep nu within [mu/(kS t), kS mu/t], kS = 1e10 */
        double lo = mu / (1e10 * t), hi = 1e10 * mu / t;
        it->t[k * MAXI + m] = t, it->nu[k * MAXI + m] = nu < lo ? lo : (nu > hi ? hi : nu);
        if (m >= ni - bd->nnl) it->e[k * NNLT + m - (ni - bd->nnl)] += alpha * de[k * NNLT + m - (ni - bd->nnl)];
      }
    }
  }
done:
  memcpy(X, it->x, sizeof(double) * (size_t)(N + 1) * NX);
  memcpy(C, it->c, sizeof(double) * (size_t)N * NX);
  memcpy(U, it->u, sizeof(double) * (size_t)N * NU);
  memcpy(L1, it->l1, sizeof(double) * (size_t)N * NX);
  memcpy(L2, it->l2, sizeof(double) * (size_t)N * NX);
  if (Tout && NUout)
    for (int k = 0; k < N; k++)
      for (int m = 0; m < ni; m++) Tout[k * ni + m] = it->t[k * MAXI + m], NUout[k * ni + m] = it->nu[k * MAXI + m];
  st->viol = 0.0;
  if (it->rho > 0)
    for (int k = 0; k + 1 < N; k++)
      for (int q = 0; q < NNL; q++) st->viol = fmax(st->viol, it->e[k * NNLT + q]);
  st->g0 = -INFINITY, st->status_solver = status, st->rho_end = it->rho;
  if (o->node0_check && !(o->soft_rho > 0)) {
    /* do_mpc registers the track constraints at node 0 as well (controller.py:69-70): rows that only involve the
     * measured state.  They cannot change the minimiser, but a measured state outside the band leaves the reference's NLP
     * without a feasible point: a converged status becomes INFEASIBLE (violation g(x0)); when the violation is below the
     * acceptable level SOLVED becomes ACCEPTABLE (IPOPT's error cannot fall below it). */
    tables_t Te = *T0;
    Te.eps_s = SMOOTHING(0.0), Te.eps_mu = 0.0;
    double g[NNL];
    cons_val(p, &Te, x0, g);
    st->g0 = fmax(g[0], fmax(g[1], g[2]));
    if (status == LTOMPC_STATUS_SOLVED || status == LTOMPC_STATUS_ACCEPTABLE) {
      if (st->g0 > o->acceptable_tol) status = LTOMPC_STATUS_INFEASIBLE, st->viol = st->g0;
      else if (st->g0 > o->tol) status = LTOMPC_STATUS_ACCEPTABLE;
    }
  }
  st->status = status, st->iters = iter, st->kkt = E0, st->obj = obj, st->mu = mu;
  it_free(it), it_free(tr);
  free(s->L), free(s->W), free(dx), free(dc), free(du), free(nl1), free(nl2), free(dt), free(dnu), free(de);
  free(sG1), free(sG2), free(sR), free(Xw);
  return 0;
}

/* ------------------------------------------------------------------ exported entry points */
void oracle_default_params(ltompc_params* p) {
  memset(p, 0, sizeof *p);
  /* data/vehicles/MX5.json as read by model.py:42-64; D_f, D_r keep the constructor default 1.0 */
  p->mass = 1000.0, p->inertia_z = 1000.0, p->length_f = 1.5, p->length_r = 1.5, p->width = 2.3;
  p->B_f = 10.0, p->C_f = 1.3, p->D_f = 1.0, p->B_r = 12.0, p->C_r = 1.2, p->D_r = 1.0;
  p->C_m = 1000.0, p->Cr_0 = 0.01, p->Cr_2 = 0.0003, p->gravity = 9.81;
  p->q_n = 0.5, p->q_mu = 3.0, p->q_vy = 1.0, p->q_v = 1.0, p->vref_scale = 0.6, p->q_B = 1e-2;
  p->r_du[0] = p->r_du[1] = 1e-2;
  for (int i = 0; i < NX; i++) p->x_lb[i] = -LTOMPC_NO_BOUND, p->x_ub[i] = LTOMPC_NO_BOUND;
  p->x_lb[0] = 0.0;
  p->x_lb[2] = -M_PI * 0.5, p->x_ub[2] = M_PI * 0.5;
  p->x_lb[3] = 0.0;
  p->x_lb[6] = -M_PI / 4, p->x_ub[6] = M_PI / 4;
  p->x_lb[7] = -1, p->x_ub[7] = 1;
  p->u_lb[0] = -2 * M_PI / 4, p->u_ub[0] = 2 * M_PI / 4;
  p->u_lb[1] = -1, p->u_ub[1] = 1;
}
void oracle_default_options(ltompc_options* o) {
  memset(o, 0, sizeof *o);
  o->t_step = 0.1, o->tol = 1e-8, o->acceptable_tol = 1e-6, o->mu_init = 0.1, o->mu_min = 1e-9;
  o->kappa_eps = 10, o->kappa_mu = 0.2, o->theta_mu = 1.5, o->tau_min = 0.99, o->bound_push = 1e-2;
  o->s_max = 100, o->delta_w_first = 1e-4, o->smooth_eps_min = 1e-4, o->smooth_scale = 1.0, o->max_iter = 1000, o->acceptable_iter = 15, o->n_linesearch = 8, o->stall_iter = 15, o->max_ls_fail = 8;
  o->resto_rho = 1000.0, o->max_soc = 0, o->resto_sticky = 0;
  o->resto_rho_max = 1e6, o->resto_rho_factor = 1e3, o->node0_check = 1, o->warm_fallback_iter = 25, o->resto_shift_retry = 1;
  o->dual_inf_max = 1e4, o->max_mu_stay = 100, o->infeasible_sticky = 1;
  o->warm_reset_on_fail = 1; /* applied by the caller (oracle.py solve(prev_status=...)): this file sees one solve at a time */
}

int oracle_rhs(const ltompc_params* p, const double* tab, int nt, const double* x, const double* u, double* f) {
  tables_t T = tables_view(tab, nt);
  rhs_val(p, &T, x, u, f);
  return 0;
}
/* fx: 8x8 row-major d f_i / d x_j ; H: sum_i lam_i d2 f_i, 8x8 */
int oracle_rhs_derivs(const ltompc_params* p, const double* tab, int nt, double eps, const double* x, const double* lam,
                      double* f, double* fx, double* H) {
  tables_t T = tables_view(tab, nt);
  T.eps_s = eps;
  rhs_jets F;
  rhs_jet(p, &T, x, &F);
  memset(fx, 0, sizeof(double) * 64), memset(H, 0, sizeof(double) * 64);
  for (int i = 0; i < 6; i++) {
    f[i] = F.f[i].v;
    for (int j = 0; j < NX; j++) fx[i * 8 + j] = F.f[i].g[j];
    for (int a = 0; a < NX; a++)
      for (int b = 0; b < NX; b++) H[a * 8 + b] += lam[i] * F.f[i].h[hidx(a, b)];
  }
  f[6] = f[7] = 0;
  return 0;
}
/* cost value/gradient/Hessian at a node; cons: gL,gR value + gradients + Hessians (2x8, 2x64) */
int oracle_cost_derivs(const ltompc_params* p, const double* tab, int nt, double eps, const double* x, int terminal,
                       double* val, double* grad, double* H) {
  tables_t T = tables_view(tab, nt);
  T.eps_s = eps;
  jet c = cost_jet(p, &T, x, terminal);
  *val = c.v;
  for (int a = 0; a < NX; a++) {
    grad[a] = c.g[a];
    for (int b = 0; b < NX; b++) H[a * 8 + b] = c.h[hidx(a, b)];
  }
  return 0;
}
int oracle_cons_derivs(const ltompc_params* p, const double* tab, int nt, double eps, const double* x, double* val,
                       double* grad, double* H) {
  tables_t T = tables_view(tab, nt);
  T.eps_s = eps;
  jet g[NNL];
  cons_jet(p, &T, x, g);
  for (int q = 0; q < NNL; q++) {
    val[q] = g[q].v;
    for (int a = 0; a < NX; a++) {
      grad[q * 8 + a] = g[q].g[a];
      for (int b = 0; b < NX; b++) H[q * 64 + a * 8 + b] = g[q].h[hidx(a, b)];
    }
  }
  return 0;
}
/* friction-ellipse constraints (front, rear): value + gradients + Hessians (2, 2x8, 2x64) */
int oracle_ell_derivs(const ltompc_params* p, const double* x, double* val, double* grad, double* H) {
  jet g[NEL];
  ell_jet(p, x, g);
  for (int q = 0; q < NEL; q++) {
    val[q] = g[q].v;
    for (int a = 0; a < NX; a++) {
      grad[q * 8 + a] = g[q].g[a];
      for (int b = 0; b < NX; b++) H[q * 64 + a * 8 + b] = g[q].h[hidx(a, b)];
    }
  }
  double v2[NEL];
  ell_val(p, x, v2);
  return (v2[0] == val[0] || fabs(v2[0] - val[0]) <= 1e-12 * fabs(val[0])) ? 0 : -1; /* (the two evaluations agree) */
}
int oracle_slip_forces(const ltompc_params* p, const double* x, int batch, double* alpha, double* Fy) {
  for (int b = 0; b < batch; b++) {
    const double* xb = x + b * NX;
    double af = atan2(xb[4] + p->length_f * xb[5], xb[3]) - xb[6];
    double ar = atan2(xb[4] - p->length_r * xb[5], xb[3]);
    double L = p->length_f + p->length_r;
    double Fnf = p->length_r * p->mass * p->gravity / L, Fnr = p->length_f * p->mass * p->gravity / L;
    alpha[b * 2] = af, alpha[b * 2 + 1] = ar;
    Fy[b * 2] = -Fnf * p->D_f * sin(p->C_f * atan(p->B_f * af));
    Fy[b * 2 + 1] = -Fnr * p->D_r * sin(p->C_r * atan(p->B_r * ar));
  }
  return 0;
}
/* plant: classical RK4, n_sub sub-steps, zero-order-hold input (stands in for CVODES at 1e-10) */
int oracle_plant_step(const ltompc_params* p, const double* tab, int nt, const double* x, const double* u,
                      int batch, double dt, int n_sub, int periodic, double* xn) {
  tables_t T = tables_view(tab, nt);
  if (periodic) T.period = tab[nt - 1] - tab[0];
  double hs = dt / n_sub;
  for (int b = 0; b < batch; b++) {
    double y[NX], k1[NX], k2[NX], k3[NX], k4[NX], z[NX];
    memcpy(y, x + b * NX, sizeof y);
    for (int s = 0; s < n_sub; s++) {
      rhs_val(p, &T, y, u + b * NU, k1);
      for (int i = 0; i < NX; i++) z[i] = y[i] + 0.5 * hs * k1[i];
      rhs_val(p, &T, z, u + b * NU, k2);
      for (int i = 0; i < NX; i++) z[i] = y[i] + 0.5 * hs * k2[i];
      rhs_val(p, &T, z, u + b * NU, k3);
      for (int i = 0; i < NX; i++) z[i] = y[i] + hs * k3[i];
      rhs_val(p, &T, z, u + b * NU, k4);
      for (int i = 0; i < NX; i++) y[i] += hs / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    }
    memcpy(xn + b * NX, y, sizeof y);
  }
  return 0;
}

/* Batched solve.  Arrays are batch-major: X: B x (N+1) x 8, C: B x N x 8, U: B x N x 2, L1/L2: B x N x 8
 * (in: warm start if warm != 0; out: solution).  stats: B x 5 doubles (status, iters, kkt, obj, mu) +
 * 4 ints packed as doubles (n_reg, n_lsfail, n_soc, n_resto) + viol + g0 + n_fallback + n_shift + status_solver + rd_max + mu_stay_max => 16 doubles per instance. */
int oracle_solve_batch(const ltompc_params* p, const ltompc_options* o, const double* tab, int nt, int N, int B,
                       const double* x0, const double* uprev, int warm, double* X, double* C, double* U,
                       double* L1, double* L2, double* u0, double* stats, int nthreads, double* Tout, double* NUout,
                       const int* prev_status, int* sticky) {
  /* prev_status (may be NULL): the solver's own status (stats[13]) of the solve the warm start comes from, per instance (option warm_reset_on_fail).
   * sticky (may be NULL; in/out): option resto_sticky, ticks for which an instance still starts in elastic mode. */
  bounds_t bd0;
  build_bounds(p, &bd0);
  const int ni0 = bd0.ni;
  tables_t T = tables_view(tab, nt);
  if (o->periodic_tables) T.period = tab[nt - 1] - tab[0];
  (void)nthreads;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; b++) {
    solve_stats st;
    int w = warm ? 1 : 0, start_elastic = 0;
    if (warm && prev_status) {
      const int ps = prev_status[b];
      /* with resto_sticky the multipliers of a converged ELASTIC problem (status INFEASIBLE) are re-used like any others */
      const int conv = ps == LTOMPC_STATUS_SOLVED || ps == LTOMPC_STATUS_ACCEPTABLE ||
                       (ps == LTOMPC_STATUS_INFEASIBLE && (o->resto_sticky > 0 || o->infeasible_sticky));
      if (o->warm_reset_on_fail && !conv) w = 2;
      if (ps == LTOMPC_STATUS_INFEASIBLE && o->infeasible_sticky) start_elastic = 2;
    }
    if (warm && sticky && o->resto_sticky > 0 && sticky[b] > 0 && !start_elastic) start_elastic = 1;
    solve_one(p, o, &T, N, x0 + (size_t)b * NX, uprev + (size_t)b * NU, w, start_elastic, X + (size_t)b * (N + 1) * NX,
              C + (size_t)b * N * NX, U + (size_t)b * N * NU, L1 + (size_t)b * N * NX, L2 + (size_t)b * N * NX,
              Tout ? Tout + (size_t)b * N * ni0 : NULL, NUout ? NUout + (size_t)b * N * ni0 : NULL, &st);
    if (sticky && o->resto_sticky > 0) {
      /* an instance that jammed on the hard constraints, or whose horizon problem is infeasible, starts its next
       * resto_sticky solves in elastic mode */
      const int jammed = st.n_resto > 0 && !start_elastic;
      if (st.status_solver == LTOMPC_STATUS_INFEASIBLE || jammed) sticky[b] = o->resto_sticky;
      else if (sticky[b] > 0) sticky[b]--;
    }
    u0[b * NU] = U[(size_t)b * N * NU], u0[b * NU + 1] = U[(size_t)b * N * NU + 1];
    double* s = stats + (size_t)b * 18;
    s[0] = st.status, s[1] = st.iters, s[2] = st.kkt, s[3] = st.obj, s[4] = st.mu, s[5] = st.n_reg, s[6] = st.n_lsfail;
    s[7] = st.n_soc, s[8] = st.n_resto, s[9] = st.viol, s[10] = st.g0, s[11] = st.n_fallback, s[12] = st.n_shift, s[13] = st.status_solver, s[14] = st.rd_max, s[15] = st.mu_stay_max, s[16] = st.trig, s[17] = st.rho_end;
  }
  return 0;
}
/* ------------------------------------------------------------------ velocity-profile generator
 * Restates src/velocity.py:14-76 with src/vehicle.py:24-35 / src/vehicleMX5.py:19-38 the way the reference writes it:
 * explicit rolled (and flipped) copies of s, k, v starting at the slowest point, python's negative index v[i-1] for i = 0.
 * (The HIP kernel does the index arithmetic on the fly instead: two implementations, one fixture.) */
static double vp_engine(const ltompc_vp_vehicle* V, double v) {
  if (V->kind == 1) return (V->T * V->C_m) - V->Cr_0 - (V->Cr_2 * (v * v));
  int n = V->n_map;
  if (v <= V->map_v[0]) return V->map_f[0];
  if (v >= V->map_v[n - 1]) return V->map_f[n - 1];
  int j = 0;
  while (j + 2 < n && v >= V->map_v[j + 1]) j++;
  double slope = (V->map_f[j + 1] - V->map_f[j]) / (V->map_v[j + 1] - V->map_v[j]);
  return slope * (v - V->map_v[j]) + V->map_f[j];
}
static double vp_tract(const ltompc_vp_vehicle* V, double v, double k) {
  const double GRAV = 9.81;
  double f, f_lat;
  if (V->kind == 1) {
    double Fn = V->mass * GRAV;
    f = V->lam * V->D * Fn;
    f_lat = V->mass * v * v * k;
  } else {
    f = V->friction_coef * V->mass * GRAV;
    f_lat = V->mass * (v * v) * k;
  }
  if (f <= f_lat) return 0.0;
  return sqrt(f * f - f_lat * f_lat);
}
int oracle_velocity_profile(const ltompc_vp_vehicle* V, int n, int batch, const double* s_all, const double* k_all, const double* s_max_all,
                            double* v_all, double* vloc_all, double* vacc_all, double* vdec_all) {
  const double GRAV = 9.81;
  double *s = malloc(sizeof(double) * 3 * (size_t)n), *k = s + n, *v = k + n;
  for (int b = 0; b < batch; b++) {
    const double *so = s_all + (size_t)b * n, *ko = k_all + (size_t)b * n;
    double *vloc = vloc_all + (size_t)b * n, *vacc = vacc_all + (size_t)b * n, *vdec = vdec_all + (size_t)b * n;
    const double s_max = s_max_all[b];
    const int closed = s_max >= 0.0;
    int m = 0;
    for (int i = 0; i < n; i++) {
      vloc[i] = sqrt(V->friction_coef * GRAV / ko[i]);
      if (vloc[i] < vloc[m]) m = i;
    }
    /* limit_acceleration: np.roll(a, -m)[j] = a[(j + m) % n] */
    for (int j = 0; j < n; j++) s[j] = so[(j + m) % n], k[j] = ko[(j + m) % n], v[j] = vloc[(j + m) % n];
    for (int i = 0; i < n; i++) {
      int wrap = i == ((n - m) % n), im = i ? i - 1 : n - 1;
      if (wrap && !closed) continue;
      if (v[i] > v[im]) {
        double traction = vp_tract(V, v[im], k[im]), eng = vp_engine(V, v[im]);
        double force = eng < traction ? eng : traction;
        double accel = force / V->mass;
        double ds = wrap ? s_max - s[im] : s[i] - s[im];
        double vlim = sqrt(v[im] * v[im] + 2 * accel * ds);
        v[i] = v[i] < vlim ? v[i] : vlim;
      }
    }
    for (int j = 0; j < n; j++) vacc[(j + m) % n] = v[j];
    /* limit_deceleration: flip(roll(a, -m))[j] = a[(n - 1 - j + m) % n] */
    for (int j = 0; j < n; j++) s[j] = so[(n - 1 - j + m) % n], k[j] = ko[(n - 1 - j + m) % n], v[j] = vloc[(n - 1 - j + m) % n];
    for (int i = 0; i < n; i++) {
      int wrap = i == m, im = i ? i - 1 : n - 1;
      if (wrap && !closed) continue;
      if (v[i] > v[im]) {
        double decel = vp_tract(V, v[im], k[im]) / V->mass;
        double ds = wrap ? s_max - s[i] : s[im] - s[i];
        double vlim = sqrt(v[im] * v[im] + 2 * decel * ds);
        v[i] = v[i] < vlim ? v[i] : vlim;
      }
    }
    for (int j = 0; j < n; j++) vdec[(n - 1 - j + m) % n] = v[j];
    for (int i = 0; i < n; i++) v_all[(size_t)b * n + i] = vacc[i] < vdec[i] ? vacc[i] : vdec[i];
  }
  free(s);
  return 0;
}

int oracle_num_ineq(const ltompc_params* p) {
  bounds_t bd;
  build_bounds(p, &bd);
  return bd.ni;
}
int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
