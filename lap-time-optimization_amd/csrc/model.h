// model.h — device-side model of the MPC hot path, hand-derived analytic first and second derivatives.
//
// What it restates (reference file:line):
//   rhs                 src/mpc/model.py:152-183     (curvilinear dynamic bicycle, 8 states / 2 inputs)
//   slip angles         src/mpc/model.py:101-104     (atan2, not atan(./vx))
//   Pacejka             src/mpc/model.py:106-114     (F_y = -F_N D sin(C atan(B alpha)), note the minus)
//   drivetrain          src/mpc/model.py:116-117,160
//   slip-angle cost     src/mpc/model.py:124-128     (atan(vy/vx) - atan(delta l_r/(l_f+l_r)))
//   stage / terminal cost  src/mpc/controller.py:51-53
//   boundary constraints   src/mpc/model.py:70-84
//   tables              src/path.py:96-101, src/mpc/track.py:30-42  (CasADi linear interpolant, linear extrapolation)
//
// Everything is fp64.  The CPU oracle (oracle/ltompc_oracle.c) obtains the same derivatives by generic
// second-order forward AD; the two implementations are compared in tests/, neither includes the other.
#pragma once
#if defined(LTOMPC_HOST_HARNESS)  // tests/host_harness: the device functions as host C++ under ASan / UBSan (test infrastructure)
#include "hip_shim.h"
#else
#include <hip/hip_runtime.h>
#endif

#include "../../include/ltompc.h"

namespace ltompc {

constexpr int NX = 8;
constexpr int NU = 2;
constexpr int MAX_XB = 16;  // finite state bounds
constexpr int MAX_UB = 4;   // finite input bounds
constexpr int NNL = 3;      // gL, gR+, gR-  (see cons_eval)
constexpr int NEL = 2;      // friction-ellipse constraints of the two axles (ellipse_eval), present when params.ell_penalty > 0
constexpr int MAX_NI = MAX_UB + 2 * MAX_XB + NNL + NEL;

// index of (i,j), i >= j, in a packed lower-triangular symmetric matrix
__host__ __device__ constexpr int sidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// Device pointers are typed as global-address-space pointers in device code: the kernels that read Consts / Work from
// device memory (instead of kernel arguments) would otherwise see generic pointers and use flat_load / flat_store
// (slower, and counted by the LDS wait counter as well).  Same size and layout on the host.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> using gptr = __attribute__((address_space(1))) T*;
#else
template <class T> using gptr = T*;
#endif

struct Tables {
  int n;
  gptr<const double> s_kappa, kappa, s_arc, n_left, n_right, v_ref;
  double g0_kappa, inv_kappa, g0_arc, inv_arc;  // first knot and (n - 1) / span of the two grids: interval estimate of lut_eval
  double period;  // options.periodic_tables: span of the grids (tables evaluated at s modulo the span), else 0
};

struct Bounds {
  int n_xb, n_ub, ni;  // ni = n_ub + 2 n_xb + NNL + nel
  int nel;             // 0, or NEL with the friction-ellipse constraints
  int xb_idx[MAX_XB];
  double xb_sgn[MAX_XB];  // -1: lower bound (h = lb - x), +1: upper (h = x - ub)
  double xb_val[MAX_XB];
  int ub_idx[MAX_UB];
  double ub_sgn[MAX_UB];
  double ub_val[MAX_UB];
};

// ---------------------------------------------------------------------------------------------- tables
// Interval i with grid[i] <= s < grid[i+1], clamped to [0, n-2]: linear extrapolation outside the grid.
// g0 = grid[0] and inv = (n - 1) / (grid[n-1] - grid[0]) come with the tables (Tables::g0_*, inv_*).
//
// value / slope / second derivative of a table.  eps = 0: exact piece-wise linear.  eps > 0: each interior
// knot's kink (J/2)|z| is replaced on |z| < W (W = half the shorter adjacent interval) by the C1 patch
// (J/2)(sqrt(z^2+eps^2) + a z^2 + b) with k(W) = W, k'(W) = 1  (DESIGN.md "non-smoothness").
//
// Memory: ONE round trip in the common case.  The interval is estimated from the (nearly) uniform spacing and the four
// knots i-1 .. i+2 that the value and the rounding of either end can need are fetched together; only when the estimate
// is off (the arc-length grid is not exactly uniform: 0.3 % of the lookups on the reference's tables) the interval is
// searched and the knots are fetched again.  (Before: grid ends, search, values, neighbours, knot = 5 dependent loads,
// 45 serialised round trips per slot in k_eval.)
__device__ __forceinline__ void lut_eval(const double* __restrict__ grid, const double* __restrict__ y, int n, double g0,
                                         double inv, double period, double s, double eps, double& val, double& slope, double& curv) {
  if (period > 0.0) s -= period * floor((s - g0) / period);  // closed track: same table lap after lap (exact no-op on the first)
  const double fi = (s - g0) * inv;
  int i = fi <= 0.0 ? 0 : (fi >= (double)(n - 2) ? n - 2 : (int)fi);
  int im = i > 0 ? i - 1 : 0, ip = i + 2 < n ? i + 2 : n - 1;
  double gm = grid[im], gi = grid[i], gi1 = grid[i + 1], gp = grid[ip];
  double ym = y[im], yi = y[i], yi1 = y[i + 1], yp = y[ip];
#if defined(__HIP_DEVICE_COMPILE__)
  // all eight values in registers HERE: otherwise the compiler sinks the loads into the (short-circuit) test below
  // and the branches after it, and the look-up is a chain of dependent round trips again
  asm volatile("" : "+v"(gm), "+v"(gi), "+v"(gi1), "+v"(gp), "+v"(ym), "+v"(yi), "+v"(yi1), "+v"(yp));
#endif
  if ((i > 0 && s < gi) || (i < n - 2 && s >= gi1)) {
    while (i > 0 && s < grid[i]) --i;
    while (i < n - 2 && s >= grid[i + 1]) ++i;
    im = i > 0 ? i - 1 : 0, ip = i + 2 < n ? i + 2 : n - 1;
    gm = grid[im], gi = grid[i], gi1 = grid[i + 1], gp = grid[ip];
    ym = y[im], yi = y[i], yi1 = y[i + 1], yp = y[ip];
  }
  double d = gi1 - gi;
  double sl = (yi1 - yi) / d;
  val = yi + sl * (s - gi);
  slope = sl;
  curv = 0.0;
  if (eps > 0.0) {
    int kn = -1;
    double z = 0.0, W = 0.0, sg = 0.0;
    if (i > 0) {
      double Wk = 0.5 * fmin(gi - gm, d), zz = s - gi;
      if (zz >= 0.0 && zz < Wk) kn = i, z = zz, W = Wk, sg = 1.0;
    }
    if (kn < 0 && i + 1 < n - 1) {
      double Wk = 0.5 * fmin(d, gp - gi1), zz = s - gi1;
      if (zz < 0.0 && -zz < Wk) kn = i + 1, z = zz, W = Wk, sg = -1.0;
    }
    if (kn > 0) {
      const bool left = kn == i;  // knots kn-1, kn, kn+1
      const double ga = left ? gm : gi, gb = left ? gi : gi1, gc = left ? gi1 : gp;
      const double ya = left ? ym : yi, yb = left ? yi : yi1, yc = left ? yi1 : yp;
      double sa = (yb - ya) / (gb - ga);
      double sb = (yc - yb) / (gc - gb);
      double J = sb - sa, R = sqrt(z * z + eps * eps), RW = sqrt(W * W + eps * eps);
      double a = (1.0 - W / RW) / (2.0 * W), b = W - RW - a * W * W;
      val += 0.5 * J * (R + a * z * z + b - sg * z);
      slope += 0.5 * J * (z / R + 2.0 * a * z - sg);
      curv = 0.5 * J * (eps * eps / (R * R * R) + 2.0 * a);
    }
  }
}
__device__ __forceinline__ double lut_val(const double* __restrict__ grid, const double* __restrict__ y, int n, double g0,
                                          double inv, double period, double s, double eps) {
  double v, sl, cv;
  lut_eval(grid, y, n, g0, inv, period, s, eps, v, sl, cv);
  return v;
}

// ---------------------------------------------------------------------------------------------- tyres
// second-order jet over (vx, vy, r, delta): value, gradient[4], packed symmetric Hessian[10]
struct Jet4 {
  double v, g[4], h[10];
};

// F_y = -K sin(C atan(B alpha)), alpha = atan2(vy + l r, vx) - delta_on * delta    (model.py:101-114)
__device__ __forceinline__ void pacejka_jet(double vx, double vy, double r, double delta, double l, double delta_on,
                                            double Bp, double Cp, double K, Jet4& F) {
  double a = vy + l * r;
  double q = a * a + vx * vx, iq = 1.0 / q;
  double ta = vx * iq, tb = -a * iq;                      // d atan2 / da, / dvx
  double taa = -2.0 * a * vx * iq * iq, tbb = -taa, tab = (a * a - vx * vx) * iq * iq;
  double alpha = atan2(a, vx) - delta_on * delta;
  double ag[4] = {tb, ta, l * ta, -delta_on};
  double ah[10];
  ah[sidx(0, 0)] = tbb;
  ah[sidx(1, 0)] = tab;
  ah[sidx(1, 1)] = taa;
  ah[sidx(2, 0)] = l * tab;
  ah[sidx(2, 1)] = l * taa;
  ah[sidx(2, 2)] = l * l * taa;
  ah[sidx(3, 0)] = ah[sidx(3, 1)] = ah[sidx(3, 2)] = ah[sidx(3, 3)] = 0.0;
  double z = Bp * alpha, d = 1.0 + z * z, t = atan(z);
  double sc, cc;
  sincos(Cp * t, &sc, &cc);
  double cb = Cp * Bp / d;
  double p0 = -K * sc;
  double p1 = -K * cc * cb;
  double p2 = K * sc * cb * cb + 2.0 * K * cc * Cp * Bp * Bp * z / (d * d);
  F.v = p0;
#pragma unroll
  for (int i = 0; i < 4; i++) F.g[i] = p1 * ag[i];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) F.h[sidx(i, j)] = p2 * ag[i] * ag[j] + p1 * ah[sidx(i, j)];
}

// product of a Jet4 with a function of delta only (value s0, first s1, second s2)
__device__ __forceinline__ void jet4_mul_delta(const Jet4& F, double s0, double s1, double s2, Jet4& P) {
  P.v = F.v * s0;
#pragma unroll
  for (int i = 0; i < 4; i++) P.g[i] = F.g[i] * s0;
  P.g[3] += F.v * s1;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) P.h[sidx(i, j)] = F.h[sidx(i, j)] * s0;
#pragma unroll
  for (int i = 0; i < 3; i++) P.h[sidx(3, i)] += F.g[i] * s1;
  P.h[sidx(3, 3)] += 2.0 * F.g[3] * s1 + F.v * s2;
}

// ---------------------------------------------------------------------------------------------- rhs
// Value-only right-hand side (plant, line search).  f[6], f[7] = u.
__device__ __forceinline__ void rhs_val(const ltompc_params& p, const Tables& T, double eps, const double* x,
                                        const double* u, double* f) {
  double kap = lut_val(T.s_kappa, T.kappa, T.n, T.g0_kappa, T.inv_kappa, T.period, x[0], eps);
  double n = x[1], mu = x[2], vx = x[3], vy = x[4], r = x[5], de = x[6], th = x[7];
  double sm, cm, sd, cd;
  sincos(mu, &sm, &cm);
  sincos(de, &sd, &cd);
  double sdot = (vx * cm - vy * sm) / (1.0 - n * kap);
  double af = atan2(vy + p.length_f * r, vx) - de;
  double ar = atan2(vy - p.length_r * r, vx);
  double L = p.length_f + p.length_r;
  double Fnf = p.length_r * p.mass * p.gravity / L, Fnr = p.length_f * p.mass * p.gravity / L;
  double Fyf = -Fnf * p.D_f * sin(p.C_f * atan(p.B_f * af));
  double Fyr = -Fnr * p.D_r * sin(p.C_r * atan(p.B_r * ar));
  double Fx = p.C_m * th - p.Cr_0 - p.Cr_2 * vx * vx;
  f[0] = sdot;
  f[1] = vx * sm + vy * cm;
  f[2] = r - kap * sdot;
  f[3] = (Fx - Fyf * sd + p.mass * vy * r) / p.mass;
  f[4] = (Fyr + Fyf * cd - p.mass * vx * r) / p.mass;
  f[5] = (Fyf * p.length_f * cd - Fyr * p.length_r + p.ptv * (sd / cd * vx / L - r)) / p.inertia_z;  // (+ Mtv, model.py:162-164; ptv = 0 in the reference)
  f[6] = u[0];
  f[7] = u[1];
}

// f[0..5], Jacobian J (6 x 8 row-major, rows 6,7 of df/dx are zero) and, if lam != nullptr,
// H += scale * sum_i lam[i] d2 f_i / dx2  accumulated into the packed symmetric 8x8 H[36].
__device__ __forceinline__ void rhs_derivs(const ltompc_params& p, const Tables& T, double eps, const double* x,
                                           double* f, double* J, const double* lam, double scale, double* H) {
  double s = x[0], n = x[1], mu = x[2], vx = x[3], vy = x[4], r = x[5], de = x[6], th = x[7];
#pragma unroll
  for (int i = 0; i < 48; i++) J[i] = 0.0;
  // ---- kinematic rows (s, n, mu) over (s, n, mu, vx, vy)
  double kap, kp, kpp;
  lut_eval(T.s_kappa, T.kappa, T.n, T.g0_kappa, T.inv_kappa, T.period, s, eps, kap, kp, kpp);
  double sm, cm;
  sincos(mu, &sm, &cm);
  double w = vx * cm - vy * sm, nd = vx * sm + vy * cm;
  double g = 1.0 / (1.0 - n * kap), g2 = g * g, g3 = g2 * g;
  double g_s = n * kp * g2, g_n = kap * g2;
  double g_ss = n * kpp * g2 + 2.0 * n * n * kp * kp * g3;
  double g_sn = kp * g2 + 2.0 * n * kap * kp * g3;
  double g_nn = 2.0 * kap * kap * g3;
  double sdot = w * g;
  // first derivatives of sdot over (s,n,mu,vx,vy)
  double S1[5] = {w * g_s, w * g_n, -nd * g, cm * g, -sm * g};
  f[0] = sdot;
  f[1] = nd;
  f[2] = r - kap * sdot;
#pragma unroll
  for (int j = 0; j < 5; j++) J[0 * 8 + j] = S1[j];
  J[1 * 8 + 2] = w, J[1 * 8 + 3] = sm, J[1 * 8 + 4] = cm;
  J[2 * 8 + 0] = -kp * sdot - kap * S1[0];
#pragma unroll
  for (int j = 1; j < 5; j++) J[2 * 8 + j] = -kap * S1[j];
  J[2 * 8 + 5] = 1.0;
  if (lam) {
    // second derivatives of sdot, packed over (s,n,mu,vx,vy) == global indices 0..4
    double S2[15];
    S2[sidx(0, 0)] = w * g_ss;
    S2[sidx(1, 0)] = w * g_sn;
    S2[sidx(1, 1)] = w * g_nn;
    S2[sidx(2, 0)] = -nd * g_s;
    S2[sidx(2, 1)] = -nd * g_n;
    S2[sidx(2, 2)] = -w * g;
    S2[sidx(3, 0)] = cm * g_s;
    S2[sidx(3, 1)] = cm * g_n;
    S2[sidx(3, 2)] = -sm * g;
    S2[sidx(3, 3)] = 0.0;
    S2[sidx(4, 0)] = -sm * g_s;
    S2[sidx(4, 1)] = -sm * g_n;
    S2[sidx(4, 2)] = -cm * g;
    S2[sidx(4, 3)] = 0.0;
    S2[sidx(4, 4)] = 0.0;
    double l0 = scale * lam[0], l1 = scale * lam[1], l2 = scale * lam[2];
    // row 0 (sdot) and the -kappa*sdot part of row 2
    double c0 = l0 - l2 * kap;
#pragma unroll
    for (int i = 0; i < 15; i++) H[i] += c0 * S2[i];  // packed (i,j<=4) coincide with the global packing
    // remaining terms of row 2: -kappa'' sdot - 2 kappa' S_s on (s,s); -kappa' S_y on (s,y)
    H[sidx(0, 0)] += l2 * (-kpp * sdot - 2.0 * kp * S1[0]);
#pragma unroll
    for (int j = 1; j < 5; j++) H[sidx(j, 0)] += l2 * (-kp * S1[j]);
    // row 1 (ndot)
    H[sidx(2, 2)] += l1 * (-nd);
    H[sidx(3, 2)] += l1 * cm;
    H[sidx(4, 2)] += l1 * (-sm);
  }
  // ---- dynamic rows (vx, vy, r) over (vx, vy, r, delta) [+ throttle, linear]
  double L = p.length_f + p.length_r;
  double Kf = p.length_r * p.mass * p.gravity / L * p.D_f, Kr = p.length_f * p.mass * p.gravity / L * p.D_r;
  Jet4 Ff, Fr, Ps, Pc;
  pacejka_jet(vx, vy, r, de, p.length_f, 1.0, p.B_f, p.C_f, Kf, Ff);
  pacejka_jet(vx, vy, r, de, -p.length_r, 0.0, p.B_r, p.C_r, Kr, Fr);
  double sd, cd;
  sincos(de, &sd, &cd);
  jet4_mul_delta(Ff, sd, cd, -sd, Ps);  // Fyf sin(delta)
  jet4_mul_delta(Ff, cd, -sd, -cd, Pc);  // Fyf cos(delta)
  double im = 1.0 / p.mass, iz = 1.0 / p.inertia_z;
  f[3] = (p.C_m * th - p.Cr_0 - p.Cr_2 * vx * vx - Ps.v) * im + vy * r;
  f[4] = (Fr.v + Pc.v) * im - vx * r;
  // torque vectoring Mtv = ptv (tan(delta) vx / L - r)  (model.py:162-164; the reference has ptv = 0: every term below vanishes)
  const double td = sd / cd, sec2 = 1.0 / (cd * cd), pz = p.ptv * iz, iL = 1.0 / L;
  f[5] = (p.length_f * Pc.v - p.length_r * Fr.v) * iz + pz * (td * vx * iL - r);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    J[3 * 8 + 3 + j] = -Ps.g[j] * im;
    J[4 * 8 + 3 + j] = (Fr.g[j] + Pc.g[j]) * im;
    J[5 * 8 + 3 + j] = (p.length_f * Pc.g[j] - p.length_r * Fr.g[j]) * iz;
  }
  J[5 * 8 + 3] += pz * td * iL, J[5 * 8 + 5] += -pz, J[5 * 8 + 6] += pz * vx * sec2 * iL;
  J[3 * 8 + 3] += -2.0 * p.Cr_2 * vx * im;
  J[3 * 8 + 4] += r;
  J[3 * 8 + 5] += vy;
  J[3 * 8 + 7] = p.C_m * im;
  J[4 * 8 + 3] += -r;
  J[4 * 8 + 5] += -vx;
  if (lam) {
    double l3 = scale * lam[3], l4 = scale * lam[4], l5 = scale * lam[5];
    double cs = -l3 * im;                                  // weight of Hess(Ps)
    double cc = l4 * im + l5 * p.length_f * iz;            // weight of Hess(Pc)
    double cr = l4 * im - l5 * p.length_r * iz;            // weight of Hess(Fr)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j <= i; j++)
        H[sidx(3 + i, 3 + j)] += cs * Ps.h[sidx(i, j)] + cc * Pc.h[sidx(i, j)] + cr * Fr.h[sidx(i, j)];
    H[sidx(3, 3)] += l3 * (-2.0 * p.Cr_2 * im);
    H[sidx(6, 3)] += l5 * pz * sec2 * iL;                    // d2 Mtv / dvx ddelta
    H[sidx(6, 6)] += l5 * pz * vx * 2.0 * sec2 * td * iL;    // d2 Mtv / ddelta2
    H[sidx(5, 4)] += l3;   // d2 (vy r)
    H[sidx(5, 3)] += -l4;  // d2 (-vx r)
  }
}

// ---------------------------------------------------------------------------------------------- cost
// Node cost (controller.py:51-53): value; if g != nullptr also gradient g[8] (+=) and packed Hessian H[36] (+=).
__device__ __forceinline__ double cost_eval(const ltompc_params& p, const Tables& T, double eps, const double* x,
                                            bool terminal, double* g, double* H) {
  double n = x[1], mu = x[2], vx = x[3], vy = x[4], de = x[6];
  double val = p.q_n * n * n + p.q_mu * mu * mu + p.q_vy * vy * vy;
  if (g) {
    g[1] += 2.0 * p.q_n * n, g[2] += 2.0 * p.q_mu * mu, g[4] += 2.0 * p.q_vy * vy;
    H[sidx(1, 1)] += 2.0 * p.q_n, H[sidx(2, 2)] += 2.0 * p.q_mu, H[sidx(4, 4)] += 2.0 * p.q_vy;
  }
  if (terminal) return val;
  double vr, vr1, vr2;
  lut_eval(T.s_arc, T.v_ref, T.n, T.g0_arc, T.inv_arc, T.period, x[0], eps, vr, vr1, vr2);
  double cv = p.vref_scale;
  double e = vx - cv * vr;
  double rho = p.length_r / (p.length_f + p.length_r);
  double q = vx * vx + vy * vy, iq = 1.0 / q;
  double d = 1.0 + rho * rho * de * de;
  double b = atan(vy / vx) - atan(rho * de);
  val += p.q_v * e * e + p.q_B * b * b;
  if (g) {
    g[0] += -2.0 * p.q_v * e * cv * vr1;
    g[3] += 2.0 * p.q_v * e;
    H[sidx(0, 0)] += 2.0 * p.q_v * (cv * cv * vr1 * vr1 - e * cv * vr2);
    H[sidx(3, 0)] += -2.0 * p.q_v * cv * vr1;
    H[sidx(3, 3)] += 2.0 * p.q_v;
    // b over (vx, vy, delta): d atan(vy/vx) = (-vy, vx)/q
    double bx = -vy * iq, by = vx * iq, bd = -rho / d;
    double bxx = 2.0 * vx * vy * iq * iq, byy = -bxx, bxy = (vy * vy - vx * vx) * iq * iq;
    double bdd = 2.0 * rho * rho * rho * de / (d * d);
    double k2 = 2.0 * p.q_B;
    g[3] += k2 * b * bx, g[4] += k2 * b * by, g[6] += k2 * b * bd;
    H[sidx(3, 3)] += k2 * (bx * bx + b * bxx);
    H[sidx(4, 3)] += k2 * (bx * by + b * bxy);
    H[sidx(4, 4)] += k2 * (by * by + b * byy);
    H[sidx(6, 3)] += k2 * (bx * bd);
    H[sidx(6, 4)] += k2 * (by * bd);
    H[sidx(6, 6)] += k2 * (bd * bd + b * bdd);
  }
  return val;
}

// ---------------------------------------------------------------------------------------------- constraints
// Track-boundary constraints g <= 0 (model.py:70-84).  The right-hand one, -n + (L/2) sin|mu| + (W/2) cos mu - N_R,
// has a convex kink at mu = 0; on |mu| <= pi/2 it is exactly the pair gR+ / gR- below (sin|mu| = max(+-sin mu)),
// which is what is solved (same feasible set and minimisers).  The left one keeps sin(sign(mu) mu) with CasADi's
// derivative convention sign' = 0.  Outputs: val[3]; if gs != nullptr the non-zero first derivatives
// gs[q] = d/ds, gn[q] = d/dn, gm[q] = d/dmu and second derivatives hss[q], hmm[q].
__device__ __forceinline__ void cons_eval(const ltompc_params& p, const Tables& T, double eps, const double* x,
                                          double* val, double* gs, double* gn, double* gm, double* hss, double* hmm) {
  double NL, NL1, NL2, NR, NR1, NR2;
  lut_eval(T.s_arc, T.n_left, T.n, T.g0_arc, T.inv_arc, T.period, x[0], eps, NL, NL1, NL2);
  lut_eval(T.s_arc, T.n_right, T.n, T.g0_arc, T.inv_arc, T.period, x[0], eps, NR, NR1, NR2);
  double hl = 0.5 * (p.length_f + p.length_r), hw = 0.5 * p.width;
  double mu = x[2], sm, cm;
  sincos(mu, &sm, &cm);
  double sg = (mu > 0.0) - (mu < 0.0);
  double sabs = sg * sm;  // sin|mu|
  val[0] = x[1] - hl * sabs + hw * cm - NL;
  val[1] = -x[1] + hl * sm + hw * cm - NR;
  val[2] = -x[1] - hl * sm + hw * cm - NR;
  if (gs) {
    gs[0] = -NL1, gs[1] = -NR1, gs[2] = -NR1;
    gn[0] = 1.0, gn[1] = -1.0, gn[2] = -1.0;
    gm[0] = -hl * sg * cm - hw * sm;
    gm[1] = hl * cm - hw * sm;
    gm[2] = -hl * cm - hw * sm;
    hss[0] = -NL2, hss[1] = -NR2, hss[2] = -NR2;
    hmm[0] = hl * sg * sg * sabs - hw * cm;
    hmm[1] = -hl * sm - hw * cm;
    hmm[2] = hl * sm - hw * cm;
  }
}

// ---------------------------------------------------------------------------------------------- friction ellipse
// model.py:86-99 get_traction_ellipse_constraint (registered as soft nl constraints in lines the reference has commented out,
// controller.py:72-74): long = rho 0.5 C_m T, ellipse_a = long^2 + F_y,a^2 - (alpha D_a)^2 <= 0 for a = front, rear; here
// normalised by the radius, g_a = (long^2 + F_y,a^2) / ell_D_a^2 - 1 (include/ltompc.h).  Derivatives over (vx, vy, r, delta, T)
// = states 3..7: gradient g[q][5], packed Hessian h[q][15].
__device__ __forceinline__ void ellipse_eval(const ltompc_params& p, const double* x, double* val, double (*g)[5], double (*h)[15]) {
  const double vx = x[3], vy = x[4], r = x[5], de = x[6], th = x[7];
  const double L = p.length_f + p.length_r;
  const double Kf = p.length_r * p.mass * p.gravity / L * p.D_f, Kr = p.length_f * p.mass * p.gravity / L * p.D_r;
  Jet4 F[2];
  pacejka_jet(vx, vy, r, de, p.length_f, 1.0, p.B_f, p.C_f, Kf, F[0]);
  pacejka_jet(vx, vy, r, de, -p.length_r, 0.0, p.B_r, p.C_r, Kr, F[1]);
  const double c = p.ell_rho * 0.5 * p.C_m, lng = c * th;
  const double iD[2] = {1.0 / (p.ell_D_f * p.ell_D_f), 1.0 / (p.ell_D_r * p.ell_D_r)};
#pragma unroll
  for (int q = 0; q < 2; q++) {
    val[q] = (lng * lng + F[q].v * F[q].v) * iD[q] - 1.0;
    if (g) {
#pragma unroll
      for (int a = 0; a < 4; a++) g[q][a] = 2.0 * F[q].v * F[q].g[a] * iD[q];
      g[q][4] = 2.0 * c * c * th * iD[q];
#pragma unroll
      for (int a = 0; a < 5; a++)
#pragma unroll
        for (int b = 0; b <= a; b++)
          h[q][sidx(a, b)] = a < 4 ? 2.0 * (F[q].g[a] * F[q].g[b] + F[q].v * F[q].h[sidx(a, b)]) * iD[q] : (b == 4 ? 2.0 * c * c * iD[q] : 0.0);
    }
  }
}
// value only (line search, initialisation): same numbers as the oracle's ell_val
__device__ __forceinline__ void ellipse_val(const ltompc_params& p, const double* x, double* val) {
  const double vx = x[3], vy = x[4], r = x[5], de = x[6], th = x[7];
  const double af = atan2(vy + p.length_f * r, vx) - de, ar = atan2(vy - p.length_r * r, vx);
  const double L = p.length_f + p.length_r;
  const double Fnf = p.length_r * p.mass * p.gravity / L, Fnr = p.length_f * p.mass * p.gravity / L;
  const double Fyf = -Fnf * p.D_f * sin(p.C_f * atan(p.B_f * af)), Fyr = -Fnr * p.D_r * sin(p.C_r * atan(p.B_r * ar));
  const double lng = p.ell_rho * 0.5 * p.C_m * th;
  val[0] = (lng * lng + Fyf * Fyf) / (p.ell_D_f * p.ell_D_f) - 1.0;
  val[1] = (lng * lng + Fyr * Fyr) / (p.ell_D_r * p.ell_D_r) - 1.0;
}

__device__ __forceinline__ double bound_h(double sgn, double val, double x) { return sgn < 0.0 ? val - x : x - val; }

}  // namespace ltompc
