// harness.cpp — TEST INFRASTRUCTURE: the solver's device code (csrc/*.h, the thread-per-slot kernels and the serial Riccati
// kernel = the LTOMPC_RICCATI=serial path of the library) compiled as host C++ and run under AddressSanitizer +
// UndefinedBehaviorSanitizer with every work buffer filled with NaN bit patterns (zero-filled device
// buffers would hide reads of words no kernel has written; the pool's GPUs run no sanitizer).
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -DLTOMPC_HOST_HARNESS -I<csrc> -I<harness> ...
// Reads a problem from a text file (tables, x0 batch, horizon, options), runs the interior-point iterations with the same
// launch sequence as ltompc_make_step_dev (identity instance list, no re-packing), prints status / iterations / u0 per
// instance.  tests/test_host_harness.py compares that with the oracle.  NOT a product path and not an oracle: nothing in
// the package or in bench.py uses it.
#include "layout.h"
#include "linearise.h"
#include "riccati.h"
#include "linesearch.h"
#include "aux_kernels.h"
#include "velocity.h"

#include <string>
#include <thread>
#include <vector>

using namespace ltompc;

static double* poisoned(size_t n) {
  double* p = (double*)malloc(n * sizeof(double));
  memset(p, 0xFF, n * sizeof(double));  // NaN as doubles (LTOMPC_POISON=1 on the device)
  return p;
}
static int* ipoisoned(size_t n, int fill) {
  int* p = (int*)malloc(n * sizeof(int));
  for (size_t i = 0; i < n; i++) p[i] = fill;
  return p;
}
static void default_params(ltompc_params* p) {  // = ltompc_default_params (ltompc.hip)
  memset(p, 0, sizeof *p);
  p->mass = 1000.0, p->inertia_z = 1000.0, p->length_f = 1.5, p->length_r = 1.5, p->width = 2.3;
  p->B_f = 10.0, p->C_f = 1.3, p->D_f = 1.0, p->B_r = 12.0, p->C_r = 1.2, p->D_r = 1.0;
  p->C_m = 1000.0, p->Cr_0 = 0.01, p->Cr_2 = 0.0003, p->gravity = 9.81;
  p->q_n = 0.5, p->q_mu = 3.0, p->q_vy = 1.0, p->q_v = 1.0, p->vref_scale = 0.6, p->q_B = 1e-2;
  p->r_du[0] = p->r_du[1] = 1e-2;
  for (int i = 0; i < NX; i++) p->x_lb[i] = -LTOMPC_NO_BOUND, p->x_ub[i] = LTOMPC_NO_BOUND;
  const double pi = 3.14159265358979323846;
  p->x_lb[0] = 0.0, p->x_lb[2] = -pi * 0.5, p->x_ub[2] = pi * 0.5, p->x_lb[3] = 0.0;
  p->x_lb[6] = -pi / 4, p->x_ub[6] = pi / 4, p->x_lb[7] = -1.0, p->x_ub[7] = 1.0;
  p->u_lb[0] = -2 * pi / 4, p->u_ub[0] = 2 * pi / 4, p->u_lb[1] = -1.0, p->u_ub[1] = 1.0;
}
static void default_options(ltompc_options* o) {  // = ltompc_default_options
  memset(o, 0, sizeof *o);
  o->t_step = 0.1, o->tol = 1e-8, o->acceptable_tol = 1e-6, o->mu_init = 0.1, o->mu_min = 1e-9;
  o->kappa_eps = 10, o->kappa_mu = 0.2, o->theta_mu = 1.5, o->tau_min = 0.99, o->bound_push = 1e-2;
  o->s_max = 100, o->delta_w_first = 1e-4, o->smooth_eps_min = 1e-4, o->smooth_scale = 1.0;
  o->max_iter = 1000, o->acceptable_iter = 15, o->n_linesearch = 8, o->stall_iter = 15, o->max_ls_fail = 8;
  o->warm_reset_on_fail = 1, o->resto_rho = 1000.0;
  o->resto_rho_max = 1e6, o->resto_rho_factor = 1e3, o->dual_inf_max = 1e4, o->max_mu_stay = 100, o->infeasible_sticky = 1, o->node0_check = 1, o->warm_fallback_iter = 25, o->resto_shift_retry = 1;
}
// further options as key=value arguments (tests of the option paths)
static bool set_option(ltompc_options* o, const char* arg) {
  const char* eq = strchr(arg, '=');
  if (!eq) return false;
  const std::string key(arg, eq - arg);
  const double v = atof(eq + 1);
#define OPT_D(name) if (key == #name) { o->name = v; return true; }
#define OPT_I(name) if (key == #name) { o->name = (int)v; return true; }
  OPT_D(dual_inf_max) OPT_D(resto_rho) OPT_D(resto_rho_max) OPT_D(resto_rho_factor) OPT_D(mu_init_warm) OPT_D(tol) OPT_D(mu_init)
  OPT_I(infeasible_sticky) OPT_I(max_mu_stay) OPT_I(node0_check) OPT_I(warm_fallback_iter) OPT_I(resto_shift_retry) OPT_I(warm_shift) OPT_I(resto_sticky) OPT_I(max_iter)
  OPT_I(warm_reset_on_fail) OPT_I(stall_iter)
#undef OPT_D
#undef OPT_I
  return false;
}
static void build_bounds(const ltompc_params& p, Bounds& b) {  // = build_bounds (ltompc.hip)
  memset(&b, 0, sizeof b);
  for (int i = 0; i < NX; i++) {
    if (p.x_lb[i] > -LTOMPC_NO_BOUND) b.xb_idx[b.n_xb] = i, b.xb_sgn[b.n_xb] = -1.0, b.xb_val[b.n_xb] = p.x_lb[i], b.n_xb++;
    if (p.x_ub[i] < LTOMPC_NO_BOUND) b.xb_idx[b.n_xb] = i, b.xb_sgn[b.n_xb] = +1.0, b.xb_val[b.n_xb] = p.x_ub[i], b.n_xb++;
  }
  for (int i = 0; i < NU; i++) {
    if (p.u_lb[i] > -LTOMPC_NO_BOUND) b.ub_idx[b.n_ub] = i, b.ub_sgn[b.n_ub] = -1.0, b.ub_val[b.n_ub] = p.u_lb[i], b.n_ub++;
    if (p.u_ub[i] < LTOMPC_NO_BOUND) b.ub_idx[b.n_ub] = i, b.ub_sgn[b.n_ub] = +1.0, b.ub_val[b.n_ub] = p.u_ub[i], b.n_ub++;
  }
  b.nel = p.ell_penalty > 0.0 ? NEL : 0;
  b.ni = b.n_ub + 2 * b.n_xb + NNL + b.nel;
}

// run a thread-per-(k, b) kernel body over a grid of `threads` threads, 64 per block
template <typename F>
static void grid64(int threads, F&& body) {
  blockDim.x = 64, gridDim.x = (threads + 63) / 64;
  for (unsigned blk = 0; blk < gridDim.x; blk++)
    for (unsigned t = 0; t < 64; t++) blockIdx.x = blk, threadIdx.x = t, body();
}

int main(int argc, char** argv) {
  if (argc < 2) return fprintf(stderr, "usage: harness problem.txt\n"), 2;
  FILE* f = fopen(argv[1], "r");
  if (!f) return 2;
  int nt, N, B, any_bounds, ticks;
  double soft_rho;
  double ell[4];
  if (fscanf(f, "%d %d %d %d %lf %d %lf %lf %lf %lf", &nt, &N, &B, &any_bounds, &soft_rho, &ticks, &ell[0], &ell[1], &ell[2], &ell[3]) != 10) return 2;
  std::vector<double> tab((size_t)6 * nt), x0((size_t)8 * B);
  for (auto& v : tab) if (fscanf(f, "%lf", &v) != 1) return 2;
  for (auto& v : x0) if (fscanf(f, "%lf", &v) != 1) return 2;
  fclose(f);
  Consts K;
  memset(&K, 0, sizeof K);
  default_params(&K.p), default_options(&K.o);
  K.o.soft_rho = soft_rho;
  for (int a = 2; a < argc; a++)
    if (!set_option(&K.o, argv[a])) return fprintf(stderr, "harness: unknown option %s\n", argv[a]), 2;
  K.p.ell_penalty = ell[0], K.p.ell_rho = ell[1], K.p.ell_D_f = ell[2], K.p.ell_D_r = ell[3];
  build_bounds(K.p, K.bd);
  const int ni = K.bd.ni, Bp = (B + 63) / 64 * 64;
  double* d_tab = (double*)malloc(sizeof(double) * 6 * nt);  // exact size: ASan sees any look-up outside the tables
  memcpy(d_tab, tab.data(), sizeof(double) * 6 * nt);
  Tables& T = K.T;
  T.n = nt, T.s_kappa = d_tab, T.kappa = d_tab + nt, T.s_arc = d_tab + 2 * nt, T.n_left = d_tab + 3 * nt, T.n_right = d_tab + 4 * nt, T.v_ref = d_tab + 5 * nt;
  T.g0_kappa = tab[0], T.inv_kappa = (double)(nt - 1) / (tab[nt - 1] - tab[0]);
  T.g0_arc = tab[2 * (size_t)nt], T.inv_arc = (double)(nt - 1) / (tab[3 * (size_t)nt - 1] - tab[2 * (size_t)nt]);
  T.period = 0.0;
  Work W;
  memset(&W, 0, sizeof W);
  W.N = N, W.B = B, W.Bp = Bp;
  const size_t n = N, bp = Bp;
  // EXACT sizes (no "+64" slack as in ltompc_create: an over-read past a buffer is an ASan error here)
  W.X = poisoned(8 * (n + 1) * bp), W.C = poisoned(8 * n * bp), W.U = poisoned(2 * n * bp), W.L1 = poisoned(8 * n * bp), W.L2 = poisoned(8 * n * bp);
  const size_t nel = K.bd.nel;
  W.T = poisoned((ni + NNL + nel) * n * bp), W.NU = poisoned(ni * n * bp);
  W.dX = poisoned(8 * (n + 1) * bp), W.dC = poisoned(8 * n * bp), W.dU = poisoned(2 * n * bp), W.nL1 = poisoned(8 * n * bp), W.nL2 = poisoned(8 * n * bp);
  W.dT = poisoned((ni + NNL + nel) * n * bp), W.dNU = poisoned(ni * n * bp);
  W.QP = poisoned((size_t)QP_NF * (n + 1) * bp), W.RC = poisoned((size_t)RC_NF * (n + 1) * bp);
  W.RS = poisoned((size_t)RS_NF * n * bp), W.SP = poisoned((size_t)SP_NF * n * bp), W.LS = poisoned((size_t)3 * (K.o.n_linesearch + 1) * n * bp);
  W.x0 = poisoned(8 * bp), W.uprev = poisoned(2 * bp), W.st = poisoned((size_t)ST_NF * bp), W.filt = poisoned((size_t)2 * FILTER_MAX * bp);
  W.si = ipoisoned((size_t)SI_NF * bp, 0), W.active = ipoisoned(K.o.max_iter + 2, 0), W.ls_list = ipoisoned(bp, -1), W.ls_count = ipoisoned(4, 0);
  W.DBG = nullptr;
  W.BK = poisoned(18 * n * bp);
  {
    int* orig = ipoisoned(bp, -1);
    for (int b = 0; b < B; b++) orig[b] = b;
    W.orig = orig;
  }
  std::vector<int> act(Bp), nact(1, B);
  for (int b = 0; b < Bp; b++) act[b] = b;
  Launch la{act.data(), nact.data(), Bp, 0};
  std::vector<double> x(x0), u0((size_t)2 * B, 0.0), xn((size_t)8 * B);
  const bool ref = !any_bounds, el = K.bd.nel > 0;
  for (int tick = 0; tick < ticks; tick++) {
    const int cold = tick == 0;
    grid64(B, [&] { k_load_x0(W, x.data(), nullptr, K.o.resto_sticky, cold ? 0 : 1); });
    if (cold) grid64(B, [&] { k_zero_uprev(W); });
    if (!cold && K.o.warm_shift) {
      grid64((N + 1) * Bp, [&] { k_shift(W, 0); });
      grid64((N + 1) * Bp, [&] { k_shift(W, 1); });
    }
    grid64(N * Bp, [&] { k_init(&K, &W, cold); });
    memset(W.active, 0, sizeof(int) * (K.o.max_iter + 2)), W.ls_count[0] = W.ls_count[1] = 0;
    for (int it = 0;; it++) {
      grid64(N * Bp, [&] { el ? (ref ? k_eval<BoundsRef, true>(&K, &W, la) : k_eval<BoundsAny, true>(&K, &W, la)) : (ref ? k_eval<BoundsRef, false>(&K, &W, la) : k_eval<BoundsAny, false>(&K, &W, la)); });
      grid64(Bp, [&] { k_riccati(&K, &W, la, it); });
      if (it >= K.o.max_iter) break;
      grid64(N * Bp, [&] { el ? (ref ? k_expand<BoundsRef, true>(&K, &W, la) : k_expand<BoundsAny, true>(&K, &W, la)) : (ref ? k_expand<BoundsRef, false>(&K, &W, la) : k_expand<BoundsAny, false>(&K, &W, la)); });
      grid64(N * Bp, [&] { el ? (ref ? k_linesearch<BoundsRef, true>(&K, &W, la, 0, Bp) : k_linesearch<BoundsAny, true>(&K, &W, la, 0, Bp)) : (ref ? k_linesearch<BoundsRef, false>(&K, &W, la, 0, Bp) : k_linesearch<BoundsAny, false>(&K, &W, la, 0, Bp)); });
      auto pick = [&](int b, int phase) {  // the 8 lanes of instance b (k_pick), as 8 OS threads
        LtLaneGroup g;
        pthread_barrier_init(&g.bar, nullptr, 8);
        std::vector<std::thread> th;
        for (int i = 0; i < 8; i++) th.emplace_back([&, i] { lt_group = &g, lt_lane = i; d_pick(K, W, b, i, phase, true); });
        for (auto& t : th) t.join();
        pthread_barrier_destroy(&g.bar);
      };
      for (int b = 0; b < B; b++) pick(b, 0);
      if (K.o.n_linesearch > 1 && W.ls_count[0] > 0) {
        const int jw = W.ls_count[0];
        grid64((K.o.n_linesearch - 1) * N * jw, [&] { el ? (ref ? k_linesearch<BoundsRef, true>(&K, &W, la, 1, jw) : k_linesearch<BoundsAny, true>(&K, &W, la, 1, jw)) : (ref ? k_linesearch<BoundsRef, false>(&K, &W, la, 1, jw) : k_linesearch<BoundsAny, false>(&K, &W, la, 1, jw)); });
        for (int j = 0; j < jw; j++) pick(W.ls_list[j], 1);
      }
      grid64(N * Bp, [&] { k_update(&K, &W, la); });
      int left = 0;
      for (int b = 0; b < B; b++) left += !W.si[(size_t)SI_DONE * Bp + b];
      if (!left) break;
    }
    grid64(B, [&] { k_store_u0(W, u0.data(), nullptr); });
    printf("tick %d\n", tick);
    for (int b = 0; b < B; b++)
      printf("%d %d %d %.17g %.17g %.17g %d %d %d %.17g %.17g\n", b, W.si[(size_t)SI_STATUS * Bp + b], W.si[(size_t)SI_ITERS * Bp + b], u0[2 * b], u0[2 * b + 1],
             W.st[(size_t)ST_E0 * Bp + b], W.si[(size_t)SI_NRESTO * Bp + b], W.si[(size_t)SI_NSHIFT * Bp + b], W.si[(size_t)SI_NFALLBACK * Bp + b],
             W.st[(size_t)ST_VIOL * Bp + b], W.st[(size_t)ST_G0 * Bp + b]);
    // plant step (k_plant) to the next tick's states
    Consts Kc = K;
    grid64(B, [&] { k_plant(Kc, B, x.data(), u0.data(), K.o.t_step, 100, xn.data()); });
    x = xn;
  }
  return 0;
}
