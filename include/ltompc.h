/*
 * ltompc.h — C ABI of the MI355X-native receding-horizon NLP solver (libltompc.so).
 *
 * Drop-in boundary for the one hot path of bruno-maruszczak/lap-time-optimization:
 *     u0 = controller.mpc.make_step(x0)                      reference src/mpc.py:142
 * i.e. do_mpc's MPC object as configured by src/mpc/controller.py:9-103 over the model of
 * src/mpc/model.py:130-185 and the look-up tables of src/path.py:96-101, src/mpc/track.py:30-42.
 * The reference has no FFI of its own (it is pure Python over do_mpc/CasADi/IPOPT); the entry points
 * below are what a ctypes binding for that path binds (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - all floating point is IEEE double; all arrays are plain C arrays, row-major, no torch types;
 *   - state  x = [s, n, mu, vx, vy, r, steering_angle, throttle]          (model.py:138-147)
 *     input  u = [steering_angle_change, throttle_change]                  (model.py:149-150)
 *   - a handle owns B independent MPC instances ("batch"); instance b is row b of every array;
 *   - every function returns 0 on success, <0 on a usage / HIP error (ltompc_last_error() has the text).
 *     Solver non-convergence is NOT an error (the reference returns IPOPT's last iterate silently,
 *     SURVEY.md §8b): it is reported per instance in `status`.
 *   - a handle is not thread-safe; distinct handles are independent.  Calls are synchronous unless the
 *     name ends in _async (those only enqueue on the handle's stream).
 */
#ifndef LTOMPC_H
#define LTOMPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define LTOMPC_NX 8
#define LTOMPC_NU 2
#define LTOMPC_NO_BOUND 1.0e30 /* |bound| >= this means "not set" (controller.py:78 `not_set`) */
#define LTOMPC_TABLE_ROWS 6    /* s_kappa, kappa, s_arc, n_left, n_right, v_ref */

/* per-instance solver status written by make_step */
#define LTOMPC_STATUS_SOLVED 0          /* scaled KKT error <= tol                                */
#define LTOMPC_STATUS_ACCEPTABLE 1      /* <= acceptable_tol for acceptable_iter iterations       */
#define LTOMPC_STATUS_MAX_ITER 2        /* iteration budget exhausted, last iterate returned      */
#define LTOMPC_STATUS_NUMERICAL 3       /* non-finite value / regularisation overflow             */
#define LTOMPC_STATUS_STALLED 4         /* no progress and the restoration phase could not help (or is
                                           switched off): IPOPT's 'restoration failed'              */
#define LTOMPC_STATUS_INFEASIBLE 5      /* the restoration phase converged, at its largest penalty
                                           (options.resto_rho_max), to a point where a track constraint stays
                                           violated by more than tol: a stationary point of the constraint
                                           violation up to 1 / resto_rho_max (IPOPT: 'converged to a point of
                                           local infeasibility'); that least-violation iterate is returned.
                                           Also: the measured state x0 itself violates a track constraint
                                           (options.node0_check: the reference's NLP has no feasible point) */

/* Vehicle + objective + bounds.  Defaults (ltompc_default_params) are the values the reference
 * actually uses, including its quirks: D_f = D_r = 1.0 because model.py:42-64
 * never reads them; car length in the boundary constraint is length_f + length_r (model.py:71). */
typedef struct ltompc_params {
  /* model.py:42-64, constructor defaults :16-29 */
  double mass, inertia_z, length_f, length_r, width;
  double B_f, C_f, D_f, B_r, C_r, D_r;
  double C_m, Cr_0, Cr_2, gravity;
  /* Torque vectoring (model.py:162-164, disabled in the reference: `Mtv = 0.0`, the line `Mtv = ptv * (rt - r)` with
   * rt = tan(delta) vx / (l_f + l_r) is commented out): yaw moment added to the r equation.  0 = the reference. */
  double ptv;
  /* controller.py:29,52-53: lterm = q_n n^2 + q_mu mu^2 + q_vy vy^2 + q_v (vx - vref_scale*v_ref(s))^2
   *                                 + q_B (atan(vy/vx) - atan(delta*l_r/(l_f+l_r)))^2 ; mterm = first three.
   * controller.py:40-41 + mpc.py:104: rterm = sum_i r_du[i] * (u_k[i] - u_{k-1}[i])^2                  */
  double q_n, q_mu, q_vy, q_v, vref_scale, q_B;
  double r_du[LTOMPC_NU];
  /* controller.py:79-103 (LTOMPC_NO_BOUND where the reference sets nothing) */
  double x_lb[LTOMPC_NX], x_ub[LTOMPC_NX];
  double u_lb[LTOMPC_NU], u_ub[LTOMPC_NU];
  /* Friction-ellipse constraints of the two axles (model.py:86-99 get_traction_ellipse_constraint; their registration as
   * soft nl constraints is commented out in the reference, controller.py:72-74: ell_penalty = 0 is the reference).
   *   long = ell_rho 0.5 C_m T ;  g_a = (long^2 + F_y,a^2) / ell_D_a^2 - 1 <= 0 at the nodes 1 .. N-1, a = front, rear,
   * always with an elastic variable that costs ell_penalty (do_mpc: soft_constraint=True, penalty_term_cons).  With
   * ell_D_f = ell_D_r = 1.0 (= alpha D with the reference's alpha = 1 and D = 1.0) and ell_rho = 1 this is the reference's
   * expression, which no tyre force in newtons can satisfy; a physical radius is the peak lateral force F_N D. */
  double ell_penalty, ell_rho, ell_D_f, ell_D_r;
} ltompc_params;

/* NLP transcription + interior-point options.  Defaults follow do_mpc 4.6.5 / IPOPT 3.14 defaults
 * as listed in SURVEY.md App. B (t_step, Radau-IIA deg 2, tol 1e-8, mu_init 0.1, monotone mu, ...). */
typedef struct ltompc_options {
  double t_step;          /* controller.py:9            0.1  */
  double tol;             /* ipopt tol                  1e-8 */
  double acceptable_tol;  /* ipopt acceptable_tol       1e-6 */
  double mu_init;         /* ipopt mu_init              0.1  */
  double mu_min;          /* tol/10                     1e-9 */
  double kappa_eps;       /* barrier_tol_factor         10   */
  double kappa_mu;        /* mu_linear_decrease_factor  0.2  */
  double theta_mu;        /* mu_superlinear_decrease_power 1.5 */
  double tau_min;         /* fraction-to-boundary       0.99 */
  double bound_push;      /* slack initialisation       1e-2 */
  double s_max;           /* dual-residual scaling cap  100  */
  double delta_w_first;   /* first Hessian regularisation 1e-4 */
  /* Smoothing of the reference NLP's kinks (|mu| in model.py:77-78, table knots): eps = max(smooth_eps_min,
   * smooth_scale * mu_barrier) in rad resp. metres; both 0 = the exact non-smooth functions. */
  double smooth_eps_min;  /* 1e-4 */
  double smooth_scale;    /* 1.0  */
  /* Barrier parameter at the start of a WARM-started solve (every make_step after the first).  0 = use mu_init, which
   * is what do_mpc/IPOPT do (IPOPT restarts at mu_init = 0.1 on every call); a smaller value (1e-3 .. 1e-2) keeps
   * the iterates close to the previous solution and saves iterations without changing the KKT point found. */
  double mu_init_warm;    /* 0 */
  /* Softened track constraints (do_mpc: set_nl_cons(..., soft_constraint=True, penalty_term_cons=soft_rho)).
   * 0 = hard constraints gL, gR <= 0 as in the reference (controller.py:69-70).  > 0: every track constraint of every
   * node gets an elastic variable e >= 0 (g - e <= 0) that costs soft_rho * e (exact L1 penalty: the solution is the
   * hard-constrained one wherever that exists and soft_rho exceeds its multipliers, a least-violation one otherwise).
   * It also is what keeps the closed loop going: from some states the lap reaches, the hard-constrained solve stalls
   * (no restoration phase here) or has no feasible point at all, and every later tick inherits the failure; with
   * soft_rho = 100 all 769 ticks of the buckmore lap converge.  The elastic variables are eliminated
   * with the slacks: same stage-QP sizes, same kernels, three more planes per interval. */
  double soft_rho;        /* 0 */
  /* Restoration phase (what IPOPT enters when its filter line search fails).  Here it is an ELASTIC mode:
   * the solve is continued on the same NLP with every track constraint relaxed by an elastic variable e >= 0 that
   * costs resto_rho * e (the machinery of soft_rho, with IPOPT's restoration penalty 1000), the barrier parameter and
   * the slacks re-centred.  When the elastic problem has converged with all e <= tol the iterate is a KKT point of the
   * hard-constrained NLP: the solve switches back to the hard constraints and terminates there (status SOLVED);
   * with some e > tol the problem is locally infeasible (status INFEASIBLE).  0 = no restoration: a failed line
   * search ends the solve with status STALLED after max_ls_fail attempts, as in round 1. */
  double resto_rho;       /* 1000 */
  /* Penalty escalation of the restoration phase.  An elastic problem that has converged with an elastic variable > tol has
   * only shown that, at its penalty, violating is cheaper than complying (the multiplier of that constraint sits at the
   * penalty); a feasible NLP whose multipliers exceed resto_rho ends there too.  So the penalty of THAT instance is
   * multiplied by resto_rho_factor (elastic variables, slacks, barrier re-centred at the current primal point, as at the
   * entry of the phase) and the elastic problem is solved again, until all elastic variables are <= tol (back to the hard
   * constraints, status SOLVED) or the penalty has reached resto_rho_max: only then the status is INFEASIBLE, a stationary
   * point of  objective / resto_rho_max + violation,  i.e. of the constraint violation itself up to 1 / resto_rho_max (IPOPT's
   * restoration phase minimises the violation alone; at 1e7 the escalated problems sit at the rounding floor and take 2 - 4
   * times the iterations, at 1e6 the least violation is reproduced to 2 % down to violations of 1e-5 m).  resto_rho_factor <= 1 or resto_rho_max <= resto_rho: no escalation,
   * INFEASIBLE means "at penalty resto_rho" (round 2's behaviour). */
  double resto_rho_max;    /* 1e6 */
  double resto_rho_factor; /* 1e3 (one step to resto_rho_max) */
  /* Early entry into the recovery steps (shifted restart, restoration phase): on the hard constraints, an iterate whose dual
   * infeasibility (max-norm of the Lagrangian's gradient, unscaled) exceeds this.  On this NLP a solve that converges stays
   * below 1e3 .. 1e4 after a warm start; the solves that exceed it are the ones whose multipliers diverge (1e5 -> 1e18) while
   * the filter keeps accepting steps of a percent - 100 to 250 iterations until the line search finally fails, the slowest
   * instances of every tick.  IPOPT has no such test (it would iterate on); it only changes WHEN the recovery starts, not
   * what the solve returns when it converges.  Warm-started solves only (from do_mpc's cold start, every node = x0, a
   * converging solve passes through dual infeasibilities of 1e7).  0 = off. */
  double dual_inf_max;     /* 1e4 */
  int max_iter;           /* controller.py:18 says 1000.  Per instance: iterations plus repeated Riccati sweeps (inertia correction) */
  int acceptable_iter;    /* ipopt acceptable_iter      15   */
  int n_linesearch;       /* step-size candidates alpha_max * 2^-l, l = 0..n_linesearch-1 */
  int stall_iter;         /* stop (status STALLED) after this many consecutive steps with alpha <= 1e-3; 0 = off */
  int max_ls_fail;        /* stop (status STALLED) after this many failed line searches in one solve; 0 = off.
                             IPOPT would enter its restoration phase at the first one and, on a locally infeasible
                             problem, end with 'restoration failed'                                       (8) */
  int warm_shift;         /* 0 (do_mpc: previous solution re-used as is) | 1: a warm start shifts the previous solution
                             by one interval (x_k <- x_{k+1}, ..., last interval repeated) before solving      (0) */
  int warm_reset_on_fail; /* 1: a warm start after a solve that did NOT converge keeps that solve's primal point but
                             restarts the equality multipliers at 0 and the barrier at mu_init (IPOPT's default
                             warm_start_init_point=no never re-uses multipliers; ours are re-used after a converged
                             solve only, the ones of a failed solve are what diverged) | 0: always re-use       (1) */
  int periodic_tables;    /* 0: tables extrapolate linearly beyond their ends (CasADi's interpolant, the reference) | 1: the track is a
                             closed loop, tables are evaluated at s modulo their span (buckmore: kappa, n_left, n_right
                             agree at both ends, v_ref to 0.2 %), so that the closed loop can run lap after lap; the kink
                             at the seam is not rounded                                      (0) */
  int max_soc;            /* second-order corrections per iteration (IPOPT's max_soc): when the full step is rejected and
                             does not reduce the constraint violation, the step is re-computed with the constraint
                             residuals of the trial point added (same KKT matrix, new right-hand side: a Riccati sweep
                             over the vectors only).  Implemented in the oracle, where it was measured not to change
                             the iteration counts of this NLP; the device library accepts 0 only    (0) */
  int resto_sticky;       /* > 0: an instance whose solve entered the restoration phase from the hard constraints, or ended
                             INFEASIBLE, starts its next resto_sticky warm-started solves directly in elastic mode (no
                             second jam on the hard constraints first: 10 - 15 iterations saved per solve, and these are
                             the instances that make the tail of a tick); the count is renewed while that keeps
                             happening.  A solve that starts in elastic mode ends like any restoration: back on the hard
                             constraints when every elastic variable is <= tol (status SOLVED), INFEASIBLE otherwise.
                             0: every solve starts on the hard constraints (IPOPT).                              (0) */
  int node0_check;        /* do_mpc registers the track constraints at the nodes 0 .. N-1 (controller.py:69-70); node 0 is the measured
                             state, so its two rows are constants: they cannot change the minimiser, but when x0 violates them
                             the reference's NLP has no feasible point whatever the solver does.  1: the solve runs as always
                             (the rows of node 0 left out), and a measured state with g(x0) > acceptable_tol turns a converged
                             status into INFEASIBLE (violation = g(x0)), one with tol < g(x0) <= acceptable_tol turns SOLVED into
                             ACCEPTABLE (IPOPT's error then cannot fall below g(x0)).  0: node 0 ignored (round 2).       (1) */
  int warm_fallback_iter; /* safeguard of options.mu_init_warm: a warm-started solve that began at mu_init_warm and has gone this
                             many iterations without a decrease of the barrier parameter (it is cycling: the small barrier
                             keeps it at a point that is not central) starts again from its current primal point with the
                             equality multipliers at 0 and the barrier at mu_init, once per solve.  0 = off.          (25) */
  int resto_shift_retry;  /* 1: before the restoration phase proper, a WARM-started solve whose line search has failed (or
                             stalled) starts once more on the hard constraints from its own starting point moved one interval
                             ahead (options.warm_shift's rule), multipliers 0, barrier at mu_init.  do_mpc re-uses the previous
                             solution un-shifted, one interval behind the new measured state, and that mismatch is what jams
                             many of these solves: from the shifted point they converge in ~15 iterations where the elastic
                             problem started at the jam point drifts into a local minimum of the violation.
                             0: straight to the restoration phase (round 2).  Ignored with warm_shift = 1.            (1) */
  int max_mu_stay;        /* warm-started solves: after this many iterations without a decrease of the barrier parameter the
                             iterates are wandering or cycling (the filter holds 16 pairs and forgets the oldest; a solve that
                             converges needs ~60 at most on this NLP; one cycling instance running to max_iter = 1000 costs a
                             tick of 8192 instances 200 ms).  On the hard constraints the recovery steps take over (shifted
                             restart, restoration phase), elsewhere the solve ends with status STALLED.  0 = off.        (100) */
  int infeasible_sticky;  /* 1: a warm-started solve that follows one the solver ended INFEASIBLE (its own verdict, at the largest
                             penalty - not the node-0 rule) starts where that one ended: in the escalated elastic problem, from
                             its primal point and multipliers, instead of going through hard attempt, shifted restart and the
                             elastic problem at resto_rho again (~100 passes; a car that cannot make the corner stays infeasible
                             for several ticks, and such chronic cases are the slowest instances of every tick).  It ends like
                             any restoration: back on the hard constraints when every elastic variable is <= tol (SOLVED),
                             INFEASIBLE otherwise.  0: every solve starts on the hard constraints (IPOPT).              (1) */
  int latency_mode;       /* which evaluation kernels a handle uses, fixed at create: 2 = thread per (interval, instance)
                             (fewest instructions per instance: throughput), 1 = 8 lanes per (interval, instance)
                             (k_eval8 / k_expand8: a third of the latency per launch, 3x the time at full load),
                             0 = 1 for batches of at most 64 instances, else 2.  The two agree to ~1e-12 per iteration;
                             bit-identical results across batches hold between handles of the same mode.     (0) */
} ltompc_options;

typedef struct ltompc_solver* ltompc_handle;

void ltompc_default_params(ltompc_params* p);
void ltompc_default_options(ltompc_options* o);
const char* ltompc_last_error(void);
const char* ltompc_version(void);

/* Replaces Controller(model, control_costs, n_horizon, t_step) + mpc.setup()   (controller.py:9-34).
 * tables: LTOMPC_TABLE_ROWS x n_table doubles, row-major (rows as named above; path.py:96-101,
 *         mpc/track.py:30-42).  n_horizon = N; batch = number of independent MPC instances;
 * device: HIP device ordinal.  Allocates every device workspace; no allocation happens per call later. */
int ltompc_create(const ltompc_params* params, const ltompc_options* options, const double* tables,
                  int n_table, int n_horizon, int batch, int device, ltompc_handle* out);
int ltompc_destroy(ltompc_handle h);

/* Run on this HIP stream (hipStream_t as void*); NULL = the handle's own stream. */
int ltompc_set_stream(ltompc_handle h, void* hip_stream);

/* Replaces `mpc.x0 = x0; mpc.set_initial_guess()`  (mpc.py:117-118): every state slot of instance b := x0[b],
 * every input := 0, u_prev := 0, multipliers reset.  x0: batch x 8 (host). */
int ltompc_set_initial_guess(ltompc_handle h, const double* x0);

/* Replaces `u0 = mpc.make_step(x0)`  (mpc.py:142).  Solves the B NLPs warm-started from the handle's
 * previous solution (un-shifted, like do_mpc), u_prev := last returned u0.
 *   x0: batch x 8 (host, in)      u0: batch x 2 (host, out)
 *   status, iters: batch ints (host, out; may be NULL). */
int ltompc_make_step(ltompc_handle h, const double* x0, double* u0, int* status, int* iters);

/* Same, device pointers, enqueue only (x0_dev: batch x 8, u0_dev: batch x 2, row-major, on the handle's
 * device; u0_dev may be NULL).  Converged instances idle; the host stops launching iterations once a polled
 * device counter says every instance has terminated (ltompc_set_poll_every).  Kernels run on the handle's
 * stream; the call returns after the last poll, the final u0 store may still be in flight on that stream. */
int ltompc_make_step_dev(ltompc_handle h, const double* x0_dev, double* u0_dev);

/* Waits until everything the handle has enqueued on its stream (the _dev entry points only enqueue) has completed. */
int ltompc_synchronize(ltompc_handle h);

/* Closed-loop rollout with FREE-RUNNING instances: every instance does n_ticks of the reference's loop body (src/mpc.py:140-153:
 * u0 = make_step(x0); x0 = plant(x0, u0)), but an instance that has converged takes its plant step and starts its next tick
 * inside the running batch instead of waiting for the slowest instance of the tick (no coupling exists between instances;
 * with make_step 40 % of a tick is spent on the 4 % of instances that need many iterations).  Per instance the
 * sequence of solves is exactly the one of make_step + plant_step in a loop: controls, statuses and iteration counts are
 * bit-identical (tests/test_gpu_parity.py).  Starts from the handle's current guess / warm start like make_step.
 *   x_dev: batch x 8 (device, in: states at tick 0, out: states after n_ticks); n_sub: RK4 sub-steps of the plant;
 *   u_log_dev: batch x n_ticks x 2, status_log_dev, iters_log_dev: batch x n_ticks (device, out; any may be NULL).
 * Synchronous.  Afterwards the handle holds every instance's last solution (warm start of a following make_step / rollout). */
int ltompc_rollout_dev(ltompc_handle h, double* x_dev, int n_ticks, int n_sub, double* u_log_dev, int* status_log_dev,
                       int* iters_log_dev);
/* interior-point iterations launched and kernel launches of the last rollout */
int ltompc_rollout_info(ltompc_handle h, long long* iterations, long long* launches);

/* Predicted trajectories of the last solve (do_mpc: mpc.opt_x_num['_x', k, 0, -1], ['_u', k, 0]).
 *   X: batch x (N+1) x 8, U: batch x N x 2 (host, out; either may be NULL). */
int ltompc_get_prediction(ltompc_handle h, double* X, double* U);

/* Per-instance results of the last solve (host, out; any may be NULL):
 *   status, iters: ints;  kkt_error: scaled KKT error E_0;  objective: NLP objective J;  mu: last barrier. */
int ltompc_get_stats(ltompc_handle h, int* status, int* iters, double* kkt_error, double* objective,
                     double* mu);

/* Full primal/dual iterate of the last solve, for KKT checks (host, out; any may be NULL):
 *   X: B x (N+1) x 8, C: B x N x 8 (first Radau point), U: B x N x 2,
 *   L1, L2: B x N x 8 (collocation multipliers).                                   */
int ltompc_get_iterate(ltompc_handle h, double* X, double* C, double* U, double* L1, double* L2);

/* Slacks t and multipliers nu of the inequality constraints of the last solve (host, out; any may be NULL):
 * B x N x n_ineq, per interval k: input bounds (lower, upper per input), bounds on the Radau point, bounds on
 * node k+1 (per state lower then upper, only the bounds that are set), gL, gR+, gR- at node k+1. */
int ltompc_get_ineq(ltompc_handle h, double* T, double* NU, int* n_ineq);

/* Plant step: replaces sim.make_step(u0) (src/mpc/simulator.py:18-20, mpc.py:143): integrates the model ODE
 * over t_step under zero-order-hold u with n_sub classical RK4 sub-steps.  x, x_next: batch x 8; u: batch x 2
 * (host).  _dev variant: device pointers, enqueue only. */
int ltompc_plant_step(ltompc_handle h, const double* x, const double* u, int n_sub, double* x_next);
int ltompc_plant_step_dev(ltompc_handle h, const double* x_dev, const double* u_dev, int n_sub,
                          double* x_next_dev);

/* Slip angles and Pacejka lateral forces (model.py:101-114; called per tick at mpc.py:148-150).
 * x: batch x 8 -> alpha: batch x 2 (front, rear), Fy: batch x 2.  Host pointers, computed on the device. */
int ltompc_slip_forces(ltompc_handle h, const double* x, int batch, double* alpha, double* Fy);

/* Device-pointer form of ltompc_set_initial_guess (enqueue only). */
int ltompc_set_initial_guess_dev(ltompc_handle h, const double* x0_dev);

/* Per-instance counters of the last solve: Hessian-regularisation retries, failed line searches (host, out). */
int ltompc_get_counters(ltompc_handle h, int* n_reg, int* n_lsfail);

/* Restoration phase of the last solve (host, out; any may be NULL): how often an instance entered it (0 or 1), and the
 * largest elastic variable at termination, i.e. the remaining violation of the track constraints in metres (> tol for
 * status INFEASIBLE; 0 for instances that ended on the hard constraints). */
int ltompc_get_restoration(ltompc_handle h, int* n_resto, double* violation);

/* Recovery steps of the last solve (host, out; any may be NULL): n_shift - the solve started again from its shifted starting
 * point (options.resto_shift_retry; 0 or 1); n_fallback - a tuned warm start fell back to mu_init (options.warm_fallback_iter);
 * g0 - the largest track constraint at the measured state x0 (options.node0_check; > 0: x0 is outside the band);
 * solver_status - the solver's own termination status, before the node-0 rule; penalty - the elastic variables' penalty at
 * termination (0: hard constraints; resto_rho_max for status INFEASIBLE after an escalated restoration). */
int ltompc_get_recovery(ltompc_handle h, int* n_shift, int* n_fallback, double* g0, int* solver_status, double* penalty);

/* Profiling: when on, every kernel launch of make_step is bracketed by HIP events on the handle's stream and
 * ltompc_get_timing returns the accumulated device time per kernel class since profiling was switched on:
 * index 0 eval (k_eval or k_eval8), 1 riccati (8 instances per wavefront), 2 expand (k_expand or k_expand8), 3 linesearch,
 * 4 pick, 5 update, 6 riccati1 (the one-instance-per-workgroup sweeps of the narrow launches: k_riccati1, one wavefront, and k_riccati1q, four), 7 step1 (their fused
 * line-search / pick / update kernel).  launches / ip_iterations (iterations launched) refer to the last make_step.
 * Arrays have 8 entries.  Any output may be NULL.
 * on = 0 off, 1 every launch, 2 + c: only the launches of kernel class c (two events per iteration instead of seven:
 * at 120 k solves/s bracketing every launch costs 4 % of the throughput, bracketing one class 1 %). */
int ltompc_set_profiling(ltompc_handle h, int on);
int ltompc_get_timing(ltompc_handle h, double* ms_by_kernel8, int* launches_by_kernel8, int* launches,
                      int* ip_iterations);
/* Per-launch log of the profiled make_steps since profiling was switched on: kernel class (index as above), launch
 * width (instances the launch was sized for: the whole batch until the first re-packing of the unfinished instances)
 * and device time in ms.  Returns the number of log entries (copies at most `capacity`); negative on error. */
int ltompc_get_launch_log(ltompc_handle h, int* kind, int* width, double* ms, int capacity);

/* Interior-point iteration (0-based, within its make_step) of every entry of the launch log; returns the number of entries. */
int ltompc_get_launch_log_iterations(ltompc_handle h, int* iteration, int capacity);

/* Number of unfinished instances after every iteration of the last make_step (entry i: instances that passed the
 * termination test of iteration i and went on); returns the number of iterations launched (0 after a rollout, which keeps
 * no per-iteration counts). */
int ltompc_get_active_history(ltompc_handle h, int* active, int capacity);

/* Histogram of the per-instance statuses of the last solve (counts8[s], s = LTOMPC_STATUS_*; entry 7 collects anything
 * else) and the sum of the instances' iteration counts, reduced on the device: cheaper than ltompc_get_stats and does
 * not disturb the packed order of the instances.  Either output may be NULL. */
int ltompc_get_status_counts(ltompc_handle h, int* counts8, long long* iterations_sum);
/* ... and the histogram of the solver's own statuses (before the node-0 rule of options.node0_check) that the last
 * ltompc_get_status_counts call reduced along with it. */
int ltompc_get_solver_status_counts(ltompc_handle h, int* counts8);

/* Poll history of the last make_step: up to `capacity` triples (iteration, unfinished instances, launch width);
 * returns the number of polls (>= 0). */
int ltompc_get_history(ltompc_handle h, int* triples, int capacity);

/* make_step polls the device's count of unfinished instances every n interior-point iterations (default 4). */
int ltompc_set_poll_every(ltompc_handle h, int n);

/* Launches of at most `width` unfinished instances use the one-instance-per-workgroup kernels (k_riccati1 / k_riccati1q,
 * k_step1), wider ones the full-width kernels; default 512, 0 = never, at most 512.  Scheduling only: an instance's result does
 * not depend on it (tests/test_gpu_parity.py).  With several handles ticking beside each other on one GPU a lower switch-over
 * is faster (their one-instance workgroups queue for the same CUs): four handles of 2048, 128 instead of 512: +2 %. */
int ltompc_set_narrow_width(ltompc_handle h, int width);

/* Debug hook: raw copy of a device work array in its device layout (see csrc/kernels.h); returns its size in bytes.
 * Work arrays are indexed by the physical slot of an instance; a solve that re-packed its unfinished instances
 * (more than 4 iterations) uses the step buffers as temporaries when it restores the caller's order,
 * so their content is meaningful after solves capped at a few iterations only (which is what the tests do). */
long long ltompc_debug_fetch(ltompc_handle h, int which, void* out, long long nbytes);

/* Test hook (not part of the reference surface): model derivatives at n points, computed by the same device
 * functions the solver kernels use.  x, lam: n x 8 -> f: n x 8; J (df/dx), H (sum_i lam_i d2f_i): n x 8 x 8;
 * cost value / gradient / Hessian of lterm and mterm: n x 2 [x 8 [x 8]]; constraints gL, gR+, gR-:
 * n x 3 [x 8 [x 8]].  eps = table smoothing length. */
int ltompc_test_model(ltompc_handle h, int n, double eps, const double* x, const double* lam, double* f,
                      double* J, double* H, double* cval, double* cgrad, double* cH, double* gval,
                      double* ggrad, double* gH);

/* Test hook: the friction-ellipse constraints (front, rear; params.ell_*) at n points: val n x 2, grad n x 2 x 8, H n x 2 x 8 x 8. */
int ltompc_test_ellipse(ltompc_handle h, int n, const double* x, double* val, double* grad, double* H);

/* ---------------------------------------------------------------------------------------------------------------------
 * Velocity-profile generator: the producer of velocities.json, the v_ref table of the hot path.
 * Replaces VelocityProfile(vehicle, s, k, s_max) of the reference (src/velocity.py:14-76, called from
 * src/trajectory.py:47-52): local limit sqrt(mu g / k), forward pass limited by engine force and remaining traction,
 * backward pass limited by the remaining traction, v = min of the two; a batch of independent profiles per call (the
 * race-line optimisers evaluate one profile per candidate line).  One GPU thread per profile (the passes are sequential
 * scans over the samples). */
typedef struct ltompc_vp_vehicle {
  int kind;                     /* 0: engine map + friction circle (src/vehicle.py:11-35), 1: MX-5 (src/vehicleMX5.py:11-79) */
  int n_map;                    /* kind 0: points of the engine map (<= 16) */
  double mass, friction_coef;   /* friction_coef: `frictionCoefficient` / `control.lambda` (local limit; kind 0: traction too) */
  double lam, D;                /* kind 1: traction(v, k, lam=2.0): F_max = lam * D * mass * g with D = (D_f + D_r) / 2 */
  double T, C_m, Cr_0, Cr_2;    /* kind 1: engine_force(v) = T C_m - Cr_0 - Cr_2 v^2 */
  double map_v[16], map_f[16];  /* kind 0: engine_force(v) = np.interp(v, map_v, map_f) */
} ltompc_vp_vehicle;

/* s, k: batch x n (arc length and curvature > 0 of the samples, WITHOUT the overlapping end point of a closed path);
 * s_max: batch (length of the closed path; < 0: open path, `s_max=None` in the reference).  Outputs batch x n, host
 * pointers, any of v_local / v_acclim / v_declim may be NULL.  Returns 0, or < 0 with ltompc_last_error(). */
int ltompc_velocity_profile(int device, const ltompc_vp_vehicle* vehicle, int n, int batch, const double* s, const double* k,
                            const double* s_max, double* v, double* v_local, double* v_acclim, double* v_declim);

#ifdef __cplusplus
}
#endif
#endif /* LTOMPC_H */
