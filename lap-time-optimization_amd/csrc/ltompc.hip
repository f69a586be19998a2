// ltompc.hip — host side of libltompc.so: the C ABI of include/ltompc.h over the kernels of kernels.h.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared ltompc.hip -o libltompc.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "kernels.h"

using namespace ltompc;

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) {
  g_err = msg;
  return -1;
}
#define HIPCHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

constexpr int NKERN = 8;  // eval, riccati, expand, linesearch, pick, update, riccati1, step1

}  // namespace

struct ltompc_solver {
  Consts K;
  Work W;
  int device = 0, N = 0, B = 0, Bp = 0, n_table = 0, max_iter = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::vector<void*> allocs;
  double* d_tables = nullptr;
  double* d_x0_rm = nullptr;   // staging, row-major B x 8
  double* d_u0_rm = nullptr;   // staging, row-major B x 2
  double* d_io = nullptr;      // staging for plant / slip forces
  int* h_active = nullptr;     // pinned
  int *d_act[2] = {nullptr, nullptr}, *d_nact[2] = {nullptr, nullptr};  // ping-pong lists of unfinished instances
  int last_compactions = 0;
  int *d_perm = nullptr, *d_orig = nullptr;  // packing: permutation of the current re-packing, original index of every physical slot
  int ls_width_env = 0;  // LTOMPC_LSW: fixed launch width (instances) of the second line-search phase (0: sized by the last list)
  int pack_num = 6;  // re-pack when at most pack_num / 8 of the launch width is still unfinished (LTOMPC_PACK_NUM)
  bool packed = false;  // the instances are in packed order (d_orig); make_step keeps them so, the accessors restore the caller's order first
  bool packing = true;  // LTOMPC_PACK=0: re-pack the list of unfinished instances only, leave their data where it is
  std::vector<int> history;  // (iteration, n_active, n_launch) triples of the last make_step's polls
  bool cold_next = true;
  bool after_rollout = false;  // the last solve was a rollout: W.active holds no per-iteration counts
  int poll_every = 4;
  int profiling = 0;  // 0 off, 1 every launch bracketed, 2 + c: launches of kernel class c only
  bool compaction = true;       // LTOMPC_COMPACT=0 switches the re-packing of unfinished instances off
  bool serial_riccati = false;  // LTOMPC_RICCATI=serial selects the one-thread-per-instance kernel (A/B checks)
  // profiling: ONE event before every launch (and one closing a run of launches before the host waits); a launch's
  // duration is the time to the next event.  Events come from a pool that lives as long as the handle.
  std::vector<hipEvent_t> ev, ev_pool;
  size_t ev_pool_used = 0;
  std::vector<int> ev_kind, ev_width, ev_iter;
  std::vector<int> log_kind, log_width, log_iter;  // per launch of the profiled make_steps since profiling was switched on
  std::vector<double> log_ms;
  int cur_width = 0;  // instances in the launches being issued
  int cur_iter = 0;   // interior-point iteration the launches being issued belong to
  int* d_counts = nullptr;  // 8 status counters + 1 x 64-bit iteration sum + 8 solver-status counters (k_status_counts)
  int solver_counts[8] = {};  // ... the last ones read (ltompc_get_solver_status_counts)
  // rollout: per pass (ring slot) the number of instances that still have ticks to do and the list of the instances that
  // converged in it; their plant steps run beside the solver, on two low-priority streams in turn (a plant kernel takes 1 - 2 ms,
  // longer than a pass: one stream cannot keep up, 1870 ms instead of 1170 ms for 20 ticks; four are slower than two)
  static constexpr int ROLL_RING = 16, ROLL_PLANTS = 4;
  int* d_roll = nullptr;    // [ROLL_RING][2]: instances not FINAL, length of the plant list
  int* d_plist = nullptr;   // [ROLL_RING][Bp]
  hipStream_t plant_streams[ROLL_PLANTS] = {};
  int n_plant_streams = 2;  // LTOMPC_PLANT_STREAMS (1 .. ROLL_PLANTS)
  hipEvent_t ev_fin[ROLL_RING] = {}, ev_done[ROLL_RING] = {};  // pass finished (solver stream) / its plant steps done (plant stream)
  long long roll_iterations = 0, roll_launches = 0;
  double ms_by_kernel[NKERN] = {};
  int launches_by_kernel[NKERN] = {};
  Consts* d_K = nullptr;  // device copies of K and W for the solver kernels
  Work* d_W = nullptr;
  bool bounds_ref = false;  // the parameters have the reference's bound pattern: kernels instantiated for it (LTOMPC_BOUNDS=any: never)
  bool ref_eval = false, ref_expand = false, ref_ls = false, ref_step1 = false;  // per kernel (LTOMPC_BOUNDS=eval,expand,...: those run generic)
  bool eval8 = true;  // LTOMPC_EVAL=slot: thread-per-slot k_eval / k_expand instead of the wave-cooperative k_eval8 / k_expand8
  int step1_width = 512;  // LTOMPC_STEP1: launches of at most this many instances use the fused step-selection kernel (0 = never)
  int sweeps_width = 16;  // LTOMPC_SWEEPS_W: launches of at most this many instances repeat a failed Riccati sweep inside the launch (up to 4 attempts)
  int ric1q_width = 16;  // LTOMPC_RIC1Q: launches of at most this many instances use the four-wavefront form of the single-instance sweep (0 = never;
                         // beside other handles' kernels its 4-wavefront workgroups gain nothing at 64 and lose 1 % at 512, alone it is 7 % faster)
  int ric1_width = 512;  // LTOMPC_RIC1: launches of at most this many instances use the one-wavefront-per-instance sweep (0 = never)
  int last_launches = 0, last_iterations = 0;

  // (P may be a gptr<T>: a global-address-space pointer in the device pass of the compiler, a plain one on the host)
  // work = true: an array that the kernels fill before they read it.  LTOMPC_POISON=1 (debug) fills those with 0xFF bytes
  // (NaN as doubles) instead of zeros, so that a read of a never-written word shows up in the results
  // (tests/test_gpu_parity.py::test_poisoned_work_buffers_give_identical_results).
  bool poison = false;
  template <typename P>
  int dalloc(P* p, size_t n, bool work = false) {
    using T = std::remove_pointer_t<P>;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) return fail(std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemsetAsync(q, (work && poison) ? 0xFF : 0, n * sizeof(T), stream);
    if (e != hipSuccess) return fail(std::string("hipMemset: ") + hipGetErrorString(e));
    allocs.push_back(q);
    *p = (P)q;
    return 0;
  }
};

namespace {

void build_bounds(const ltompc_params& p, Bounds& b) {
  std::memset(&b, 0, sizeof b);
  for (int i = 0; i < NX; i++) {  // per state: lower, then upper (same order as the oracle)
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

struct Launcher {
  ltompc_solver* h;
  int launches = 0;
  size_t lds = 0;  // dynamic LDS bytes of the next launch (reset after it)
  int block_threads = 64;  // workgroup size of the next launch (reset after it)
  int stamp(int kind) {  // kind < 0: closes the preceding launch without opening one
    if (h->ev_pool_used == h->ev_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return fail("hipEventCreate failed");
      h->ev_pool.push_back(e);
    }
    hipEvent_t e = h->ev_pool[h->ev_pool_used++];
    if (hipEventRecord(e, h->stream) != hipSuccess) return fail("hipEventRecord failed");
    h->ev.push_back(e), h->ev_kind.push_back(kind), h->ev_width.push_back(h->cur_width), h->ev_iter.push_back(h->cur_iter);
    return 0;
  }
  int close() { return (h->profiling && !h->ev_kind.empty() && h->ev_kind.back() >= 0) ? stamp(-1) : 0; }
  template <typename Kern, typename... Args>
  int run(int kind, Kern kern, int threads_total, Args... args) {
    dim3 block(block_threads), grid((threads_total + block_threads - 1) / block_threads);
    const size_t lds_bytes = lds;
    lds = 0, block_threads = 64;
    if (h->profiling == 1) {
      if (stamp(kind)) return -1;
    } else if (h->profiling >= 2) {  // one class only: open before its launches, close before the next launch of another class
      if (kind == h->profiling - 2) {
        if (stamp(kind)) return -1;
      } else if (!h->ev_kind.empty() && h->ev_kind.back() >= 0 && stamp(-1)) return -1;
    }
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, h->stream, args...);
    launches++;
    return 0;
  }
};

// The kernels of one interior-point iteration over the instances of `la` (make_step and the closed-loop rollout share it).
int launch_iteration(ltompc_solver* h, Launcher& L, const Launch& la, const int it, const int n_launch, const int ls_width, const bool ell,
                     const bool riccati_only) {
  const int N = h->N, np = la.n_pad;
    if (h->eval8 ? L.run(0, k_eval8, N * np * 8, h->d_K, h->d_W, la) : L.run(0, ell ? (h->ref_eval ? k_eval<BoundsRef, true> : k_eval<BoundsAny, true>) : (h->ref_eval ? k_eval<BoundsRef, false> : k_eval<BoundsAny, false>), N * np, h->d_K, h->d_W, la)) return -1;
    if (h->serial_riccati) {
      if (L.run(1, k_riccati, np, h->d_K, h->d_W, la, it)) return -1;
    } else {
      // measured: letting the stragglers retry inside a launch (max_sweeps 4 when n_launch <= 256) finishes them in
      // fewer launches but doubles the time of every narrow launch: 193 ms vs 145 ms per tick at B = 8192
      const int max_sweeps = 1;
      if (n_launch <= h->ric1_width) {
        if (n_launch <= h->ric1q_width) {  // four wavefronts per instance
          L.lds = ric1q_lds_bytes(N), L.block_threads = 256;
          if (L.run(6, k_riccati1q, n_launch * 256, h->K, h->W, la, it, n_launch <= h->sweeps_width ? 4 : 1)) return -1;
        } else {
          L.lds = ric1_lds_bytes(N);
          if (L.run(6, k_riccati1, n_launch * 64, h->K, h->W, la, it, n_launch <= h->sweeps_width ? 4 : 1)) return -1;  // one wavefront per instance
        }
      } else if (L.run(1, k_riccati8, np * 8, h->K, h->W, la, it, max_sweeps)) return -1;  // 8 lanes per instance
    }
    if (riccati_only) return 0;  // (make_step's last pass: only finalises the statuses, MAX_ITER)
    if (h->eval8 ? L.run(2, k_expand8, N * np * 8, h->d_K, h->d_W, la) : L.run(2, ell ? (h->ref_expand ? k_expand<BoundsRef, true> : k_expand<BoundsAny, true>) : (h->ref_expand ? k_expand<BoundsRef, false> : k_expand<BoundsAny, false>), N * np, h->d_K, h->d_W, la)) return -1;
    if (n_launch <= h->step1_width) {
      // one workgroup per instance does both line-search phases, the filter test and the update
      L.block_threads = 320;
      if (L.run(7, ell ? (h->ref_step1 ? k_step1<BoundsRef, true> : k_step1<BoundsAny, true>) : (h->ref_step1 ? k_step1<BoundsRef, false> : k_step1<BoundsAny, false>), n_launch * 320, h->d_K, h->d_W, la)) return -1;
    } else {
      const auto kls = ell ? (h->ref_ls ? k_linesearch<BoundsRef, true> : k_linesearch<BoundsAny, true>) : (h->ref_ls ? k_linesearch<BoundsRef, false> : k_linesearch<BoundsAny, false>);
      if (L.run(3, kls, N * np, h->d_K, h->d_W, la, 0, np)) return -1;
      if (L.run(4, k_pick, np * 8, h->d_K, h->d_W, la, 0)) return -1;  // 8 lanes per instance
      if (h->K.o.n_linesearch > 1) {  // remaining step candidates, only for instances whose full step was rejected
        const int jw = np < ls_width ? np : ls_width;  // launch width of the second phase (longer lists are covered grid-stride)
        if (L.run(3, kls, (h->K.o.n_linesearch - 1) * N * jw, h->d_K, h->d_W, la, 1, jw)) return -1;
        if (L.run(4, k_pick, jw * 8, h->d_K, h->d_W, la, 1)) return -1;
      }
      if (L.run(5, k_update, N * np, h->d_K, h->d_W, la)) return -1;
    }
  return 0;
}

int collect_profile(ltompc_solver* h) {
  for (size_t i = 0; i + 1 < h->ev_kind.size(); i++) {
    if (h->ev_kind[i] < 0) continue;
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    h->ms_by_kernel[h->ev_kind[i]] += ms;
    h->launches_by_kernel[h->ev_kind[i]] += 1;
    h->log_kind.push_back(h->ev_kind[i]), h->log_width.push_back(h->ev_width[i]), h->log_iter.push_back(h->ev_iter[i]), h->log_ms.push_back(ms);
  }
  h->ev.clear(), h->ev_kind.clear(), h->ev_width.clear(), h->ev_iter.clear();
  h->ev_pool_used = 0;
  return 0;
}

// planes [F][NK][Bp] (device) -> batch-major B x NK x F (host)
int planes_to_host(ltompc_solver* h, const double* dev, int F, int NK, double* out) {
  if (!out) return 0;
  std::vector<double> tmp((size_t)F * NK * h->Bp);
  HIPCHECK(hipMemcpyAsync(tmp.data(), dev, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->B; b++)
    for (int k = 0; k < NK; k++)
      for (int f = 0; f < F; f++) out[((size_t)b * NK + k) * F + f] = tmp[((size_t)f * NK + k) * h->Bp + b];
  return 0;
}


// Back to the caller's order: slot j returns to orig[j].  make_step leaves the instances in the order of its last
// re-packing (it only maps x0 in and u0 out through `orig`); everything that exposes per-instance arrays calls this first.
int ensure_unpacked(ltompc_solver* h) {
  if (!h->packed) return 0;
  const int B = h->B, N = h->N;
  hipLaunchKernelGGL(k_pack_inverse, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->d_orig, h->d_perm, B);
  const int nthreads = (N + 1) * B;
  for (int pass = 0; pass < 2; pass++)
    hipLaunchKernelGGL(k_pack, dim3((nthreads + 255) / 256), dim3(256), 0, h->stream, h->W, h->d_perm, (const int*)nullptr, B, h->d_orig,
                       h->K.bd.ni, h->K.bd.nel, pass);
  HIPCHECK(hipGetLastError());
  h->packed = false;
  return 0;
}

}  // namespace

extern "C" {

const char* ltompc_last_error(void) { return g_err.c_str(); }
const char* ltompc_version(void) { return "ltompc 0.6 (gfx950, fp64; interior point with shifted-restart + elastic restoration and penalty escalation, block-structured interval evaluation, LDS-staged wave-cooperative Riccati, data re-packing)"; }

void ltompc_default_params(ltompc_params* p) {
  std::memset(p, 0, sizeof *p);
  // data/vehicles/MX5.json as read by model.py:42-64; D_f, D_r keep the constructor default 1.0 (model.py:22,26)
  p->mass = 1000.0, p->inertia_z = 1000.0, p->length_f = 1.5, p->length_r = 1.5, p->width = 2.3;
  p->B_f = 10.0, p->C_f = 1.3, p->D_f = 1.0, p->B_r = 12.0, p->C_r = 1.2, p->D_r = 1.0;
  p->C_m = 1000.0, p->Cr_0 = 0.01, p->Cr_2 = 0.0003, p->gravity = 9.81;
  // controller.py:29,52-53; mpc.py:104
  p->q_n = 0.5, p->q_mu = 3.0, p->q_vy = 1.0, p->q_v = 1.0, p->vref_scale = 0.6, p->q_B = 1e-2;
  p->r_du[0] = p->r_du[1] = 1e-2;
  // controller.py:79-103
  for (int i = 0; i < NX; i++) p->x_lb[i] = -LTOMPC_NO_BOUND, p->x_ub[i] = LTOMPC_NO_BOUND;
  const double pi = 3.14159265358979323846;
  p->x_lb[0] = 0.0;
  p->x_lb[2] = -pi * 0.5, p->x_ub[2] = pi * 0.5;
  p->x_lb[3] = 0.0;
  p->x_lb[6] = -pi / 4, p->x_ub[6] = pi / 4;
  p->x_lb[7] = -1.0, p->x_ub[7] = 1.0;
  p->u_lb[0] = -2 * pi / 4, p->u_ub[0] = 2 * pi / 4;
  p->u_lb[1] = -1.0, p->u_ub[1] = 1.0;
}

void ltompc_default_options(ltompc_options* o) {
  std::memset(o, 0, sizeof *o);
  o->t_step = 0.1, o->tol = 1e-8, o->acceptable_tol = 1e-6, o->mu_init = 0.1, o->mu_min = 1e-9;
  o->kappa_eps = 10, o->kappa_mu = 0.2, o->theta_mu = 1.5, o->tau_min = 0.99, o->bound_push = 1e-2;
  o->s_max = 100, o->delta_w_first = 1e-4, o->smooth_eps_min = 1e-4, o->smooth_scale = 1.0;
  o->max_iter = 1000, o->acceptable_iter = 15, o->n_linesearch = 8, o->stall_iter = 15, o->max_ls_fail = 8;
  o->warm_reset_on_fail = 1;
  o->resto_rho = 1000.0, o->max_soc = 0, o->resto_sticky = 0;
  o->resto_rho_max = 1e6, o->resto_rho_factor = 1e3, o->dual_inf_max = 1e4, o->max_mu_stay = 100, o->infeasible_sticky = 1, o->node0_check = 1, o->warm_fallback_iter = 25, o->resto_shift_retry = 1;
}

int ltompc_create(const ltompc_params* params, const ltompc_options* options, const double* tables, int n_table,
                  int n_horizon, int batch, int device, ltompc_handle* out) {
  if (!params || !options || !tables || !out) return fail("ltompc_create: null argument");
  if (n_table < 4) return fail("ltompc_create: n_table must be >= 4");
  if (n_horizon < 2 || n_horizon > 4096) return fail("ltompc_create: n_horizon out of range [2, 4096]");
  if (batch < 1) return fail("ltompc_create: batch must be >= 1");
  if (options->n_linesearch < 1 || options->n_linesearch > MAX_LS) return fail("ltompc_create: n_linesearch out of range");
  if (!(options->t_step > 0)) return fail("ltompc_create: t_step must be positive");
  if (!(options->soft_rho >= 0) || !std::isfinite(options->soft_rho)) return fail("ltompc_create: soft_rho must be >= 0 (0 = hard track constraints)");
  if (!(options->resto_rho >= 0) || !std::isfinite(options->resto_rho)) return fail("ltompc_create: resto_rho must be >= 0 (0 = no restoration phase)");
  if (!(params->ell_penalty >= 0) || !std::isfinite(params->ell_penalty)) return fail("ltompc_create: ell_penalty must be >= 0 (0 = no friction-ellipse constraints)");
  if (params->ell_penalty > 0 && (!(params->ell_D_f > 0) || !(params->ell_D_r > 0) || !std::isfinite(params->ell_rho)))
    return fail("ltompc_create: friction-ellipse constraints need ell_D_f > 0, ell_D_r > 0 and a finite ell_rho");
  if (options->resto_sticky < 0) return fail("ltompc_create: resto_sticky must be >= 0");
  if (!std::isfinite(options->resto_rho_max) || !std::isfinite(options->resto_rho_factor) || options->resto_rho_factor < 0)
    return fail("ltompc_create: resto_rho_max and resto_rho_factor must be finite, resto_rho_factor >= 0 (<= 1: no penalty escalation)");
  if (!(options->dual_inf_max >= 0)) return fail("ltompc_create: dual_inf_max must be >= 0 (0 = off)");
  if (options->max_mu_stay < 0) return fail("ltompc_create: max_mu_stay must be >= 0 (0 = off)");
  if (options->warm_fallback_iter < 0) return fail("ltompc_create: warm_fallback_iter must be >= 0 (0 = off)");
  if (options->max_soc != 0) return fail("ltompc_create: max_soc must be 0 (the second-order correction exists in the oracle only, see include/ltompc.h)");
  for (int r = 0; r < LTOMPC_TABLE_ROWS; r++)
    for (int i = 0; i < n_table; i++)
      if (!std::isfinite(tables[(size_t)r * n_table + i])) return fail("ltompc_create: non-finite table entry");
  for (int i = 1; i < n_table; i++)
    if (!(tables[i] > tables[i - 1]) || !(tables[2 * (size_t)n_table + i] > tables[2 * (size_t)n_table + i - 1]))
      return fail("ltompc_create: table grids must be strictly increasing");
  int ndev = 0;
  HIPCHECK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("ltompc_create: no such HIP device");
  HIPCHECK(hipSetDevice(device));
  auto* h = new ltompc_solver();
  h->device = device, h->N = n_horizon, h->B = batch, h->Bp = (batch + 63) / 64 * 64, h->n_table = n_table;
  h->max_iter = options->max_iter;
  h->K.p = *params, h->K.o = *options;
  build_bounds(*params, h->K.bd);
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return fail("hipStreamCreate failed");
  }
  h->stream = h->own_stream;
  {
    const char* e = getenv("LTOMPC_RICCATI");
    h->serial_riccati = e && std::string(e) == "serial";
    const char* po = getenv("LTOMPC_POISON");
    h->poison = po && std::string(po) == "1";
    const char* c = getenv("LTOMPC_COMPACT");
    h->compaction = !(c && std::string(c) == "0");
    const char* pk = getenv("LTOMPC_PACK");
    h->packing = !(pk && std::string(pk) == "0");
    const char* lw = getenv("LTOMPC_LSW");
    if (lw && atoi(lw) >= 64) h->ls_width_env = atoi(lw);
    const char* pn = getenv("LTOMPC_PACK_NUM");
    if (pn && atoi(pn) >= 1 && atoi(pn) <= 7) h->pack_num = atoi(pn);
    const char* ev = getenv("LTOMPC_EVAL");  // slot | wave: overrides options.latency_mode (tests, experiments)
    h->eval8 = options->latency_mode == 1 || (options->latency_mode == 0 && batch <= 64);
    if (ev) h->eval8 = std::string(ev) == "wave";
    if (params->ell_penalty > 0.0) h->eval8 = false;  // (the 8-lanes-per-slot kernels do not have the friction-ellipse constraints)
    {
      unsigned ulb = 0, uub = 0, xlb = 0, xub = 0;
      for (int i = 0; i < 2; i++) ulb |= (params->u_lb[i] > -LTOMPC_NO_BOUND) << i, uub |= (params->u_ub[i] < LTOMPC_NO_BOUND) << i;
      for (int i = 0; i < 8; i++) xlb |= (params->x_lb[i] > -LTOMPC_NO_BOUND) << i, xub |= (params->x_ub[i] < LTOMPC_NO_BOUND) << i;
      const char* bp = getenv("LTOMPC_BOUNDS");  // any: the run-time pattern kernels (tests)
      h->bounds_ref = ulb == BoundsRef::ulb && uub == BoundsRef::uub && xlb == BoundsRef::xlb && xub == BoundsRef::xub &&
                      !(bp && std::string(bp) == "any");
      const std::string sel = bp ? bp : "";
      h->ref_eval = h->bounds_ref && sel.find("eval") == std::string::npos;
      h->ref_expand = h->bounds_ref && sel.find("expand") == std::string::npos;
      h->ref_ls = h->bounds_ref && sel.find("linesearch") == std::string::npos;
      h->ref_step1 = h->bounds_ref && sel.find("step1") == std::string::npos;
    }
    const char* s1 = getenv("LTOMPC_STEP1");
    if (s1) h->step1_width = atoi(s1);
    const char* t = getenv("LTOMPC_RIC1");
    if (t) h->ric1_width = atoi(t);
    const char* npl = getenv("LTOMPC_PLANT_STREAMS");
    if (npl) h->n_plant_streams = std::min((int)ltompc_solver::ROLL_PLANTS, std::max(1, std::atoi(npl)));
    const char* sw = getenv("LTOMPC_SWEEPS_W");
    if (sw) h->sweeps_width = atoi(sw);
    const char* rq = getenv("LTOMPC_RIC1Q");
    if (rq) h->ric1q_width = atoi(rq);
    // k_riccati1 / k_riccati1q stage the whole horizon of an instance in LDS (160 KiB per CU on gfx950)
    if (std::max(ric1_lds_bytes(n_horizon), ric1q_lds_bytes(n_horizon)) > 150 * 1024) h->ric1_width = 0;
    // (the attribute belongs to the kernel, not to the handle: always the cap, so that handles with different horizons
    //  do not lower each other's limit)
    else if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_riccati1), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
             hipFuncSetAttribute(reinterpret_cast<const void*>(k_riccati1q), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      h->ric1_width = 0;
    }
  }
  const size_t N = h->N, Bp = h->Bp;
  const int ni = h->K.bd.ni;
  Work& W = h->W;
  W.N = h->N, W.B = h->B, W.Bp = h->Bp;
  int rc = 0;
  rc |= h->dalloc(&h->d_tables, (size_t)LTOMPC_TABLE_ROWS * n_table);
  rc |= h->dalloc(&W.X, 8 * (N + 1) * Bp, true), rc |= h->dalloc(&W.C, 8 * N * Bp, true), rc |= h->dalloc(&W.U, 2 * N * Bp, true);
  rc |= h->dalloc(&W.L1, 8 * N * Bp, true), rc |= h->dalloc(&W.L2, 8 * N * Bp, true);
  const size_t nel = h->K.bd.nel;
  rc |= h->dalloc(&W.T, (ni + NNL + nel) * N * Bp, true), rc |= h->dalloc(&W.NU, ni * N * Bp, true);  // T: slacks + elastic variables
  rc |= h->dalloc(&W.dX, 8 * (N + 1) * Bp, true), rc |= h->dalloc(&W.dC, 8 * N * Bp, true), rc |= h->dalloc(&W.dU, 2 * N * Bp, true);
  rc |= h->dalloc(&W.nL1, 8 * N * Bp, true), rc |= h->dalloc(&W.nL2, 8 * N * Bp, true);
  rc |= h->dalloc(&W.dT, (ni + NNL + nel) * N * Bp, true), rc |= h->dalloc(&W.dNU, ni * N * Bp, true);
  // (k_riccati8's staging fetches one field past each STAGE block, k <= N - 1: that word is the first of the next block, and
  //  block N, the terminal node, follows the last stage block; no padding needed)
  rc |= h->dalloc(&W.QP, (size_t)QP_NF * (N + 1) * Bp, true), rc |= h->dalloc(&W.RC, (size_t)RC_NF * (N + 1) * Bp, true);
  rc |= h->dalloc(&W.RS, (size_t)RS_NF * N * Bp, true), rc |= h->dalloc(&W.SP, (size_t)SP_NF * N * Bp, true);
  rc |= h->dalloc(&W.LS, (size_t)3 * (options->n_linesearch + 1) * N * Bp, true);
  rc |= h->dalloc(&W.x0, 8 * Bp), rc |= h->dalloc(&W.uprev, 2 * Bp);
  rc |= h->dalloc(&W.st, (size_t)ST_NF * Bp), rc |= h->dalloc(&W.filt, (size_t)2 * FILTER_MAX * Bp);
  rc |= h->dalloc(&W.si, (size_t)SI_NF * Bp), rc |= h->dalloc(&W.active, (size_t)h->max_iter + 2);
  rc |= h->dalloc(&h->d_act[0], Bp), rc |= h->dalloc(&h->d_act[1], Bp), rc |= h->dalloc(&h->d_nact[0], 4), rc |= h->dalloc(&h->d_nact[1], 4);
  rc |= h->dalloc(&W.ls_list, Bp), rc |= h->dalloc(&W.ls_count, 4);
  rc |= h->dalloc(&h->d_perm, Bp), rc |= h->dalloc(&h->d_orig, Bp);
  rc |= h->dalloc(&W.BK, 18 * N * Bp, true);  // starting point of the current solve (options.resto_shift_retry)
  rc |= h->dalloc(&h->d_counts, 32), rc |= h->dalloc(&h->d_roll, 2 * ltompc_solver::ROLL_RING);
  rc |= h->dalloc(&h->d_plist, (size_t)ltompc_solver::ROLL_RING * h->Bp);
  if (getenv("LTOMPC_DBG")) rc |= h->dalloc(&W.DBG, 8 * N * Bp);
  rc |= h->dalloc(&h->d_x0_rm, 8 * Bp), rc |= h->dalloc(&h->d_u0_rm, 2 * Bp), rc |= h->dalloc(&h->d_io, 32 * Bp);
  if (rc) {
    ltompc_destroy(h);
    return -1;
  }
  if (hipHostMalloc((void**)&h->h_active, sizeof(int) * 4) != hipSuccess) {
    ltompc_destroy(h);
    return fail("hipHostMalloc failed");
  }
  W.orig = (gptr<const int>)h->d_orig;  // identity whenever the instances are not packed
  hipLaunchKernelGGL(k_act_identity, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->d_orig, h->d_perm, h->B);
  if (hipMemcpyAsync(h->d_tables, tables, sizeof(double) * LTOMPC_TABLE_ROWS * n_table, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess) {
    ltompc_destroy(h);
    return fail("table upload failed");
  }
  Tables& T = h->K.T;
  T.n = n_table;
  using tp = gptr<const double>;
  T.s_kappa = (tp)h->d_tables, T.kappa = (tp)(h->d_tables + n_table), T.s_arc = (tp)(h->d_tables + 2 * (size_t)n_table);
  T.n_left = (tp)(h->d_tables + 3 * (size_t)n_table), T.n_right = (tp)(h->d_tables + 4 * (size_t)n_table), T.v_ref = (tp)(h->d_tables + 5 * (size_t)n_table);
  T.g0_kappa = tables[0], T.inv_kappa = (double)(n_table - 1) / (tables[n_table - 1] - tables[0]);
  T.period = options->periodic_tables ? tables[n_table - 1] - tables[0] : 0.0;
  T.g0_arc = tables[2 * (size_t)n_table], T.inv_arc = (double)(n_table - 1) / (tables[3 * (size_t)n_table - 1] - tables[2 * (size_t)n_table]);
  // K is complete now: the solver kernels read it from device memory
  if (h->dalloc(&h->d_K, 1) || h->dalloc(&h->d_W, 1) ||
      hipMemcpyAsync(h->d_K, &h->K, sizeof(Consts), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipMemcpyAsync(h->d_W, &h->W, sizeof(Work), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess) {
    ltompc_destroy(h);
    return fail("ltompc_create: upload of the constants failed");
  }
  *out = h;
  return 0;
}

int ltompc_destroy(ltompc_handle h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->h_active) (void)hipHostFree(h->h_active);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  for (auto& ps : h->plant_streams)
    if (ps) (void)hipStreamDestroy(ps);
  for (int i = 0; i < ltompc_solver::ROLL_RING; i++) {
    if (h->ev_fin[i]) (void)hipEventDestroy(h->ev_fin[i]);
    if (h->ev_done[i]) (void)hipEventDestroy(h->ev_done[i]);
  }
  delete h;
  return 0;
}

int ltompc_set_stream(ltompc_handle h, void* hip_stream) {
  if (!h) return fail("null handle");
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return 0;
}

int ltompc_set_profiling(ltompc_handle h, int on) {
  if (!h) return fail("null handle");
  if (on < 0 || on >= 2 + NKERN) return fail("ltompc_set_profiling: mode out of range");
  h->profiling = on;
  for (int i = 0; i < NKERN; i++) h->ms_by_kernel[i] = 0, h->launches_by_kernel[i] = 0;
  h->log_kind.clear(), h->log_width.clear(), h->log_iter.clear(), h->log_ms.clear();
  return 0;
}

int ltompc_set_narrow_width(ltompc_handle h, int width) {
  if (!h || width < 0 || width > 512) return fail("ltompc_set_narrow_width: width must be in 0 .. 512");
  if (h->ric1_width > 0 || width == 0) h->ric1_width = width;  // (stays 0 when the horizon's stage blocks do not fit the LDS)
  if (h->step1_width > 0 || width == 0) h->step1_width = width;  // (LTOMPC_STEP1=0 / LTOMPC_RIC1=0, the test switches, stay off)
  return 0;
}

int ltompc_set_poll_every(ltompc_handle h, int n) {
  if (!h || n < 1) return fail("ltompc_set_poll_every: bad argument");
  h->poll_every = n;
  return 0;
}

int ltompc_set_initial_guess_dev(ltompc_handle h, const double* x0_dev) {
  if (!h || !x0_dev) return fail("ltompc_set_initial_guess: null argument");
  HIPCHECK(hipSetDevice(h->device));
  h->packed = false;  // a cold start overwrites the whole iterate: nothing to restore
  hipLaunchKernelGGL(k_act_identity, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->d_orig, h->d_perm, h->B);  // (slot -> caller's index: identity again)
  hipLaunchKernelGGL(k_load_x0, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->W, x0_dev, (const int*)nullptr, 0, 0);
  hipLaunchKernelGGL(k_zero_uprev, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->W);
  hipLaunchKernelGGL(k_init, dim3((h->N * h->Bp + 63) / 64), dim3(64), 0, h->stream, h->d_K, h->d_W, 1);
  HIPCHECK(hipGetLastError());
  h->cold_next = true;  // the next make_step starts from this guess
  return 0;
}

int ltompc_set_initial_guess(ltompc_handle h, const double* x0) {
  if (!h || !x0) return fail("ltompc_set_initial_guess: null argument");
  HIPCHECK(hipSetDevice(h->device));
  HIPCHECK(hipMemcpyAsync(h->d_x0_rm, x0, sizeof(double) * 8 * h->B, hipMemcpyHostToDevice, h->stream));
  int rc = ltompc_set_initial_guess_dev(h, h->d_x0_rm);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(h->stream));
  return 0;
}

int ltompc_make_step_dev(ltompc_handle h, const double* x0_dev, double* u0_dev) {
  if (!h || !x0_dev) return fail("ltompc_make_step: null argument");
  HIPCHECK(hipSetDevice(h->device));
  const int B = h->B, N = h->N, Bp = h->Bp;
  const bool ell = h->K.bd.nel > 0;  // kernels instantiated with / without the friction-ellipse constraints
  Launcher L{h};
  hipLaunchKernelGGL(k_load_x0, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->W, x0_dev, (const int*)(h->packed ? h->d_orig : nullptr),
                     h->K.o.resto_sticky, h->cold_next ? 0 : 1);
  if (h->cold_next) hipLaunchKernelGGL(k_zero_uprev, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->W);
  if (!h->cold_next && h->K.o.warm_shift) {
    hipLaunchKernelGGL(k_shift, dim3(((N + 1) * Bp + 63) / 64), dim3(64), 0, h->stream, h->W, 0);
    hipLaunchKernelGGL(k_shift, dim3(((N + 1) * Bp + 63) / 64), dim3(64), 0, h->stream, h->W, 1);
  }
  hipLaunchKernelGGL(k_init, dim3((N * Bp + 63) / 64), dim3(64), 0, h->stream, h->d_K, h->d_W, h->cold_next ? 1 : 0);
  HIPCHECK(hipMemsetAsync(h->W.active, 0, sizeof(int) * ((size_t)h->max_iter + 2), h->stream));
  HIPCHECK(hipMemsetAsync(h->W.ls_count, 0, 2 * sizeof(int), h->stream));
  h->cold_next = false, h->after_rollout = false;
  // all instances unfinished: identity list
  int cur = 0, n_launch = B;
  hipLaunchKernelGGL(k_act_identity, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->d_act[0], h->d_nact[0], B);
  Launch la{};
  auto set_launch = [&](int n) {
    n_launch = n;
    h->cur_width = n;
    la.act = (gptr<const int>)h->d_act[cur], la.nact = (gptr<const int>)h->d_nact[cur], la.n_pad = (n + 63) / 64 * 64;
  };
  set_launch(B);
  h->last_compactions = 0;
  h->history.clear();
  int it = 0;
  bool force_eval_next = false;
  int ls_width = h->ls_width_env ? h->ls_width_env : 512;  // until the first poll of this solve
  for (;; it++) {
    const int np = la.n_pad;
    h->cur_iter = it;
    la.force_eval = force_eval_next ? 1 : 0;
    force_eval_next = false;
    {
      const int rc = launch_iteration(h, L, la, it, n_launch, ls_width, ell, it >= h->max_iter);
      if (rc < 0) return -1;
    }
    if (it >= h->max_iter) break;  // that pass only finalised the statuses (MAX_ITER)
    if ((it + 1) % h->poll_every == 0) {
      if (L.close()) return -1;
      HIPCHECK(hipMemcpyAsync(h->h_active, h->W.active + it, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipMemcpyAsync(h->h_active + 1, h->W.ls_count + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipStreamSynchronize(h->stream));
      const int n_active = h->h_active[0];  // instances that passed the termination test of iteration `it`
      // Second line-search phase: one thread per (candidate, interval, rejected instance).  Its launch costs in
      // proportion to its width whether the threads find work or not (measured: 52 us at 192 instances, 130 us at 512,
      // 358 us at 2048, for ~250 rejected instances), and a list longer than the width costs whole extra passes: sized
      // by the length of the last list, with a margin.
      if (h->ls_width_env == 0) {
        const int last = h->h_active[1];
        ls_width = std::min(std::max(64, (last + last / 2 + 63) / 64 * 64), 2048);
      }
      h->history.push_back(it), h->history.push_back(n_active), h->history.push_back(n_launch);
      if (n_active == 0) break;
      if (h->compaction && n_active <= (h->pack_num * n_launch) / 8) {
        // finished instances only idle inside a launch, but they keep whole wavefronts alive: re-pack
        if (h->packing && n_launch > 512) {  // (narrow launches run one wavefront or workgroup per instance: nothing to coalesce)
          // ... the instances themselves (the first nact slots are permuted, unfinished ones first; k_pack in
          // aux_kernels.h), so that the lanes of a wavefront keep touching neighbouring addresses
          if (!h->packed) hipLaunchKernelGGL(k_act_identity, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->d_orig, h->d_perm, B);
          h->packed = true;
          hipLaunchKernelGGL(k_pack_perm, dim3(1), dim3(1024), 0, h->stream, h->d_nact[cur], h->W.si + (size_t)SI_DONE * Bp, h->d_perm,
                             h->d_act[cur ^ 1], h->d_nact[cur ^ 1]);
          const int nthreads = (N + 1) * n_launch;
          for (int pass = 0; pass < 2; pass++)
            hipLaunchKernelGGL(k_pack, dim3((nthreads + 255) / 256), dim3(256), 0, h->stream, h->W, h->d_perm, h->d_nact[cur], n_launch,
                               h->d_orig, h->K.bd.ni, h->K.bd.nel, pass);
          force_eval_next = true;  // the stage blocks of instances that would skip the evaluation stayed behind
        } else {
          hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, h->d_act[cur], h->d_nact[cur],
                             h->W.si + (size_t)SI_DONE * Bp, h->d_act[cur ^ 1], h->d_nact[cur ^ 1]);
        }
        cur ^= 1;
        set_launch(n_active);  // upper bound of the compacted count; kernels test against the device-side count
        h->last_compactions++;
      }
    }
  }
  if (L.close()) return -1;
  hipLaunchKernelGGL(k_store_u0, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->W, u0_dev, (const int*)(h->packed ? h->d_orig : nullptr));
  HIPCHECK(hipGetLastError());
  h->last_launches = L.launches + 3;
  h->last_iterations = it + 1;
  if (h->profiling) {
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (collect_profile(h)) return -1;
  }
  return 0;
}

int ltompc_synchronize(ltompc_handle h) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return 0;
}

int ltompc_get_stats(ltompc_handle h, int* status, int* iters, double* kkt_error, double* objective, double* mu) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  if (ensure_unpacked(h)) return -1;
  std::vector<int> si((size_t)SI_NF * h->Bp);
  std::vector<double> st((size_t)ST_NF * h->Bp);
  HIPCHECK(hipMemcpyAsync(si.data(), h->W.si, si.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(st.data(), h->W.st, st.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->B; b++) {
    if (status) status[b] = si[(size_t)SI_STATUS * h->Bp + b];
    if (iters) iters[b] = si[(size_t)SI_ITERS * h->Bp + b];
    if (kkt_error) kkt_error[b] = st[(size_t)ST_E0 * h->Bp + b];
    if (objective) objective[b] = st[(size_t)ST_OBJ * h->Bp + b];
    if (mu) mu[b] = st[(size_t)ST_MU * h->Bp + b];
  }
  return 0;
}

int ltompc_get_counters(ltompc_handle h, int* n_reg, int* n_lsfail) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  if (ensure_unpacked(h)) return -1;
  std::vector<int> si((size_t)SI_NF * h->Bp);
  HIPCHECK(hipMemcpyAsync(si.data(), h->W.si, si.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->B; b++) {
    if (n_reg) n_reg[b] = si[(size_t)SI_NREG * h->Bp + b];
    if (n_lsfail) n_lsfail[b] = si[(size_t)SI_NLSFAIL * h->Bp + b];
  }
  return 0;
}

int ltompc_get_restoration(ltompc_handle h, int* n_resto, double* violation) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  if (ensure_unpacked(h)) return -1;
  std::vector<int> si((size_t)SI_NF * h->Bp);
  std::vector<double> st((size_t)ST_NF * h->Bp);
  HIPCHECK(hipMemcpyAsync(si.data(), h->W.si, si.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(st.data(), h->W.st, st.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->B; b++) {
    if (n_resto) n_resto[b] = si[(size_t)SI_NRESTO * h->Bp + b];
    // (meaningful while the elastic variables exist: 0 once the solve is back on the hard constraints; g(x0) when the
    //  node-0 rule decided the status)
    const bool node0_inf = si[(size_t)SI_NODE0 * h->Bp + b] && si[(size_t)SI_STATUS * h->Bp + b] == LTOMPC_STATUS_INFEASIBLE;
    if (violation) violation[b] = (st[(size_t)ST_RHO * h->Bp + b] > 0.0 || node0_inf) ? st[(size_t)ST_VIOL * h->Bp + b] : 0.0;
  }
  return 0;
}

int ltompc_get_recovery(ltompc_handle h, int* n_shift, int* n_fallback, double* g0, int* solver_status, double* penalty) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  if (ensure_unpacked(h)) return -1;
  std::vector<int> si((size_t)SI_NF * h->Bp);
  std::vector<double> st((size_t)ST_NF * h->Bp);
  HIPCHECK(hipMemcpyAsync(si.data(), h->W.si, si.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(st.data(), h->W.st, st.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->B; b++) {
    if (n_shift) n_shift[b] = si[(size_t)SI_NSHIFT * h->Bp + b];
    if (n_fallback) n_fallback[b] = si[(size_t)SI_NFALLBACK * h->Bp + b];
    if (g0) g0[b] = st[(size_t)ST_G0 * h->Bp + b];
    const int node0 = si[(size_t)SI_NODE0 * h->Bp + b];
    if (solver_status) solver_status[b] = node0 ? node0 - 1 : si[(size_t)SI_STATUS * h->Bp + b];
    if (penalty) penalty[b] = st[(size_t)ST_RHO * h->Bp + b];
  }
  return 0;
}

// Closed-loop rollout with free-running instances (rollout.h): n_ticks of [make_step -> plant step] per instance.
int ltompc_rollout_dev(ltompc_handle h, double* x_dev, int n_ticks, int n_sub, double* u_log_dev, int* status_log_dev, int* iters_log_dev) {
  if (!h || !x_dev) return fail("ltompc_rollout: null argument");
  if (n_ticks < 1 || n_sub < 1) return fail("ltompc_rollout: n_ticks and n_sub must be >= 1");
  if (h->K.o.warm_shift) return fail("ltompc_rollout: options.warm_shift is not supported by the rollout");
  if (h->eval8) return fail("ltompc_rollout: latency-mode handles (8-lanes-per-slot kernels) are not supported by the rollout");
  HIPCHECK(hipSetDevice(h->device));
  if (ensure_unpacked(h)) return -1;  // the rollout works in the caller's order (index-list compaction only)
  constexpr int RING = ltompc_solver::ROLL_RING;
  if (!h->plant_streams[0]) {
    // low priority: the plant steps are not urgent, and the runtime keeps a pool of hardware queues per priority level, so
    // a plant stream never shares its hardware queue with the solver stream (sharing one serialises the 2 ms plant kernels
    // with the solver's: measured 1250 - 1870 ms instead of 1150 ms for 20 ticks, depending on what other streams the process has)
    int prio_low = 0, prio_high = 0;
    HIPCHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    for (int i = 0; i < h->n_plant_streams; i++) HIPCHECK(hipStreamCreateWithPriority(&h->plant_streams[i], hipStreamNonBlocking, prio_low));
    for (int i = 0; i < RING; i++) {
      HIPCHECK(hipEventCreateWithFlags(&h->ev_fin[i], hipEventDisableTiming));
      HIPCHECK(hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming));
    }
  }
  bool done_pending[RING] = {};
  const int B = h->B, N = h->N, Bp = h->Bp;
  const bool ell = h->K.bd.nel > 0;
  const int prof = h->profiling;
  h->profiling = 0;  // (per-launch events are a make_step facility)
  Launcher L{h};
  hipLaunchKernelGGL(k_roll_begin, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->W, n_ticks);
  if (h->cold_next) hipLaunchKernelGGL(k_zero_uprev, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->W);
  HIPCHECK(hipMemsetAsync(h->W.ls_count, 0, 2 * sizeof(int), h->stream));
  int cur = 0, n_launch = B;
  h->history.clear();
  hipLaunchKernelGGL(k_act_identity, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->d_act[0], h->d_nact[0], B);
  Launch la{};
  auto set_launch = [&](int n) {
    n_launch = n;
    la.act = (gptr<const int>)h->d_act[cur], la.nact = (gptr<const int>)h->d_nact[cur], la.n_pad = (n + 63) / 64 * 64;
  };
  set_launch(B);
  int ls_width = h->ls_width_env ? h->ls_width_env : 512;
  const long long it_cap = (long long)n_ticks * ((long long)h->max_iter + 8) + 64;  // every instance finishes every solve within max_iter + 1 passes
  const int ring = h->max_iter + 2;
  long long it = 0;
  bool first = true;
  int rc = 0;
  // (inside the loop a HIP error leaves through ONE exit: the plant streams and the solver stream are drained and the handle's
  //  profiling mode restored before the call returns - plant kernels may still be writing x_dev and the lists otherwise)
#define ROLLCHECK(expr)                                                                          \
  {                                                                                              \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) { rc = fail(std::string(#expr) + ": " + hipGetErrorString(e_)); break; } \
  }
  for (;; it++) {
    const int np = la.n_pad, slot = (int)(it % RING);
    int* const d_cnt = h->d_roll + 2 * slot;
    int* const d_list = h->d_plist + (size_t)slot * Bp;
    if (done_pending[slot]) ROLLCHECK(hipStreamWaitEvent(h->stream, h->ev_done[slot], 0));  // (the plant kernel that read this slot's list, RING passes ago)
    ROLLCHECK(hipMemsetAsync(d_cnt, 0, 2 * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_roll_mark, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->W, (const double*)x_dev, h->K.o.resto_sticky,
                       (first && h->cold_next) ? 1 : 0);  // (after set_initial_guess there is no solve before this one to take stock of)
    hipLaunchKernelGGL(k_roll_init, dim3((N * np + 63) / 64), dim3(64), 0, h->stream, h->d_K, h->d_W, la, (first && h->cold_next) ? 1 : 0);
    la.force_eval = 0;
    if (launch_iteration(h, L, la, -1, n_launch, ls_width, ell, false) < 0) { rc = -1; break; }  // (-1: no per-iteration count of unfinished instances, a make_step facility)
    hipLaunchKernelGGL(k_roll_finish, dim3((np + 63) / 64), dim3(64), 0, h->stream, h->W, la, u_log_dev, status_log_dev, iters_log_dev, n_ticks,
                       d_cnt, d_list);
    // The plant steps of the instances that have just converged (the list k_roll_finish made: a plant step is ~1 ms of one
    // lane's work, so the wavefronts are dense) on another stream, after this pass's k_roll_finish.  Measured and left at that:
    // the solver kernels are 10 - 15 % slower at full width while plant wavefronts are resident, and k_riccati8 (exactly one
    // wavefront per SIMD at 8192 instances) 0.29 -> 0.45 ms; holding the plant kernel back until the Riccati kernel has run and
    // making the next one wait for it restores k_riccati8, but beside the big kernels a plant step takes 1.5 ms, longer than a
    // pass, and the solver stream then waits 0.5 ms per pass (scratch/rtrace.sh).
    {
      hipStream_t ps = h->plant_streams[it % h->n_plant_streams];
      ROLLCHECK(hipEventRecord(h->ev_fin[slot], h->stream));
      ROLLCHECK(hipStreamWaitEvent(ps, h->ev_fin[slot], 0));
      hipLaunchKernelGGL(k_roll_plant, dim3((n_launch + 63) / 64), dim3(64), 0, ps, h->K, h->W, x_dev, h->K.o.t_step, n_sub, (const int*)(d_cnt + 1),
                         (const int*)d_list);
      ROLLCHECK(hipEventRecord(h->ev_done[slot], ps));
      done_pending[slot] = true;
    }
    first = false;
    if ((it + 1) % h->poll_every == 0) {
      ROLLCHECK(hipMemcpyAsync(h->h_active, d_cnt, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      ROLLCHECK(hipMemcpyAsync(h->h_active + 1, h->W.ls_count + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      ROLLCHECK(hipStreamSynchronize(h->stream));
      const int n_left = h->h_active[0];  // instances that still have ticks to do (counted before this iteration's plant steps)
      h->history.push_back((int)it), h->history.push_back(n_left), h->history.push_back(n_launch);
      if (h->ls_width_env == 0) {
        const int last = h->h_active[1];
        ls_width = std::min(std::max(64, (last + last / 2 + 63) / 64 * 64), 2048);
      }
      if (n_left == 0) break;
      if (it >= it_cap) { rc = fail("ltompc_rollout: iteration cap reached (internal error)"); break; }
      if (h->compaction && n_left <= (h->pack_num * n_launch) / 8) {
        // instances that have done all their ticks leave the launches (index list only: their data stays where it is)
        hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, h->d_act[cur], h->d_nact[cur], h->W.si + (size_t)SI_FINAL * Bp,
                           h->d_act[cur ^ 1], h->d_nact[cur ^ 1]);
        cur ^= 1;
        // (an instance can turn FINAL on the plant stream after this count: the list may keep it one round longer, harmless)
        set_launch(n_left);
      }
    }
  }
#undef ROLLCHECK
  // the one exit: drain every stream the rollout used (also after an error), restore the handle's state
  for (int i = 0; i < h->n_plant_streams; i++)
    if (hipStreamSynchronize(h->plant_streams[i]) != hipSuccess && rc == 0) rc = fail("ltompc_rollout: plant stream failed");
  if (hipStreamSynchronize(h->stream) != hipSuccess && rc == 0) rc = fail("ltompc_rollout: solver stream failed");
  if (hipGetLastError() != hipSuccess && rc == 0) rc = fail("ltompc_rollout: a kernel launch failed");
  h->profiling = prof;
  h->cold_next = false;
  h->roll_iterations = it + 1, h->roll_launches = L.launches;
  h->last_iterations = 0;  // (no per-iteration history after a rollout: ltompc_get_active_history returns 0)
  h->after_rollout = true;
  return rc;
}

int ltompc_rollout_info(ltompc_handle h, long long* iterations, long long* launches) {
  if (!h) return fail("null handle");
  if (iterations) *iterations = h->roll_iterations;
  if (launches) *launches = h->roll_launches;
  return 0;
}

int ltompc_make_step(ltompc_handle h, const double* x0, double* u0, int* status, int* iters) {
  if (!h || !x0 || !u0) return fail("ltompc_make_step: null argument");
  HIPCHECK(hipSetDevice(h->device));
  for (size_t i = 0; i < (size_t)8 * h->B; i++)
    if (!std::isfinite(x0[i])) return fail("ltompc_make_step: non-finite x0");
  HIPCHECK(hipMemcpyAsync(h->d_x0_rm, x0, sizeof(double) * 8 * h->B, hipMemcpyHostToDevice, h->stream));
  int rc = ltompc_make_step_dev(h, h->d_x0_rm, h->d_u0_rm);
  if (rc) return rc;
  HIPCHECK(hipMemcpyAsync(u0, h->d_u0_rm, sizeof(double) * 2 * h->B, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  if (status || iters) return ltompc_get_stats(h, status, iters, nullptr, nullptr, nullptr);
  return 0;
}

int ltompc_get_prediction(ltompc_handle h, double* X, double* U) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  if (ensure_unpacked(h)) return -1;
  if (planes_to_host(h, h->W.X, 8, h->N + 1, X)) return -1;
  return planes_to_host(h, h->W.U, 2, h->N, U);
}

int ltompc_get_iterate(ltompc_handle h, double* X, double* C, double* U, double* L1, double* L2) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  if (ensure_unpacked(h)) return -1;
  if (planes_to_host(h, h->W.X, 8, h->N + 1, X)) return -1;
  if (planes_to_host(h, h->W.C, 8, h->N, C)) return -1;
  if (planes_to_host(h, h->W.U, 2, h->N, U)) return -1;
  if (planes_to_host(h, h->W.L1, 8, h->N, L1)) return -1;
  return planes_to_host(h, h->W.L2, 8, h->N, L2);
}

int ltompc_get_ineq(ltompc_handle h, double* T, double* NU, int* n_ineq) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  if (ensure_unpacked(h)) return -1;
  if (n_ineq) *n_ineq = h->K.bd.ni;
  if (planes_to_host(h, h->W.T, h->K.bd.ni, h->N, T)) return -1;
  return planes_to_host(h, h->W.NU, h->K.bd.ni, h->N, NU);
}

int ltompc_plant_step_dev(ltompc_handle h, const double* x_dev, const double* u_dev, int n_sub, double* x_next_dev) {
  if (!h || !x_dev || !u_dev || !x_next_dev) return fail("ltompc_plant_step: null argument");
  if (n_sub < 1) return fail("ltompc_plant_step: n_sub must be >= 1");
  HIPCHECK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_plant, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->K, h->B, x_dev, u_dev, h->K.o.t_step, n_sub, x_next_dev);
  HIPCHECK(hipGetLastError());
  return 0;
}

int ltompc_plant_step(ltompc_handle h, const double* x, const double* u, int n_sub, double* x_next) {
  if (!h || !x || !u || !x_next) return fail("ltompc_plant_step: null argument");
  HIPCHECK(hipSetDevice(h->device));
  double *dx = h->d_io, *du = h->d_io + 8 * (size_t)h->Bp, *dn = h->d_io + 10 * (size_t)h->Bp;
  HIPCHECK(hipMemcpyAsync(dx, x, sizeof(double) * 8 * h->B, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(du, u, sizeof(double) * 2 * h->B, hipMemcpyHostToDevice, h->stream));
  int rc = ltompc_plant_step_dev(h, dx, du, n_sub, dn);
  if (rc) return rc;
  HIPCHECK(hipMemcpyAsync(x_next, dn, sizeof(double) * 8 * h->B, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return 0;
}

int ltompc_slip_forces(ltompc_handle h, const double* x, int batch, double* alpha, double* Fy) {
  if (!h || !x || !alpha || !Fy) return fail("ltompc_slip_forces: null argument");
  if (batch < 1 || batch > h->B) return fail("ltompc_slip_forces: batch must be in [1, handle batch]");
  HIPCHECK(hipSetDevice(h->device));
  double *dx = h->d_io, *da = h->d_io + 8 * (size_t)h->Bp, *df = h->d_io + 10 * (size_t)h->Bp;
  HIPCHECK(hipMemcpyAsync(dx, x, sizeof(double) * 8 * batch, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_slip_forces, dim3((batch + 63) / 64), dim3(64), 0, h->stream, h->K, batch, dx, da, df);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(alpha, da, sizeof(double) * 2 * batch, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(Fy, df, sizeof(double) * 2 * batch, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return 0;
}

int ltompc_get_timing(ltompc_handle h, double* ms_by_kernel8, int* launches_by_kernel8, int* launches, int* ip_iterations) {
  if (!h) return fail("null handle");
  for (int i = 0; i < NKERN; i++) {
    if (ms_by_kernel8) ms_by_kernel8[i] = h->ms_by_kernel[i];
    if (launches_by_kernel8) launches_by_kernel8[i] = h->launches_by_kernel[i];
  }
  if (launches) *launches = h->last_launches;
  if (ip_iterations) *ip_iterations = h->last_iterations;
  return 0;
}

int ltompc_get_launch_log(ltompc_handle h, int* kind, int* width, double* ms, int capacity) {
  if (!h) return fail("null handle");
  const int n = (int)h->log_kind.size();
  for (int i = 0; i < n && i < capacity; i++) {
    if (kind) kind[i] = h->log_kind[i];
    if (width) width[i] = h->log_width[i];
    if (ms) ms[i] = h->log_ms[i];
  }
  return n;
}

int ltompc_get_launch_log_iterations(ltompc_handle h, int* iteration, int capacity) {
  if (!h) return fail("null handle");
  const int n = (int)h->log_iter.size();
  for (int i = 0; i < n && i < capacity; i++)
    if (iteration) iteration[i] = h->log_iter[i];
  return n;
}

int ltompc_get_active_history(ltompc_handle h, int* active, int capacity) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  const int n = h->after_rollout ? 0 : std::min(h->last_iterations, h->max_iter + 2);
  if (active && capacity > 0 && n > 0) {
    HIPCHECK(hipMemcpyAsync(active, h->W.active, sizeof(int) * std::min(n, capacity), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  }
  return n;
}

int ltompc_get_status_counts(ltompc_handle h, int* counts8, long long* iterations_sum) {
  if (!h) return fail("null handle");
  HIPCHECK(hipSetDevice(h->device));
  HIPCHECK(hipMemsetAsync(h->d_counts, 0, sizeof(int) * 32, h->stream));
  hipLaunchKernelGGL(k_status_counts, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, h->W, h->d_counts,
                     reinterpret_cast<unsigned long long*>(h->d_counts + 8));
  HIPCHECK(hipGetLastError());
  int host[32];
  HIPCHECK(hipMemcpyAsync(host, h->d_counts, sizeof host, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 8; i++) {
    if (counts8) counts8[i] = host[i];
    h->solver_counts[i] = host[16 + i];
  }
  if (iterations_sum) std::memcpy(iterations_sum, host + 8, sizeof(long long));
  return 0;
}

int ltompc_get_solver_status_counts(ltompc_handle h, int* counts8) {
  if (!h || !counts8) return fail("ltompc_get_solver_status_counts: null argument");
  for (int i = 0; i < 8; i++) counts8[i] = h->solver_counts[i];
  return 0;
}

int ltompc_get_history(ltompc_handle h, int* triples, int capacity) {
  if (!h) return fail("null handle");
  int n = (int)h->history.size() / 3;
  for (int i = 0; i < n && i < capacity; i++)
    for (int j = 0; j < 3; j++) triples[3 * i + j] = h->history[3 * i + j];
  return n;
}

// Debug hook: raw copy of a device work array (layout as on the device).  which: 0 QP, 1 RC, 2 RS, 3 SP, 4 LS, 5 dX, 6 dU,
// 7 dC, 8 dT, 9 dNU, 10 nL1, 11 nL2, 12 st, 13 si (ints).  Returns the number of bytes of the array (copies min(nbytes, size)).
long long ltompc_debug_fetch(ltompc_handle h, int which, void* out, long long nbytes) {
  if (!h) return fail("null handle");
  if (hipSetDevice(h->device) != hipSuccess || ensure_unpacked(h)) return fail("ltompc_debug_fetch: could not restore the caller's order");
  const size_t N = h->N, Bp = h->Bp, ni = h->K.bd.ni;
  const Work& W = h->W;
  const void* src[15] = {W.QP, W.RC, W.RS, W.SP, W.LS, W.dX, W.dU, W.dC, W.dT, W.dNU, W.nL1, W.nL2, W.st, W.si, W.DBG};
  const size_t sz[15] = {QP_NF * (N + 1) * Bp * 8, RC_NF * (N + 1) * Bp * 8, RS_NF * N * Bp * 8, SP_NF * N * Bp * 8,
                         3 * ((size_t)h->K.o.n_linesearch + 1) * N * Bp * 8, 8 * (N + 1) * Bp * 8, 2 * N * Bp * 8, 8 * N * Bp * 8,
                         ni * N * Bp * 8, ni * N * Bp * 8, 8 * N * Bp * 8, 8 * N * Bp * 8, (size_t)ST_NF * Bp * 8, (size_t)SI_NF * Bp * 4,
                         W.DBG ? 8 * N * Bp * 8 : 0};
  if (which < 0 || which > 14) return fail("ltompc_debug_fetch: bad array id");
  if (out && nbytes > 0) {
    size_t n = std::min((size_t)nbytes, sz[which]);
    if (hipSetDevice(h->device) != hipSuccess || hipMemcpyAsync(out, src[which], n, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
      return fail("ltompc_debug_fetch: copy failed");
  }
  return (long long)sz[which];
}

// Test hook: model derivatives at n points (host arrays): x, lam: n x 8 -> f: n x 8, J, H: n x 64 (row-major 8x8),
// cost value/grad/Hessian for lterm and mterm (n x 2 [x 8 [x 8]]), constraints gL, gR+, gR- (n x 3 [x 8 [x 8]]).
int ltompc_test_model(ltompc_handle h, int n, double eps, const double* x, const double* lam, double* f, double* J,
                      double* H, double* cval, double* cgrad, double* cH, double* gval, double* ggrad, double* gH) {
  if (!h || n < 1) return fail("ltompc_test_model: bad argument");
  HIPCHECK(hipSetDevice(h->device));
  const size_t sizes[12] = {8, 8, 8, 64, 64, 2, 16, 128, 3, 24, 192, 0};
  double* dev[11];
  const double* src[2] = {x, lam};
  double* dst[9] = {f, J, H, cval, cgrad, cH, gval, ggrad, gH};
  for (int i = 0; i < 11; i++) HIPCHECK(hipMalloc((void**)&dev[i], sizeof(double) * sizes[i] * n));
  for (int i = 0; i < 2; i++) HIPCHECK(hipMemcpy(dev[i], src[i], sizeof(double) * sizes[i] * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_test_model, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->K, n, eps, dev[0], dev[1], dev[2], dev[3],
                     dev[4], dev[5], dev[6], dev[7], dev[8], dev[9], dev[10]);
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 9; i++) HIPCHECK(hipMemcpy(dst[i], dev[2 + i], sizeof(double) * sizes[2 + i] * n, hipMemcpyDeviceToHost));
  for (int i = 0; i < 11; i++) (void)hipFree(dev[i]);
  return 0;
}

// Test hook: friction-ellipse constraints (front, rear) at n points (host arrays): val n x 2, grad n x 2 x 8, H n x 2 x 64.
int ltompc_test_ellipse(ltompc_handle h, int n, const double* x, double* val, double* grad, double* H) {
  if (!h || n < 1 || !x || !val || !grad || !H) return fail("ltompc_test_ellipse: bad argument");
  HIPCHECK(hipSetDevice(h->device));
  double *dx, *dv, *dg, *dH;
  HIPCHECK(hipMalloc((void**)&dx, sizeof(double) * 8 * n));
  HIPCHECK(hipMalloc((void**)&dv, sizeof(double) * 2 * n));
  HIPCHECK(hipMalloc((void**)&dg, sizeof(double) * 16 * n));
  HIPCHECK(hipMalloc((void**)&dH, sizeof(double) * 128 * n));
  HIPCHECK(hipMemcpy(dx, x, sizeof(double) * 8 * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_test_ellipse, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->K, n, dx, dv, dg, dH);
  HIPCHECK(hipStreamSynchronize(h->stream));
  HIPCHECK(hipMemcpy(val, dv, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(grad, dg, sizeof(double) * 16 * n, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(H, dH, sizeof(double) * 128 * n, hipMemcpyDeviceToHost));
  (void)hipFree(dx), (void)hipFree(dv), (void)hipFree(dg), (void)hipFree(dH);
  return 0;
}

int ltompc_velocity_profile(int device, const ltompc_vp_vehicle* veh, int n, int batch, const double* s, const double* k,
                            const double* s_max, double* v, double* v_local, double* v_acclim, double* v_declim) {
  if (!veh || !s || !k || !s_max || !v) return fail("ltompc_velocity_profile: null argument");
  if (n < 2 || batch < 1) return fail("ltompc_velocity_profile: need n >= 2 samples and batch >= 1");
  if (veh->kind != 0 && veh->kind != 1) return fail("ltompc_velocity_profile: vehicle kind must be 0 (engine map) or 1 (MX-5)");
  if (veh->kind == 0 && (veh->n_map < 2 || veh->n_map > 16)) return fail("ltompc_velocity_profile: engine map needs 2 .. 16 points");
  if (!(veh->mass > 0)) return fail("ltompc_velocity_profile: mass must be positive");
  for (size_t i = 0; i < (size_t)n * batch; i++)
    if (!(k[i] >= 0.0) || !std::isfinite(s[i])) return fail("ltompc_velocity_profile: curvature must be >= 0 and s finite");
  int ndev = 0;
  HIPCHECK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("ltompc_velocity_profile: no such HIP device");
  HIPCHECK(hipSetDevice(device));
  const size_t nb = sizeof(double) * (size_t)n * batch;
  double *d_s = nullptr, *d_k = nullptr, *d_m = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc((void**)&d_s, nb);
  if (e == hipSuccess) e = hipMalloc((void**)&d_k, nb);
  if (e == hipSuccess) e = hipMalloc((void**)&d_m, sizeof(double) * batch);
  if (e == hipSuccess) e = hipMalloc((void**)&d_out, 4 * nb);
  if (e == hipSuccess) e = hipMemcpy(d_s, s, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_k, k, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_m, s_max, sizeof(double) * batch, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const size_t nn = (size_t)n * batch;
    hipLaunchKernelGGL(k_velocity_profile, dim3((batch + 63) / 64), dim3(64), 0, 0, *veh, n, batch, d_s, d_k, d_m, d_out, d_out + nn, d_out + 2 * nn, d_out + 3 * nn);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(v, d_out, nb, hipMemcpyDeviceToHost);
    if (e == hipSuccess && v_local) e = hipMemcpy(v_local, d_out + nn, nb, hipMemcpyDeviceToHost);
    if (e == hipSuccess && v_acclim) e = hipMemcpy(v_acclim, d_out + 2 * nn, nb, hipMemcpyDeviceToHost);
    if (e == hipSuccess && v_declim) e = hipMemcpy(v_declim, d_out + 3 * nn, nb, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_s), (void)hipFree(d_k), (void)hipFree(d_m), (void)hipFree(d_out);
  if (e != hipSuccess) return fail(std::string("ltompc_velocity_profile: ") + hipGetErrorString(e));
  return 0;
}

}  // extern "C"
