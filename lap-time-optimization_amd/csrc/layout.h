// layout.h — buffers, indices and shared helpers of the HIP kernels (see kernels.h for the overview).
//
// HBM layout: every per-(k,b) quantity of the iterate / step / partial arrays is a plane [field][k][Bp] with the
// instance index fastest, so that the 64 lanes of a wavefront (consecutive b, same k) read/write 512 contiguous bytes
// per field; the stage-QP and Riccati buffers are [k][b / 8][field][b % 8] (see PG below).
#pragma once
#include "model.h"

namespace ltompc {


constexpr int FILTER_MAX = 16;
constexpr double DW_KEEP = 1e-5;  // regularisation below this is dropped to exactly 0
constexpr int MAX_LS = 12;
constexpr double ELASTIC_CP_VIOL = 0.1;  // [m] violation of a track constraint above which its elastic variable starts on the central path
// Penalty scale S of an instance (include/ltompc.h, resto_rho_max): with elastic variables that cost rho > RHO_UNIT the solve runs
// in the units of  objective / S + RHO_UNIT * violation,  S = rho / RHO_UNIT - tolerances, barrier parameter, regularisation and the
// objective side of the filter are S times their options (at rho = 1e7 the unscaled tolerance 1e-8 on gradients of size 1e7 is
// below the rounding floor; scaled, the solver sees the problem it handles at rho = 1000).  S = 1 otherwise, and x * 1.0 is exact:
// nothing changes for the other instances.
constexpr double RHO_UNIT = 1000.0;
__host__ __device__ __forceinline__ double pen_scale(const double rho) { return rho > RHO_UNIT ? rho / RHO_UNIT : 1.0; }

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
enum : int { RS_rd = 0, RS_rp, RS_cmax, RS_cmin, RS_smult, RS_cost, RS_emax, RS_NF };  // RS_emax: largest elastic variable
// step partials written by k_expand
enum : int { SP_apri = 0, SP_adua, SP_gphid, SP_NF };
// per-instance double state
enum : int {
  ST_MU = 0, ST_EPS, ST_EPS_NEXT, ST_DW_LAST, ST_FORCE_REG, ST_ALPHA, ST_ADUA, ST_E0, ST_OBJ, ST_TAU,
  ST_THETA0, ST_THMAX, ST_THMIN, ST_DW, ST_DW_TRY,
  ST_C00,  // lterm(x_0) for the current ST_EPS: a constant of the solve between two changes of the table smoothing
  ST_RHO,  // penalty of the elastic variables of the track constraints: options.soft_rho, or options.resto_rho while
           // the instance is in its restoration phase (0: hard constraints)
  ST_VIOL, // largest elastic variable seen by the last termination test (g(x0) when the node-0 rule decided the status)
  ST_G0,   // largest track constraint at the measured state x0 (options.node0_check), k_init
  ST_NF
};
// per-instance int state
// SI_LSMORE: the full step was rejected by the filter test, the remaining step candidates have to be evaluated.
// SI_RETRY: the last Riccati sweep failed the inertia test; the next launch repeats it with ST_DW_TRY (no new
// evaluation).  SI_SKIP_EVAL: the iterate did not move (failed line search), k_eval's output is still valid.
// SI_RESTO: restoration phase (elastic mode): 0 not entered, 1 solving the elastic problem, 2 back on the
// hard constraints.  SI_REINIT: set when the phase is entered; the next evaluation kernel first re-initialises the
// slacks / multipliers of its slot from the primal point (the Riccati head clears the flag).
enum : int { SI_STATUS = 0, SI_ITERS, SI_NACC, SI_NTINY, SI_NFILT, SI_DONE, SI_STEP, SI_NREG, SI_NLSFAIL, SI_RETRY, SI_TRIES,
             SI_SKIP_EVAL, SI_LSMORE, SI_PREV, SI_RESTO, SI_REINIT, SI_NRESTO,
             SI_STICKY,   // option resto_sticky: solves for which the instance still starts in elastic mode (kept between make_steps)
             SI_STARTEL,  // this solve started in elastic mode
             SI_SWEEPS,   // passes this solve has used: one per Riccati head, plus one per sweep repeated inside a launch - what a
                          // launch of ONE sweep per pass would have needed, so that the iteration budget (options.max_iter counts
                          // passes) cuts a solve at the same point whatever the launch widths were
             // closed-loop rollout (rollout.h): where the instance is in its tick cycle, ticks it still has to do, finished for good
             SI_PHASE, SI_TICKS, SI_FINAL,
             SI_WARM,     // this solve is warm-started (k_init)
             SI_SHIFT,    // options.resto_shift_retry: d_pick has sent the solve back to its own starting point moved one interval ahead;
                          // d_update (the next slot-parallel kernel) loads it from the backup planes, the next head clears the flag
             SI_NSHIFT,   // times that happened in this solve (0 or 1)
             SI_SINCEMU,  // options.warm_fallback_iter: iterations since the barrier parameter last decreased
             SI_FBARMED,  // ... the fallback is still available to this solve (it started at mu_init_warm)
             SI_NFALLBACK,
             SI_BLOWUP,   // options.dual_inf_max: the head found the dual infeasibility beyond the limit; d_pick of this iteration sends
                          // the solve to its recovery steps instead of taking the step
             SI_NODE0,    // options.node0_check turned the solver's status (value - 1: SOLVED / ACCEPTABLE) into INFEASIBLE: the next
                          // solve is warm-started as after a converged one
             SI_NF };  // SI_PREV: status of the previous make_step (k_load_x0)

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
  // options.resto_shift_retry: the primal starting point of the current solve, [18][N][Bp]: x_{k+1} (8), c_k (8), u_k (2) of slot k,
  // indexed by the CALLER's instance index orig[b] (re-packing moves instances between slots, not this array)
  gptr<double> BK;
  gptr<const int> orig;  // slot -> caller's index (identity while the instances are not packed)
};

// What changes from launch to launch (kernel argument; Work and Consts are read from device memory).  Compaction of the
// unfinished instances: thread j of a launch works on instance act[j], j < nact[0] <= the launch width.  The list is
// sorted (stable compaction), so while nothing has finished it is the identity and accesses coalesce.
struct Launch {
  gptr<const int> act;
  gptr<const int> nact;
  int n_pad;  // launch width rounded up to a multiple of 64
  int force_eval;  // the instances have just been moved (k_pack): the stage blocks of the last launch are not where they were
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


// per-instance scalars: `st` / `si` are local copies of W.st / W.si in the functions that use these
#define STD(f) st[(size_t)(f) * W.Bp + b]
#define STI(f) si[(size_t)(f) * W.Bp + b]

// no s_barrier and, unlike __syncthreads(), must not drain the outstanding global loads/stores (vmcnt): only the
// compiler has to keep the LDS accesses in order.
#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

#if !defined(LTOMPC_HOST_HARNESS)  // (the harness brings its own: 8 OS threads and a barrier, hip_shim.h)
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

#else
using ::grp_max;
using ::grp_min;
using ::grp_sum;
#endif

}  // namespace ltompc
