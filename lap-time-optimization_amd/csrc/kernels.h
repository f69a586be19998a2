// kernels.h — HIP kernels of the batched interior-point NLP solver (gfx950).
//
// One make_step (reference src/mpc.py:142, do_mpc MPC.make_step -> IPOPT) = up to max_iter interior-point
// iterations, each a fixed sequence of kernels over the instances that have not finished:
//
//   layout.h      buffers, field indices, Work / Launch / Consts, wave-level helpers
//   linearise.h   (kernels instantiated for the reference's bound pattern, BoundsRef, and for a run-time pattern)
//                 k_eval    thread = (interval k, instance b): derivatives of dynamics / cost / constraints at the
//                           Radau point and the next node, KKT-residual partials, block-structured elimination of the
//                           collocation variables -> stage QP blocks (A, B, b, Q, S, R, q, r)
//                 k_expand  thread = (k, b): collocation steps and multipliers, slack / inequality-multiplier steps,
//                           fraction-to-the-boundary partials
//   eval8.h       k_eval8 / k_expand8: the same with 8 lanes per (k, b) (latency mode)
//   riccati.h     k_riccati8 (8 instances per wavefront) / k_riccati1 (one instance per wavefront) / k_riccati1q (one instance
//                 on four wavefronts, launches of at most 16 instances) / k_riccati (one thread per instance, reference implementation): KKT error, termination, barrier update, Riccati
//                 backward sweep with inertia-correcting regularisation, forward rollout
//   linesearch.h  k_linesearch (filter measures of the step candidates), k_pick (filter test, step length, stall
//                 bookkeeping), k_update (z += alpha dz), k_step1 (the three fused, one workgroup per instance)
//   aux_kernels.h k_init, k_shift, re-packing of the unfinished instances (k_pack_perm / k_pack move their data to the
//                 front of the batch, k_compact re-packs the index list only), I/O, k_plant (RK4 plant step), test hooks
//   rollout.h     k_roll_*: closed-loop rollout with free-running instances (ltompc_rollout_dev)
//   velocity.h    k_velocity_profile: the forward / backward speed-profile passes of src/velocity.py
#pragma once
#include "layout.h"
#include "linearise.h"
#include "riccati.h"
#include "linesearch.h"
#include "aux_kernels.h"
#include "eval8.h"
#include "velocity.h"
#include "rollout.h"
