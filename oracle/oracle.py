"""ctypes binding of the CPU oracle (oracle/ltompc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package never does (it fails loudly when the HIP library is missing instead of falling back to this).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libltompc_oracle.so")
NX, NU = 8, 2


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "mass", "inertia_z", "length_f", "length_r", "width", "B_f", "C_f", "D_f", "B_r", "C_r", "D_r",
        "C_m", "Cr_0", "Cr_2", "gravity", "ptv", "q_n", "q_mu", "q_vy", "q_v", "vref_scale", "q_B")] + [
        ("r_du", C.c_double * 2), ("x_lb", C.c_double * 8), ("x_ub", C.c_double * 8),
        ("u_lb", C.c_double * 2), ("u_ub", C.c_double * 2)] + [
        (n, C.c_double) for n in ("ell_penalty", "ell_rho", "ell_D_f", "ell_D_r")]


class Options(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "t_step", "tol", "acceptable_tol", "mu_init", "mu_min", "kappa_eps", "kappa_mu", "theta_mu", "tau_min",
        "bound_push", "s_max", "delta_w_first", "smooth_eps_min", "smooth_scale", "mu_init_warm", "soft_rho", "resto_rho",
        "resto_rho_max", "resto_rho_factor", "dual_inf_max")] + [
        ("max_iter", C.c_int), ("acceptable_iter", C.c_int), ("n_linesearch", C.c_int), ("stall_iter", C.c_int),
        ("max_ls_fail", C.c_int), ("warm_shift", C.c_int), ("warm_reset_on_fail", C.c_int), ("periodic_tables", C.c_int), ("max_soc", C.c_int), ("resto_sticky", C.c_int),
        ("node0_check", C.c_int), ("warm_fallback_iter", C.c_int), ("resto_shift_retry", C.c_int), ("max_mu_stay", C.c_int), ("infeasible_sticky", C.c_int), ("latency_mode", C.c_int)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "ltompc_oracle.c"), os.path.join(_HERE, "..", "include", "ltompc.h")]  # (the structs)
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def default_params() -> Params:
    p = Params()
    lib().oracle_default_params(C.byref(p))
    return p


def default_options() -> Options:
    o = Options()
    lib().oracle_default_options(C.byref(o))
    return o


class Oracle:
    """CPU restatement of the make_step path over one set of tables."""

    def __init__(self, tables_packed: np.ndarray, params: Params | None = None, options: Options | None = None):
        self.tab = np.ascontiguousarray(tables_packed, dtype=np.float64)
        assert self.tab.ndim == 2 and self.tab.shape[0] == 6
        self.nt = self.tab.shape[1]
        self.p = params or default_params()
        self.o = options or default_options()

    # ---- model pieces -------------------------------------------------------------
    def rhs(self, x, u):
        x = np.ascontiguousarray(x, float); u = np.ascontiguousarray(u, float)
        f = np.zeros(8)
        lib().oracle_rhs(C.byref(self.p), _p(self.tab), self.nt, _p(x), _p(u), _p(f))
        return f

    def rhs_derivs(self, x, lam, eps=0.0):
        x = np.ascontiguousarray(x, float); lam = np.ascontiguousarray(lam, float)
        f, fx, H = np.zeros(8), np.zeros((8, 8)), np.zeros((8, 8))
        lib().oracle_rhs_derivs(C.byref(self.p), _p(self.tab), self.nt, C.c_double(eps), _p(x), _p(lam), _p(f), _p(fx), _p(H))
        return f, fx, H

    def cost_derivs(self, x, terminal=False, eps=0.0):
        x = np.ascontiguousarray(x, float)
        v = C.c_double(); g, H = np.zeros(8), np.zeros((8, 8))
        lib().oracle_cost_derivs(C.byref(self.p), _p(self.tab), self.nt, C.c_double(eps), _p(x), int(terminal), C.byref(v), _p(g), _p(H))
        return v.value, g, H

    def cons_derivs(self, x, eps=0.0):
        x = np.ascontiguousarray(x, float)
        v, g, H = np.zeros(3), np.zeros((3, 8)), np.zeros((3, 8, 8))
        lib().oracle_cons_derivs(C.byref(self.p), _p(self.tab), self.nt, C.c_double(eps), _p(x), _p(v), _p(g), _p(H))
        return v, g, H

    def ell_derivs(self, x):
        x = np.ascontiguousarray(x, float)
        v, g, H = np.zeros(2), np.zeros((2, 8)), np.zeros((2, 8, 8))
        rc = lib().oracle_ell_derivs(C.byref(self.p), _p(x), _p(v), _p(g), _p(H))
        assert rc == 0
        return v, g, H

    def slip_forces(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), float)
        a, F = np.zeros((x.shape[0], 2)), np.zeros((x.shape[0], 2))
        lib().oracle_slip_forces(C.byref(self.p), _p(x), x.shape[0], _p(a), _p(F))
        return a, F

    def plant_step(self, x, u, dt=None, n_sub=400):
        x = np.ascontiguousarray(np.atleast_2d(x), float); u = np.ascontiguousarray(np.atleast_2d(u), float)
        xn = np.zeros_like(x)
        lib().oracle_plant_step(C.byref(self.p), _p(self.tab), self.nt, _p(x), _p(u), x.shape[0],
                                C.c_double(dt if dt is not None else self.o.t_step), int(n_sub), int(self.o.periodic_tables), _p(xn))
        return xn

    # ---- NLP solve ----------------------------------------------------------------
    def solve(self, x0, N, uprev=None, warm=None, nthreads=0, prev_status=None, sticky=None):
        """x0: (B,8).  warm: dict(X,C,U,L1,L2) of a previous solve or None (do_mpc set_initial_guess).
        prev_status: status of the solve `warm` comes from - when `warm` is a result of this method its "status_solver" is used
        (the solver's own termination status; "status" may be INFEASIBLE by the node-0 rule) - (option warm_reset_on_fail of include/ltompc.h: an instance
        whose previous solve did not converge keeps the primal point, restarts L1 = L2 = 0 and the barrier at mu_init).
        sticky: int32 array (B,), in/out: option resto_sticky (ticks for which an instance starts in elastic mode); the
        caller keeps it between ticks (the device library keeps it in the handle).
        Returns dict(u0, X, C, U, L1, L2, status, iters, kkt, obj, mu, n_reg, n_lsfail, n_soc, n_resto, viol, g0, n_fallback)."""
        x0 = np.ascontiguousarray(np.atleast_2d(x0), float)
        B = x0.shape[0]
        uprev = np.zeros((B, 2)) if uprev is None else np.ascontiguousarray(np.atleast_2d(uprev), float)
        if warm is None:
            X, Cc, U = np.zeros((B, N + 1, 8)), np.zeros((B, N, 8)), np.zeros((B, N, 2))
            L1, L2 = np.zeros((B, N, 8)), np.zeros((B, N, 8))
        else:
            X, Cc, U, L1, L2 = (np.ascontiguousarray(warm[k], float).copy() for k in ("X", "C", "U", "L1", "L2"))
        u0, st = np.zeros((B, 2)), np.zeros((B, 18))
        ni = int(lib().oracle_num_ineq(C.byref(self.p)))
        Tt, Nu = np.zeros((B, N, ni)), np.zeros((B, N, ni))
        ps = None
        if warm is not None and prev_status is not None:
            # (what counts is how the solver itself ended, before the node-0 rule of options.node0_check turned a converged
            #  solve into INFEASIBLE: a result dict of this class carries it)
            if isinstance(warm, dict) and "status_solver" in warm: prev_status = warm["status_solver"]
            ps = np.ascontiguousarray(np.asarray(prev_status).reshape(B), dtype=np.int32)
        if sticky is not None:
            assert sticky.dtype == np.int32 and sticky.shape == (B,) and sticky.flags["C_CONTIGUOUS"]
        ip = C.POINTER(C.c_int)
        lib().oracle_solve_batch(C.byref(self.p), C.byref(self.o), _p(self.tab), self.nt, int(N), int(B), _p(x0),
                                 _p(uprev), int(warm is not None), _p(X), _p(Cc), _p(U), _p(L1), _p(L2), _p(u0),
                                 _p(st), int(nthreads), _p(Tt), _p(Nu),
                                 ps.ctypes.data_as(ip) if ps is not None else None,
                                 sticky.ctypes.data_as(ip) if sticky is not None else None)
        return dict(u0=u0, X=X, C=Cc, U=U, L1=L1, L2=L2, T=Tt, NU=Nu, status=st[:, 0].astype(int), iters=st[:, 1].astype(int),
                    kkt=st[:, 2], obj=st[:, 3], mu=st[:, 4], n_reg=st[:, 5].astype(int), n_lsfail=st[:, 6].astype(int),
                    n_soc=st[:, 7].astype(int), n_resto=st[:, 8].astype(int), viol=st[:, 9], g0=st[:, 10],
                    n_fallback=st[:, 11].astype(int), n_shift=st[:, 12].astype(int), status_solver=st[:, 13].astype(int), rd_max=st[:, 14], mu_stay_max=st[:, 15].astype(int), trig=st[:, 16].astype(int), penalty=st[:, 17])


class VpVehicle(C.Structure):
    """ltompc_vp_vehicle (include/ltompc.h)."""
    _fields_ = [("kind", C.c_int), ("n_map", C.c_int), ("mass", C.c_double), ("friction_coef", C.c_double), ("lam", C.c_double),
                ("D", C.c_double), ("T", C.c_double), ("C_m", C.c_double), ("Cr_0", C.c_double), ("Cr_2", C.c_double),
                ("map_v", C.c_double * 16), ("map_f", C.c_double * 16)]


def velocity_profile(veh: VpVehicle, S, K, s_max):
    """CPU restatement of src/velocity.py:14-76 for a batch of profiles: (v, v_local, v_acclim, v_declim), each (B, n)."""
    S, K = np.ascontiguousarray(np.atleast_2d(S), float), np.ascontiguousarray(np.atleast_2d(K), float)
    sm = np.ascontiguousarray(s_max, float)
    out = [np.empty_like(S) for _ in range(4)]
    lib().oracle_velocity_profile(C.byref(veh), S.shape[1], S.shape[0], _p(S), _p(K), _p(sm), *(_p(a) for a in out))
    return tuple(out)


def num_threads() -> int:
    return int(lib().oracle_num_threads())
