"""ctypes binding of libltompc.so (include/ltompc.h).  No CPU fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._build import LIB

NX, NU = 8, 2
NO_BOUND = 1.0e30
STATUS_NAMES = {0: "solved", 1: "acceptable", 2: "max_iter", 3: "numerical", 4: "stalled", 5: "infeasible"}


class Params(C.Structure):
    """ltompc_params (include/ltompc.h)."""
    _fields_ = [(n, C.c_double) for n in (
        "mass", "inertia_z", "length_f", "length_r", "width", "B_f", "C_f", "D_f", "B_r", "C_r", "D_r",
        "C_m", "Cr_0", "Cr_2", "gravity", "ptv", "q_n", "q_mu", "q_vy", "q_v", "vref_scale", "q_B")] + [
        ("r_du", C.c_double * 2), ("x_lb", C.c_double * 8), ("x_ub", C.c_double * 8),
        ("u_lb", C.c_double * 2), ("u_ub", C.c_double * 2)] + [
        (n, C.c_double) for n in ("ell_penalty", "ell_rho", "ell_D_f", "ell_D_r")]


class Options(C.Structure):
    """ltompc_options (include/ltompc.h)."""
    _fields_ = [(n, C.c_double) for n in (
        "t_step", "tol", "acceptable_tol", "mu_init", "mu_min", "kappa_eps", "kappa_mu", "theta_mu", "tau_min",
        "bound_push", "s_max", "delta_w_first", "smooth_eps_min", "smooth_scale", "mu_init_warm", "soft_rho", "resto_rho",
        "resto_rho_max", "resto_rho_factor", "dual_inf_max")] + [
        ("max_iter", C.c_int), ("acceptable_iter", C.c_int), ("n_linesearch", C.c_int), ("stall_iter", C.c_int),
        ("max_ls_fail", C.c_int), ("warm_shift", C.c_int), ("warm_reset_on_fail", C.c_int), ("periodic_tables", C.c_int), ("max_soc", C.c_int), ("resto_sticky", C.c_int),
        ("node0_check", C.c_int), ("warm_fallback_iter", C.c_int), ("resto_shift_retry", C.c_int), ("max_mu_stay", C.c_int), ("infeasible_sticky", C.c_int), ("latency_mode", C.c_int)]


class LtompcError(RuntimeError):
    pass


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib() -> C.CDLL:
    """Load libltompc.so; raise if it has not been built (there is deliberately no other compute path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise LtompcError(
                f"{LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own HIP runtime; libltompc.so links the system one.  In one process the first one loaded
        # serves both, and it has to be torch's: with the system runtime initialised first (a handle created before `import torch`)
        # torch finds no devices ("No HIP GPUs are available", measured on the MI355X box).  So torch is imported here when it is
        # installed; without torch the library runs on the system runtime alone.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB)
        L.ltompc_last_error.restype = C.c_char_p
        L.ltompc_version.restype = C.c_char_p
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise LtompcError(lib().ltompc_last_error().decode())


def dptr(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def iptr(a: np.ndarray):
    return a.ctypes.data_as(_ip)


def default_params() -> Params:
    p = Params()
    lib().ltompc_default_params(C.byref(p))
    return p


def default_options() -> Options:
    o = Options()
    lib().ltompc_default_options(C.byref(o))
    return o
