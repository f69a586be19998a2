"""Host-side mirror of the reference's MPC interface (same names, argument meaning and error behaviour), over the GPU solver.

    reference                                   here
    ---------------------------------------     ------------------------------------------------
    mpc.track.Track(veh, track, method, n)      Track(...)            src/mpc/track.py:11
    mpc.model.VehicleModel(json, track)         VehicleModel(...)     src/mpc/model.py:12
    mpc.controller.Controller(model, costs,     Controller(...)       src/mpc/controller.py:9
        n_horizon=10, t_step=0.1, n_robust=0)
    controller.mpc.x0 = x0                      same                  src/mpc.py:117
    controller.mpc.set_initial_guess()          same                  src/mpc.py:118
    u0 = controller.mpc.make_step(x0)           same, (8,1) -> (2,1)  src/mpc.py:142   <- the hot path
    mpc.simulator.Simulator(model).simulator    same, make_step(u0)   src/mpc/simulator.py:14-20, mpc.py:114,143

A counterpart of the reference's closed-loop driver (src/mpc.py:86-173) is `closed_loop()` below.
"""
from __future__ import annotations

import json
import os
import re

import numpy as np

from . import _lib
from .solver import BatchedMPC
from .tables import TrackTables, build_tables, default_vehicle_json, _DATA


class Track:
    """Look-up tables of one (vehicle, track, method) race line (src/mpc/track.py:11-42)."""

    def __init__(self, vehicle_name="MX-5", track_name="buckmore", method_name="curvature", n_samples=846,
                 data_dir: str | None = None):
        # the reference resolves <cwd>/data/plots/<vehicle>/<track>/<method> (mpc/track.py:12-13); the one race line
        # it can actually run on ships with this package
        if data_dir is None:
            key = f"{track_name}_{vehicle_name.lower().replace('-', '')}_{method_name}"
            data_dir = os.path.join(_DATA, "tracks", key)
        if not os.path.isdir(data_dir):
            raise FileNotFoundError(data_dir)
        self.n_samples = n_samples
        self.tables: TrackTables = build_tables(data_dir, n_samples)


class VehicleModel:
    """Vehicle parameters (src/mpc/model.py:12-64).  Same loader semantics: JSON with // and /* */ comments,
    reads exactly the keys the reference reads; D_f, D_r are NOT read and stay 1.0."""

    def __init__(self, params_file_path: str | None, track: Track, torque_vectoring: bool = False):
        """torque_vectoring: use the file's `ptv` in the r equation (model.py:162, `Mtv = self.ptv * (rt - r)`, which the
        reference has commented out in favour of `Mtv = 0.0`); False = the reference's behaviour."""
        self.track = track
        self.params = _lib.default_params()
        self.load_params(params_file_path or default_vehicle_json())
        self.params.ptv = float(self.ptv) if torque_vectoring else 0.0

    @staticmethod
    def remove_comments(json_str: str) -> str:
        json_str = re.sub(r"//.*", "", json_str)
        return re.sub(r"/\*.*?\*/", "", json_str, flags=re.DOTALL)

    def load_params(self, path: str):
        with open(path) as f:
            data = json.loads(self.remove_comments(f.read()))
        p = self.params
        p.inertia_z = data["rotational_inertia"]
        self.name = data["name"]
        p.mass = data["mass"]
        p.length_f, p.length_r, p.width = data["length_f"], data["length_r"], data["width"]
        p.B_f, p.C_f = data["frontTire"]["B_f"], data["frontTire"]["C_f"]
        p.B_r, p.C_r = data["rearTire"]["B_r"], data["rearTire"]["C_r"]
        p.C_m = data["control"]["C_m"]
        p.Cr_0, p.Cr_2 = data["Cr_0"], data["Cr_2"]
        self.ptv = data["ptv"]  # read and unused, like the reference (Mtv = 0, model.py:164)


class _MPC:
    """The subset of do_mpc.controller.MPC that src/mpc.py touches."""

    def __init__(self, solver: BatchedMPC):
        self._solver = solver
        self._x0 = None
        self._guess_set = False

    @property
    def x0(self):
        return self._x0

    @x0.setter
    def x0(self, val):
        val = np.asarray(val, dtype=np.float64)
        if val.size != 8 * self._solver.B:
            raise AssertionError(f"x0 must have {8 * self._solver.B} elements, got shape {val.shape}")
        self._x0 = val.reshape(self._solver.B, 8).copy()

    def set_initial_guess(self):
        if self._x0 is None:
            raise AssertionError("set mpc.x0 before set_initial_guess()")
        self._solver.set_initial_guess(self._x0)
        self._guess_set = True

    def make_step(self, x0):
        x0 = np.asarray(x0, dtype=np.float64)
        single = x0.shape == (8, 1) or x0.shape == (8,)
        if not self._guess_set:  # do_mpc warns and uses its default guess; here: all slots = x0, like set_initial_guess
            self._x0 = x0.reshape(self._solver.B, 8).copy()
            self.set_initial_guess()
        u0 = self._solver.make_step(x0.reshape(self._solver.B, 8))
        self.solver_stats = dict(status=self._solver.status.copy(), iters=self._solver.iters.copy())
        return u0.reshape(2, 1) if single else u0


class Controller:
    """src/mpc/controller.py:9-34: NLP weights, bounds and IPOPT settings; builds the device solver."""

    def __init__(self, model: VehicleModel, control_costs, n_horizon: int = 10, t_step: float = 0.1, n_robust: int = 0,
                 batch: int = 1, device: int = 0, options=None, soft_constraint: bool = False,
                 penalty_term_cons: float = 100.0, traction_ellipse: bool = False, rho: float = 1.0, alpha: float = 1.0,
                 ellipse_penalty: float = 1.0, ellipse_radius=None):
        """soft_constraint / penalty_term_cons: the keyword arguments of do_mpc's set_nl_cons, applied to the two track
        constraints of controller.py:69-70 (the reference passes neither: hard constraints, with which the closed loop
        stops converging part-way round buckmore; see options.soft_rho in include/ltompc.h and DESIGN.md §6).
        traction_ellipse: register the two friction-ellipse constraints the reference has commented out (controller.py:72-74,
        `set_constraints(rho, alpha)` of controller.py:57 with its rho = alpha = 1, `soft_constraint=True`, do_mpc's default
        penalty 1): long = rho C_m T / 2, long^2 + F_y^2 <= (alpha D)^2 per axle.  ellipse_radius = None takes the reference's
        literal D_f = D_r = 1.0 (a radius of 1 N: unsatisfiable, which is presumably why the lines are commented out); a
        pair (D_f, D_r) in newtons, e.g. the peak lateral forces F_N D, gives the physical constraint."""
        control_costs = np.asarray(control_costs, dtype=np.float64)
        assert control_costs.shape == (2, 1)  # controller.py:38
        if n_robust != 0:
            raise NotImplementedError("multi-stage robust MPC is disabled in the reference (n_robust=0)")
        self.model = model
        self.t_step = t_step
        p = model.params
        p.r_du[0], p.r_du[1] = float(control_costs[0, 0]), float(control_costs[1, 0])
        p.q_n, p.q_mu, p.q_B = 0.5, 3.0, 1e-2  # controller.py:29
        o = options or _lib.default_options()
        o.t_step = t_step
        if soft_constraint:
            if not penalty_term_cons > 0:
                raise ValueError("penalty_term_cons must be positive")
            o.soft_rho = float(penalty_term_cons)
        if traction_ellipse:
            if not ellipse_penalty > 0:
                raise ValueError("ellipse_penalty must be positive")
            Df, Dr = (p.D_f, p.D_r) if ellipse_radius is None else ellipse_radius
            p.ell_penalty, p.ell_rho, p.ell_D_f, p.ell_D_r = float(ellipse_penalty), float(rho), float(alpha * Df), float(alpha * Dr)
        self.solver = BatchedMPC(model.track.tables, n_horizon=n_horizon, batch=batch, params=p, options=o, device=device)
        model._solver = self.solver  # (Simulator(model) runs the plant on the same handle)
        self.mpc = _MPC(self.solver)


class _Sim:
    """The subset of do_mpc.simulator.Simulator that src/mpc.py touches: attribute x0, make_step(u0) -> y."""

    def __init__(self, model: VehicleModel, n_sub: int):
        self._model, self._n_sub, self.x0 = model, n_sub, None
        self._solver = None

    def _plant(self, batch: int) -> BatchedMPC:
        # the plant step runs on the device through a solver handle: the controller's (registered on the model by
        # Controller) when it has the same batch size, else a minimal handle of its own
        if self._solver is None or self._solver.B != batch:
            shared = getattr(self._model, "_solver", None)
            if shared is not None and shared.B == batch:
                self._solver = shared
            else:
                self._solver = BatchedMPC(self._model.track.tables, n_horizon=2, batch=batch, params=self._model.params)
        return self._solver

    def make_step(self, u0):
        u0 = np.asarray(u0, dtype=np.float64)
        single = u0.shape == (2, 1)
        if self.x0 is None:
            raise AssertionError("set simulator.x0 before make_step()")
        u = u0.reshape(-1, 2)
        solver = self._plant(u.shape[0])
        x = np.asarray(self.x0, dtype=np.float64).reshape(solver.B, 8)
        xn = solver.plant_step(x, u, self._n_sub)
        self.x0 = xn.reshape(8, 1) if single else xn
        return self.x0


class Simulator:
    """src/mpc/simulator.py:14-20: `Simulator(model).simulator` is the plant with t_step = 0.1 (the plotting part of the
    reference's class is out of scope).  Takes the VehicleModel like the reference (mpc.py:114); a Controller is accepted
    too (its model is used)."""

    def __init__(self, model, n_sub: int = 400):
        if isinstance(model, Controller):
            model = model.model
        self.model = model
        self.simulator = _Sim(model, n_sub)


def closed_loop(controller: Controller, x0, steps: int, out_json: str | None = None):
    """Counterpart of the reference's loop (src/mpc.py:117-159): make_step -> plant -> identity estimator.
    Returns dict(x, y, u, Fy, alpha) with the reference's sim_results.json schema."""
    x0 = np.reshape(np.asarray(x0, dtype=np.float64), (-1, 1))
    sim = Simulator(controller.model).simulator
    sim.x0 = x0
    controller.mpc.x0 = x0
    controller.mpc.set_initial_guess()
    X = np.zeros((steps + 1, 8, 1)); Y = np.zeros((steps + 1, 8, 1)); U = np.zeros((steps + 1, 2, 1))
    Fys = np.zeros((steps + 1, 2)); alphas = np.zeros((steps + 1, 2))
    X[0] = Y[0] = x0
    for i in range(1, steps + 1):
        u0 = controller.mpc.make_step(x0)
        y = sim.make_step(u0)
        x0 = y  # StateFeedback estimator is the identity (mpc.py:119-120,144)
        X[i], Y[i], U[i] = x0, y, u0
        a, F = controller.solver.slip_forces(x0.reshape(1, 8))
        alphas[i], Fys[i] = a[0], F[0]
    data = {"x": X.tolist(), "y": Y.tolist(), "u": U.tolist(), "Fy": Fys.tolist(), "alpha": alphas.tolist()}
    if out_json:
        with open(out_json, "w") as f:
            json.dump(data, f)
    return data
