"""Synthetic initial states for the batched benchmark / parity configurations.

The reference runs exactly one MPC instance from x0 = [0,0,0,5,0,0,0,0.1] (src/mpc.py:107-110); the batch
configurations of BASELINE.json sample many x0 along the buckmore race line with the recipe below.
"""
from __future__ import annotations

import numpy as np

from .tables import TrackTables

X0_REFERENCE = np.array([0.0, 0.0, 0.0, 5.0, 0.0, 0.0, 0.0, 0.1])  # src/mpc.py:107-110
SEED = 20250614


def _interp(grid, y, s):
    return np.interp(s, grid, y)


def sample_x0(tables: TrackTables, batch: int, seed: int = SEED, width: float = 2.3,
              lookahead_margin: float = 150.0) -> np.ndarray:
    """(batch, 8) feasible initial states: s ~ U(0, s_max - margin); n inside the drivable band;
    small heading / slip; vx around 0.6 v_ref(s) (the objective's speed target, controller.py:53)."""
    rng = np.random.default_rng(seed)
    s = rng.uniform(0.0, tables.s_max - lookahead_margin, batch)
    nl, nr = _interp(tables.s_arc, tables.n_left, s), _interp(tables.s_arc, tables.n_right, s)
    vref = _interp(tables.s_arc, tables.v_ref, s)
    kap = _interp(tables.s_kappa, tables.kappa, s)
    n_mid, w = 0.5 * (nl - nr), 0.5 * (nl + nr - width)
    n = n_mid + rng.uniform(-0.4, 0.4, batch) * w
    mu = np.clip(rng.normal(0.0, 0.03, batch), -0.1, 0.1)
    vx = 0.6 * vref * rng.uniform(0.8, 1.1, batch)
    vy = rng.normal(0.0, 0.05, batch)
    r = kap * vx + rng.normal(0.0, 0.02, batch)
    delta = np.clip(np.arctan(3.0 * kap) + rng.normal(0.0, 0.01, batch), -0.3, 0.3)
    thr = rng.uniform(-0.2, 0.5, batch)
    return np.ascontiguousarray(np.stack([s, n, mu, vx, vy, r, delta, thr], axis=1))
