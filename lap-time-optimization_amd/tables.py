"""Host-side table builder: race-line / boundary JSON files -> the four 846-point look-up tables.

Own implementation of the set-up arithmetic the reference does in
  src/path.py:11-26   (periodic cubic B-spline through the control points, chord-length parameter)
  src/path.py:156-172 (arc length by cumulative trapezoid of |dP/du| on u = linspace(0, length, n))
  src/path.py:132-154 (signed curvature on the uniform grid linspace(0, s_max, n), u from np.interp)
  src/mpc/track.py:113-169 (distance from each race-line sample to the nearest sampled boundary point
                            within 10 m; ValueError if there is none)
  src/mpc/track.py:39-42   (velocities.json laid, as is, on the arc-length grid)
The result (TrackTables) is what the HIP kernels read; nothing here runs per control tick.
scipy's FITPACK wrappers are used exactly as the reference uses them (north_star: host code stays Python).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass

import numpy as np
from scipy.integrate import cumulative_trapezoid
from scipy.interpolate import splev, splprep

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


@dataclass
class TrackTables:
    """Piece-wise-linear tables of the MPC.

    s_kappa : (n,) uniform grid for kappa          (path.py:142-143)
    kappa   : (n,) signed curvature [1/m]
    s_arc   : (n,) non-uniform arc-length grid for n_left / n_right / v_ref (mpc/track.py:30-42)
    n_left, n_right : (n,) distance race line -> boundary [m]
    v_ref   : (n,) reference speed [m/s]
    """
    s_kappa: np.ndarray
    kappa: np.ndarray
    s_arc: np.ndarray
    n_left: np.ndarray
    n_right: np.ndarray
    v_ref: np.ndarray

    @property
    def n(self) -> int:
        return int(self.kappa.shape[0])

    @property
    def s_max(self) -> float:
        return float(self.s_arc[-1])

    def packed(self) -> np.ndarray:
        """(6, n) float64, C-contiguous: rows s_kappa, kappa, s_arc, n_left, n_right, v_ref (the C-ABI layout)."""
        return np.ascontiguousarray(
            np.stack([self.s_kappa, self.kappa, self.s_arc, self.n_left, self.n_right, self.v_ref]), dtype=np.float64)

    @staticmethod
    def load_npz(path: str) -> "TrackTables":
        z = np.load(path, allow_pickle=False)
        return TrackTables(*(np.asarray(z[k], dtype=np.float64) for k in
                             ("s_kappa", "kappa", "s_arc", "n_left", "n_right", "v_ref")))


def _load_xy(path):
    with open(path) as f:
        d = json.load(f)
    return np.array([d["path"]["x"], d["path"]["y"]], dtype=float)


class _Spline:
    """Periodic cubic B-spline through control points + arc-length sampling (path.py:18-33, 87-94, 156-185)."""

    def __init__(self, controls: np.ndarray, n_samples: int, closed: bool = True):
        chord = np.append(0.0, np.cumsum(np.linalg.norm(np.diff(controls, axis=1), axis=0)))
        self.tck, _ = splprep(controls, u=chord, k=3, s=0, per=closed)
        self.length = chord[-1]
        self.u = np.linspace(0.0, self.length, n_samples)
        dx, dy = splev(self.u, self.tck, der=1)
        self.arc = cumulative_trapezoid(np.sqrt(dx * dx + dy * dy), self.u, initial=0)

    def points(self):
        return np.asarray(splev(self.u, self.tck))

    def signed_curvature_on_uniform_s(self, n_samples: int):
        s = np.linspace(0.0, self.arc[-1], n_samples)
        u = np.interp(s, self.arc, self.u)
        dx, dy = splev(u, self.tck, der=1)
        ddx, ddy = splev(u, self.tck, der=2)
        return s, (dx * ddy - dy * ddx) / (dx * dx + dy * dy) ** 1.5


def _nearest_boundary_distance(line_pts: np.ndarray, bound_pts: np.ndarray, radius: float = 10.0) -> np.ndarray:
    # mpc/track.py:136-159: the perpendicular-line sort only orders candidates; the value returned is the
    # smallest Euclidean distance (<= radius) from the race-line sample to any sampled boundary point.
    d = np.hypot(line_pts[0][:, None] - bound_pts[0][None, :], line_pts[1][:, None] - bound_pts[1][None, :])
    dmin = d.min(axis=1)
    if np.any(dmin > radius):
        raise ValueError(f"No point found within the radius of {radius}")
    return dmin


def build_tables(track_dir: str | None = None, n_samples: int = 846) -> TrackTables:
    """Build the LUTs from `path.json, left.json, right.json, velocities.json` in `track_dir`.

    Default directory is the shipped MX-5 / buckmore / curvature race line, the only one the reference's
    MPC can run on; n_samples = 846 as hard-coded at src/mpc.py:88.
    """
    track_dir = track_dir or os.path.join(_DATA, "tracks", "buckmore_mx5_curvature")
    line = _Spline(_load_xy(os.path.join(track_dir, "path.json")), n_samples)
    left = _Spline(_load_xy(os.path.join(track_dir, "left.json")), n_samples)
    right = _Spline(_load_xy(os.path.join(track_dir, "right.json")), n_samples)
    with open(os.path.join(track_dir, "velocities.json")) as f:
        v = np.array(json.load(f)["velocities"], dtype=float)
    if v.shape[0] != n_samples:
        raise ValueError(f"velocities.json has {v.shape[0]} entries, need n_samples={n_samples} (src/mpc.py:88)")
    s_k, kappa = line.signed_curvature_on_uniform_s(n_samples)
    # the reference evaluates the race line at u = interp(arc, arc, u) = u_sampled for the distance tables
    lp = line.points()
    return TrackTables(s_kappa=s_k, kappa=kappa, s_arc=line.arc.copy(),
                       n_left=_nearest_boundary_distance(lp, left.points()),
                       n_right=_nearest_boundary_distance(lp, right.points()),
                       v_ref=v)


def default_vehicle_json() -> str:
    return os.path.join(_DATA, "vehicles", "MX5.json")
