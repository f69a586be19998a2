"""Velocity-profile generator: host-side mirror of the reference's `VelocityProfile(vehicle, s, k, s_max)`
(src/velocity.py:14-76) and of its two vehicle classes (src/vehicle.py, src/vehicleMX5.py), over ltompc_velocity_profile.
A batch of profiles per call: `VelocityProfile.batch(vehicle, S (B,n), K (B,n), s_max (B,))`."""
from __future__ import annotations

import ctypes as C
import json
import re

import numpy as np

from ._lib import check, dptr, lib


class VpVehicle(C.Structure):
    """ltompc_vp_vehicle (include/ltompc.h)."""
    _fields_ = [("kind", C.c_int), ("n_map", C.c_int), ("mass", C.c_double), ("friction_coef", C.c_double), ("lam", C.c_double),
                ("D", C.c_double), ("T", C.c_double), ("C_m", C.c_double), ("Cr_0", C.c_double), ("Cr_2", C.c_double),
                ("map_v", C.c_double * 16), ("map_f", C.c_double * 16)]


class Vehicle:
    """src/vehicle.py:11-35: point mass with an engine map and a friction circle (e.g. data/vehicles/tbr18.json)."""

    def __init__(self, path: str):
        d = json.load(open(path))
        self.name, self.mass, self.friction_coef = d["name"], d["mass"], d["frictionCoefficient"]
        self.engine_profile = [d["engineMap"]["v"], d["engineMap"]["f"]]

    def c_struct(self) -> VpVehicle:
        v = VpVehicle(kind=0, n_map=len(self.engine_profile[0]), mass=self.mass, friction_coef=self.friction_coef)
        if not 2 <= v.n_map <= 16:
            raise ValueError("engine map needs 2 .. 16 points")
        for i, (a, b) in enumerate(zip(*self.engine_profile)):
            v.map_v[i], v.map_f[i] = a, b
        return v


class VehicleMX5:
    """src/vehicleMX5.py:11-79 (the keys its loader reads; JSON with // and /* */ comments)."""

    def __init__(self, vehicle_filepath: str):
        txt = re.sub(r"/\*.*?\*/", "", re.sub(r"//.*", "", open(vehicle_filepath).read()), flags=re.DOTALL)
        d = json.loads(txt)
        self.name, self.mass = d["name"], d["mass"]
        self.D_f, self.D_r = d["frontTire"]["D_f"], d["rearTire"]["D_r"]
        self.C_m, self.Cr_0, self.Cr_2 = d["control"]["C_m"], d["Cr_0"], d["Cr_2"]
        self.T, self.friction_coef, self.ro_long = d["control"]["T"], d["control"]["lambda"], d["control"]["ro_long"]

    def c_struct(self) -> VpVehicle:
        # traction(v, k, lam=2.0): VelocityProfile calls it without lam (velocity.py:43,69)
        return VpVehicle(kind=1, n_map=0, mass=self.mass, friction_coef=self.friction_coef, lam=2.0, D=(self.D_f + self.D_r) * 0.5,
                         T=self.T, C_m=self.C_m, Cr_0=self.Cr_0, Cr_2=self.Cr_2)


class VelocityProfile:
    """VelocityProfile(vehicle, s, k, s_max=None) -> .v, .v_local, .v_acclim, .v_declim   (src/velocity.py:14-26)."""

    def __init__(self, vehicle, s, k, s_max=None, device: int = 0):
        self.vehicle, self.s, self.s_max = vehicle, s, s_max
        out = self.batch(vehicle, np.asarray(s, float)[None], np.asarray(k, float)[None], [-1.0 if s_max is None else s_max], device)
        self.v, self.v_local, self.v_acclim, self.v_declim = (a[0] for a in out)

    @staticmethod
    def batch(vehicle, S, K, s_max, device: int = 0):
        """B profiles at once: S, K (B, n); s_max (B,) with < 0 (or None entries) for open paths.  Returns (v, v_local, v_acclim, v_declim)."""
        S, K = np.ascontiguousarray(S, float), np.ascontiguousarray(K, float)
        if S.ndim != 2 or S.shape != K.shape:
            raise ValueError("s and k must be (batch, n) arrays of the same shape")
        sm = np.ascontiguousarray([-1.0 if v is None else float(v) for v in s_max], float)
        if sm.shape != (S.shape[0],):
            raise ValueError("one s_max per profile")
        out = [np.empty_like(S) for _ in range(4)]
        veh = vehicle.c_struct()
        check(lib().ltompc_velocity_profile(int(device), C.byref(veh), S.shape[1], S.shape[0], dptr(S), dptr(K), dptr(sm), *(dptr(a) for a in out)))
        return tuple(out)
