#!/usr/bin/env python3
"""Generate the golden fixture of the velocity-profile generator by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference).  Imported from the reference, unmodified: `src/velocity.py`
(VelocityProfile), `src/vehicleMX5.py` (VehicleMX5), `src/vehicle.py` (Vehicle), `src/path.py` (Path; its module-level
`import casadi` is satisfied by an empty stand-in module: nothing on this path calls CasADi).
Inputs: the race line the reference ships (data/plots/MX-5/buckmore/curvature/path.json, 847 points, closed) as a Path, sampled
like Trajectory.update (`src/trajectory.py:40-52`: s = linspace(0, length, ns)[:-1], k = |curvature|); a second, OPEN case
(first 300 samples, s_max = None) and the tbr18 point-mass vehicle (engine map) exercise the other branches.
Output: tests/golden/velocity_profiles.npz with inputs (s, k, s_max, vehicle parameters) and the reference's outputs
(v_local, v_acclim, v_declim, v), plus the shipped velocities.json for a sanity comparison.
"""
import json, os, sys, types
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "velocity_profiles.npz")


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    ca = types.ModuleType("casadi")
    ca.MX = type("MX", (), {})
    sys.modules["casadi"] = ca
    sys.path.insert(0, os.path.join(REF, "src"))
    os.chdir(REF)
    from path import Path                      # reference code
    from velocity import VelocityProfile       # reference code
    from vehicleMX5 import VehicleMX5          # reference code
    from vehicle import Vehicle                # reference code

    pj = json.load(open("data/plots/MX-5/buckmore/curvature/path.json"))["path"]
    pts = np.array([pj["x"], pj["y"]])
    path = Path(pts[:, :-1], True) if np.allclose(pts[:, 0], pts[:, -1]) else Path(pts, True)
    ns = 847
    s_all = np.linspace(0, path.length, ns)
    s = s_all[:-1]
    k = path.curvature(s)
    mx5 = VehicleMX5("data/vehicles/MX5.json")
    tbr = Vehicle("data/vehicles/tbr18.json")
    out = dict(s=s, k=k, s_max=np.array(path.length))
    for name, veh in (("mx5", mx5), ("tbr18", tbr)):
        vp = VelocityProfile(veh, s, k, path.length)                 # closed path
        vo = VelocityProfile(veh, s[:300].copy(), k[:300].copy())    # open path (s_max = None)
        for tag, p in (("closed", vp), ("open", vo)):
            for f in ("v_local", "v_acclim", "v_declim", "v"):
                out[f"{name}_{tag}_{f}"] = np.asarray(getattr(p, f), float)
    out["mx5_params"] = np.array([mx5.mass, mx5.friction_coef, mx5.T, mx5.C_m, mx5.Cr_0, mx5.Cr_2, 0.5 * (mx5.D_f + mx5.D_r)])
    out["tbr18_params"] = np.array([tbr.mass, tbr.friction_coef])
    out["tbr18_engine_v"] = np.asarray(tbr.engine_profile[0], float)
    out["tbr18_engine_f"] = np.asarray(tbr.engine_profile[1], float)
    out["shipped_velocities"] = np.array(json.load(open("data/plots/MX-5/buckmore/curvature/velocities.json"))["velocities"])
    np.savez_compressed(OUT, **out)
    d = out["mx5_closed_v"] - out["shipped_velocities"]
    print("mx5 closed: v in [%.3f, %.3f]; vs shipped velocities.json: max |diff| %.3f m/s, mean %.4f" % (out["mx5_closed_v"].min(), out["mx5_closed_v"].max(), np.abs(d).max(), np.abs(d).mean()))
    print({k_: (v.shape if hasattr(v, "shape") else v) for k_, v in out.items() if not k_.startswith(("mx5_c", "mx5_o", "tbr18_c", "tbr18_o"))})


if __name__ == "__main__":
    main()
