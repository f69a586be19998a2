#!/usr/bin/env python3
"""Generate the golden look-up-table fixture by running the REFERENCE's own table code.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference, it only reads the committed fixture `tests/golden/tables_buckmore_mx5_curvature.npz`.

What is imported from the reference: `src/mpc/track.py` (Track) and `src/path.py`
(ControllerReferencePath), unmodified.  Their only use of CasADi on this path is the
*construction* of `ca.interpolant(...)` objects around already-computed numpy tables
(path.py:98-101, mpc/track.py:31-42); none of the table arithmetic goes through CasADi.
CasADi is not installed here, so a 10-line stand-in module that only records the
(grid, values) pairs is placed in sys.modules.  The numbers stored below are the numpy
arrays the reference computed (scipy FITPACK + numpy), not anything produced by the stub.
"""
import os, sys, types, json
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tables_buckmore_mx5_curvature.npz")


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    ca = types.ModuleType("casadi")

    class MX:  # only used for isinstance checks in the reference
        pass

    def interpolant(name, kind, grid, values):
        g = np.asarray(grid[0], dtype=float)
        v = np.asarray(values, dtype=float)
        return lambda s: np.interp(s, g, v)

    ca.MX = MX
    ca.interpolant = interpolant
    sys.modules["casadi"] = ca
    sys.path.insert(0, os.path.join(REF, "src"))
    os.chdir(REF)
    from mpc.track import Track  # reference code

    tr = Track("MX-5", "buckmore", "curvature", 846)
    op = tr.optimal_path
    tab = np.array(op.curvature_lookup_table, dtype=float)
    out = dict(
        s_kappa=tab[:, 0].copy(),                       # uniform grid, path.py:142-143
        kappa=tab[:, 1].copy(),                         # signed curvature, path.py:145-153
        s_arc=np.asarray(op.arc_lengths_sampled, float),  # non-uniform arc grid, path.py:156-172
        u_sampled=np.asarray(op.u_sampled, float),
        n_left=np.asarray(tr.bound_dist_table["left"], float),   # mpc/track.py:113-169
        n_right=np.asarray(tr.bound_dist_table["right"], float),
        v_ref=np.asarray(tr.velocities, float),         # velocities.json laid on s_arc, mpc/track.py:39-42
    )
    np.savez_compressed(OUT, **out)
    sums = {k: float(np.sum(v)) for k, v in out.items()}
    sums["s_max"] = float(out["s_arc"][-1])
    print(json.dumps(sums, indent=1))


if __name__ == "__main__":
    main()
