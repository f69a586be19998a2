"""Replay a problem dumped by scratch/slow_dump.py in the oracle (ORACLE_TRACE=1 for the iteration log).
usage: python scratch/replay_dump.py file.npz [key=value options]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module("lap-time-optimization_amd")
from oracle import oracle as orc
d = np.load(sys.argv[1])
tables = pkg.TrackTables.load_npz(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tables_buckmore_mx5_curvature.npz"))
o = orc.default_options()
for a in sys.argv[2:]:
    k, v = a.split("="); setattr(o, k, type(getattr(o, k))(float(v)))
O = orc.Oracle(tables.packed(), options=o)
warm = {k: d[k][None] for k in ("X", "C", "U", "L1", "L2")}
r = O.solve(d["x0"][None], warm["U"].shape[1], d["up"][None], warm, prev_status=np.array([int(d["prev_status"])]))
print("GPU: status", int(d["status"]), "iters", int(d["iters"]), "n_resto", int(d["n_resto"]), "n_shift", int(d["n_shift"]), "viol %.3e" % float(d["viol"]))
print("oracle: status", r["status"][0], "solver", r["status_solver"][0], "iters", r["iters"][0], "n_resto", r["n_resto"][0], "n_shift", r["n_shift"][0], "n_reg", r["n_reg"][0], "viol %.3e" % r["viol"][0], "g0 %.2e" % r["g0"][0])
