"""Extreme initial states inside a normal batch: the solver must return a status for them (no hang, no crash), and the other
instances of the batch must get exactly the result they get without them."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N, B = 20, 256
x0 = ltompc.sample_x0(T, B, seed=5)
o = ltompc.default_options(); o.latency_mode, o.max_iter = 2, 150
def run(x):
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x)
    out = []
    xx = x.copy()
    for t in range(2):
        u = m.make_step(xx); s = m.stats(); out.append((u.copy(), s["status"].copy(), s["iters"].copy())); xx = m.plant_step(xx, u, 50)
    m.close(); return out
ref = run(x0)
weird = {
    "vx = 0": lambda x: x.__setitem__(3, 0.0),
    "vx = 0, vy = 0, r = 0": lambda x: (x.__setitem__(3, 0.0), x.__setitem__(4, 0.0), x.__setitem__(5, 0.0)),
    "vx = 1e-9": lambda x: x.__setitem__(3, 1e-9),
    "vx = 120 m/s": lambda x: x.__setitem__(3, 120.0),
    "n = +40 m (far off the track)": lambda x: x.__setitem__(1, 40.0),
    "n = -40 m": lambda x: x.__setitem__(1, -40.0),
    "mu = pi/2": lambda x: x.__setitem__(2, np.pi / 2),
    "mu = -3": lambda x: x.__setitem__(2, -3.0),
    "s = -50 m (before the table)": lambda x: x.__setitem__(0, -50.0),
    "s = 5000 m (beyond the table)": lambda x: x.__setitem__(0, 5000.0),
    "delta = 1.5 rad, T = 5": lambda x: (x.__setitem__(6, 1.5), x.__setitem__(7, 5.0)),
    "r = 50 rad/s": lambda x: x.__setitem__(5, 50.0),
    "vy = -30 m/s": lambda x: x.__setitem__(4, -30.0),
    "1 - n kappa = 0 (on the centre of curvature)": None,
}
x = x0.copy()
idx = {}
pos = [3, 11, 20, 37, 64, 65, 100, 127, 128, 129, 200, 201, 250, 255]
for (name, f), j in zip(weird.items(), pos):
    idx[name] = j
    if f is not None: f(x[j])
    else:
        kap = np.interp(x[j, 0], T.s_kappa, T.kappa); x[j, 1] = 1.0 / kap if abs(kap) > 1e-6 else 1e6
got = run(x)
others = np.ones(B, bool); others[pos] = False
for t in range(2):
    print(f"tick {t}: the {others.sum()} untouched instances identical to the run without the extreme ones: u0 {np.array_equal(got[t][0][others], ref[t][0][others])}, "
          f"status {np.array_equal(got[t][1][others], ref[t][1][others])}, iters {np.array_equal(got[t][2][others], ref[t][2][others])}")
for name, j in idx.items():
    print(f"  {name:46s}: status {ltompc.STATUS_NAMES[got[0][1][j]] if hasattr(ltompc, 'STATUS_NAMES') else got[0][1][j]} after {got[0][2][j]} iterations, u0 {got[0][0][j]}, finite {np.isfinite(got[0][0][j]).all()} | next tick: status {got[1][1][j]}, u0 finite {np.isfinite(got[1][0][j]).all()}")
# the oracle on the same extreme states (cold tick)
from oracle import oracle as orc
oo = orc.default_options(); oo.max_iter = 150
O = orc.Oracle(T.packed(), options=oo)
r = O.solve(x[pos], N, nthreads=8)
print("GPU vs oracle on the extreme states (status, iterations):")
for (name, j), q in zip(idx.items(), range(len(pos))):
    print(f"  {name:46s}: gpu ({got[0][1][j]}, {got[0][2][j]})  oracle ({r['status'][q]}, {r['iters'][q]})  |du0| {np.abs(got[0][0][j] - r['u0'][q]).max():.2e}")
