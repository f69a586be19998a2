"""Kernels instantiated for the reference's bound pattern vs the run-time pattern ones: same bits?"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 300, 20
x0 = ltompc.sample_x0(T, B, seed=9)
def run(maxit):
    o = ltompc.default_options(); o.max_iter = maxit
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    u = m.make_step(x0); it = m.iterate(); s = m.stats(); m.close()
    return u, it, s
for maxit in (1, 1000):
    os.environ.pop("LTOMPC_BOUNDS", None)
    a = run(maxit)
    for which in ("eval", "expand", "linesearch", "step1", "any"):
        os.environ["LTOMPC_BOUNDS"] = which
        b = run(maxit)
        d = {k: float(np.abs(a[1][k] - b[1][k]).max()) for k in ("X", "C", "U", "L1", "L2", "T", "NU")}
        bad = np.where(np.abs(a[1]["X"] - b[1]["X"]).max(axis=(1, 2)) > 0)[0]
        print(maxit, which, d, "instances that differ:", bad[:10], len(bad), flush=True)
