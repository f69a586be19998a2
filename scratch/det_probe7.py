import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
names = ["QP", "RC", "RS", "SP", "LS", "dX", "dU", "dC", "dT", "dNU", "nL1", "nL2", "st", "si"]
def run(mi):
    o = ltompc.default_options(); o.max_iter = mi; o.n_linesearch = 1
    m = ltompc.BatchedMPC(T, N, B, options=o)
    m.set_initial_guess(x0); u0 = m.make_step(x0)
    out = {n: m.debug_fetch(i) for i, n in enumerate(names)}
    out["X"] = m.iterate()["X"]
    m.close(); return out
for mi in (1, 2):
    runs = [run(mi) for _ in range(5)]
    for r in range(1, 5):
        diffs = []
        for n in names + ["X"]:
            a, b = runs[0][n], runs[r][n]
            neq = (a != b) & ~(np.isnan(a.astype(float)) & np.isnan(b.astype(float)))
            if neq.any(): diffs.append((n, int(neq.sum()), np.where(neq.ravel())[0][:4].tolist()))
        print("mi", mi, "run", r, diffs, flush=True)
