import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
def run(mi):
    o = ltompc.default_options(); o.max_iter = mi
    m = ltompc.BatchedMPC(T, N, B, options=o)
    m.set_initial_guess(x0); u0 = m.make_step(x0); it = m.iterate(); st = m.stats(); m.close()
    return it, st
mi = 2
runs = [run(mi) for _ in range(6)]
a, sa = runs[0]
for r, (b, sb) in enumerate(runs[1:]):
    for i in (696, 731, 6852, 6939, 2380):
        out = []
        for key in ("X", "C", "U", "L1", "L2", "T", "NU"):
            d = np.abs(a[key][i] - b[key][i]).reshape(a[key].shape[1], -1).max(axis=1)
            ks = np.where(d > 0)[0]
            if len(ks): out.append((key, ks.tolist()[:8], float(d.max())))
        if out: print("run", r + 1, "inst", i, out)
