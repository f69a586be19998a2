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
for mi in (2, 3):
    runs = [run(mi) for _ in range(4)]
    a, sa = runs[0]
    for r, (b, sb) in enumerate(runs[1:]):
        for key in ("X", "C", "U", "L1", "L2", "T", "NU"):
            d = np.abs(a[key] - b[key]).reshape(B, -1).max(axis=1)
            idx = np.where(d > 0)[0]
            if len(idx): print("mi", mi, "run", r + 1, key, "differs for", idx[:6], "max", d.max(), "first k:", [int(np.argmax(np.abs(a[key][i] - b[key][i]).reshape(a[key].shape[1], -1).max(axis=1) > 0)) for i in idx[:6]])
        for key in ("iters", "n_reg", "n_lsfail", "status", "mu"):
            idx = np.where(sa[key] != sb[key])[0]
            if len(idx): print("   stat", key, idx[:6], sa[key][idx[:6]], sb[key][idx[:6]])
    print("mi", mi, "stats of 696/731:", {k: sa[k][[696, 731]] for k in ("iters", "n_reg", "n_lsfail", "mu")})
