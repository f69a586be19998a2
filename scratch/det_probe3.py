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
for mi in [int(a) for a in sys.argv[1:]]:
    a, sa = run(mi); b, sb = run(mi)
    bad = set()
    for key in ("X", "C", "U", "L1", "L2", "T", "NU"):
        d = np.abs(a[key] - b[key]).reshape(B, -1).max(axis=1)
        bad |= set(np.where(d > 0)[0].tolist())
    print("max_iter", mi, "instances differing", sorted(bad)[:12], "n", len(bad), flush=True)
