import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
nls = int(sys.argv[1])
def run(mi):
    o = ltompc.default_options(); o.max_iter = mi; o.n_linesearch = nls
    m = ltompc.BatchedMPC(T, N, B, options=o)
    m.set_initial_guess(x0); u0 = m.make_step(x0); it = m.iterate(); st = m.stats(); m.close()
    return it, st
for mi in (3, 8):
    runs = [run(mi) for _ in range(4)]
    a, sa = runs[0]
    bad = set()
    for (b, sb) in runs[1:]:
        d = np.abs(a["X"] - b["X"]).reshape(B, -1).max(axis=1); bad |= set(np.where(d > 0)[0].tolist())
    print("n_linesearch", nls, "max_iter", mi, "nondeterministic instances:", sorted(bad)[:10], len(bad))
