import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 300
m = ltompc.BatchedMPC(T, N, B, options=o)
m.set_initial_guess(x0); u0 = m.make_step(x0); st = m.stats()
perm = np.random.default_rng(0).permutation(B)
m.set_initial_guess(x0[perm]); u0p = m.make_step(x0[perm]); stp = m.stats()
d = np.abs(u0p - u0[perm]).max(axis=1)
bad = np.where(d > 0)[0]
print("differ:", len(bad), "status a", st["status"][perm][bad][:20], "status p", stp["status"][bad][:20], "iters a", st["iters"][perm][bad][:20], "iters p", stp["iters"][bad][:20], "maxdiff", d.max())
print("status equal:", np.array_equal(st["status"][perm], stp["status"]), "iters equal", np.array_equal(st["iters"][perm], stp["iters"]))
