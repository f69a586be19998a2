import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, MI = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x0 = ltompc.sample_x0(T, max(B, 2))[:B]
o = ltompc.default_options(); o.max_iter = MI
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
print("B", B, "N", N, "max_iter", MI, "ok status", np.bincount(m.status, minlength=5), "iters max", m.iters.max(), flush=True)
