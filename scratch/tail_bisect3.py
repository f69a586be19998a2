import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, MI, ticks = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x0 = ltompc.sample_x0(T, max(B, 2))[:B]
o = ltompc.default_options(); o.max_iter = MI
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
for t in range(ticks):
    u0 = m.make_step(x0)
    print("B", B, "N", N, "max_iter", MI, "tick", t, "ok status", np.bincount(m.status, minlength=5), "iters max", m.iters.max(), flush=True)
    x0 = m.plant_step(x0, u0)
