import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = int(sys.argv[1]), 40
x0 = ltompc.sample_x0(T, B) if B > 1 else ltompc.X0_REFERENCE[None].copy()
o = ltompc.default_options(); o.max_iter = 60
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
x0 = m.plant_step(x0, u0)
u0 = m.make_step(x0)
print("history", m.history().tolist())
print("done", m.timing()["ip_iterations"], np.bincount(m.stats()["status"], minlength=5))
