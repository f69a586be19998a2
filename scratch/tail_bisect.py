import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
x0 = ltompc.X0_REFERENCE[None].copy()
o = ltompc.default_options(); o.max_iter = 30
m = ltompc.BatchedMPC(T, 10, 1, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
print("mask", os.environ.get("LTOMPC_DEBUG_SWEEPS"), "ok u0", u0, m.status, m.iters, flush=True)
