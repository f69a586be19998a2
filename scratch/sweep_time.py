import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B = int(sys.argv[1])
x0 = ltompc.sample_x0(T, max(B, 8))[:B]
o = ltompc.default_options(); o.max_iter = 30
m = ltompc.BatchedMPC(T, 40, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0); x0 = m.plant_step(x0, u0)
m.set_profiling(True)
u0 = m.make_step(x0)
tm = m.timing()
print("B", B, "extra", os.environ.get("LTOMPC_DEBUG_SWEEPS"), {k: round(1e3*v/max(1,tm["launches_by_kernel"][k]),1) for k,v in tm["ms"].items()}, "us/launch; launches", tm["launches_by_kernel"]["riccati"], "nreg", m.stats()["n_reg"].sum())
