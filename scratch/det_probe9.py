import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 1; o.n_linesearch = 1
m = ltompc.BatchedMPC(T, N, B, options=o)
m.set_initial_guess(x0); u0 = m.make_step(x0)
# after: iteration 0 complete (update applied), then eval+riccati of iteration 1.  SP/dT/dC are from iteration 0's expand,
# but dX/dU were overwritten by iteration 1's riccati, and X/U/T were updated.  => undo the update with alpha to check.
sp = m.debug_fetch(3).reshape(3, N, B)
st = m.debug_fetch(12).reshape(-1, B)
print("alpha", st[5, :4], "mu", st[0, :4])
m.close()
