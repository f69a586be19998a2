"""Short full-width workload for PMC counter passes: one cold solve capped at 25 iterations, B = 8192, N = 40."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
o = ltompc.default_options(); o.max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 25
x0 = ltompc.sample_x0(T, B, seed=ltompc.scenarios.SEED)
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
m.make_step(x0)
print("done", (m.status == 0).mean(), m.iters.mean())
m.close()
