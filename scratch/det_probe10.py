import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
def run(mi):
    o = ltompc.default_options(); o.max_iter = mi; o.n_linesearch = 1
    m = ltompc.BatchedMPC(T, N, B, options=o)
    m.set_initial_guess(x0); u0 = m.make_step(x0)
    d = m.debug_fetch(14).reshape(8, N, B); m.close(); return d
runs = [run(1) for _ in range(4)]
for r in range(1, 4):
    print("run", r, "planes differing (count):", [(p, int((runs[r][p] != runs[0][p]).sum())) for p in range(8)])
print("k=39 b=0: per run planes 0..7:", [np.round(runs[r][:, 39, 0], 5).tolist() for r in range(4)])
