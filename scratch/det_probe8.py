import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
def run(mi):
    o = ltompc.default_options(); o.max_iter = mi; o.n_linesearch = 1
    m = ltompc.BatchedMPC(T, N, B, options=o)
    m.set_initial_guess(x0); u0 = m.make_step(x0)
    sp = m.debug_fetch(3).reshape(3, N, B); m.close(); return sp
runs = [run(1) for _ in range(4)]
for r in range(1, 4):
    d = runs[r] != runs[0]
    print("run", r, "planes with diffs", [int(d[p].sum()) for p in range(3)], "k with diffs in gphid", np.unique(np.where(d[2])[0]), "apri k", np.unique(np.where(d[0])[0]))
print("gphid[k=39, b=0..5] per run:", [runs[r][2, 39, :6] for r in range(4)])
print("gphid[k=38, b=0..5] run0:", runs[0][2, 38, :6])
