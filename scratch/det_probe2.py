import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = int(sys.argv[1]), 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 300
m = ltompc.BatchedMPC(T, N, B, options=o)
res = []
for rep in range(3):
    m.set_initial_guess(x0); u0 = m.make_step(x0); st = m.stats(); res.append((u0.copy(), st["iters"].copy()))
for rep in (1, 2):
    d = np.abs(res[rep][0] - res[0][0]).max(axis=1)
    print("B", B, "tail", os.environ.get("LTOMPC_TAIL"), "rep", rep, "differ", (d > 0).sum(), "iters differ", (res[rep][1] != res[0][1]).sum(), "idx", np.where(d > 0)[0][:8])
