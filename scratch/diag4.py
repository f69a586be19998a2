import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 150
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
for tick in range(3):
    x0 = m.plant_step(x0, u0)
    u0 = m.make_step(x0)
st = m.stats(); it = st["iters"]; s = st["status"]
for code in (0, 2, 4):
    sel = s == code
    if sel.sum(): print("status", code, "n", sel.sum(), "iters pct 50/90/99/max", np.percentile(it[sel], [50, 90, 99, 100]), " >40:", (it[sel] > 40).sum(), ">60:", (it[sel] > 60).sum(), ">100:", (it[sel] > 100).sum())
sel = (s == 4)
print("stalled: kkt", np.sort(st["kkt"][sel])[::12][:12], "lsfail hist", np.bincount(st["n_lsfail"][sel], minlength=9), "nreg med", np.median(st["n_reg"][sel]))
sel = (s == 0) & (it > 60)
print("slow solved:", sel.sum(), "nreg", st["n_reg"][sel][:12], "lsfail", st["n_lsfail"][sel][:12])
