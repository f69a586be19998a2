import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 400
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
for tick in range(3):
    x0 = m.plant_step(x0, u0)
    t0 = time.time(); u0 = m.make_step(x0); dt = time.time() - t0
    st = m.stats(); it = st["iters"]
    print(f"tick {tick}: {dt:.3f}s launched {m.timing()['ip_iterations']} iters pct50/90/99/99.9/max = {np.percentile(it,[50,90,99,99.9,100])} status {np.bincount(st['status'], minlength=5)}")
    for thr in (30, 40, 60, 100, 150, 200):
        print(f"    iters > {thr}: {(it>thr).sum()}", end="")
    print()
    slow = np.where(it > 100)[0]
    print("    slow: status", st["status"][slow][:20], "kkt", st["kkt"][slow][:6], "nreg", st["n_reg"][slow][:10], "lsfail", st["n_lsfail"][slow][:10])
    np.save(f"gpurun_out/slow_x0_tick{tick}.npy", x0[slow])
    np.save(f"gpurun_out/slow_idx_tick{tick}.npy", slow)
