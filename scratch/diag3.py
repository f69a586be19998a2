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
    m.set_profiling(tick == 2)
    t0 = time.time(); u0 = m.make_step(x0); dt = time.time() - t0
    st = m.stats(); it = st["iters"]
    print(f"tick {tick}: {dt*1e3:.1f} ms launched {m.timing()['ip_iterations']} status {np.bincount(st['status'], minlength=5)} pct {np.percentile(it,[50,90,99,99.9,100])}")
print("history", m.history().tolist())
tm = m.timing(); print({k: round(v,1) for k,v in tm["ms"].items()})
