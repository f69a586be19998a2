import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 150
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
pe = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m.set_poll_every(pe)
u0 = m.make_step(x0)
for tick in range(3):
    x0 = m.plant_step(x0, u0)
    t0 = time.perf_counter(); u0 = m.make_step(x0); dt = time.perf_counter() - t0
    print(f"poll_every {pe} tick {tick}: {dt*1e3:.1f} ms")
h = m.history()
print("(iteration, active after it, launch width):", [tuple(int(v) for v in r) for r in h[:16]])
it = m.stats()["iters"]
print("instances still iterating after iteration i:", [(i, int((it > i).sum())) for i in (8, 12, 14, 16, 18, 20, 22, 24, 26, 28, 32, 36, 40)])
