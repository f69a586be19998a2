"""Batch-1 solve loop (for a kernel trace): 12 warm ticks of make_step + plant."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
x0 = ltompc.X0_REFERENCE[None].copy()
m = ltompc.BatchedMPC(T, 40, 1); m.set_initial_guess(x0)
for t in range(12):
    t0 = time.perf_counter(); u = m.make_step(x0); dt = time.perf_counter() - t0
    if t >= 9: print(f"tick {t}: {dt*1e3:.2f} ms, iters {m.iters[0]}, launched {m.timing()['ip_iterations']}")
    x0 = m.plant_step(x0, u)
