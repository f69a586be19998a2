"""BASELINE config 5: closed-loop lap of buckmore from the reference's x0, receding horizon N = 60."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
x = ltompc.X0_REFERENCE[None].copy()
o = ltompc.default_options(); o.max_iter = 300
o.soft_rho = float(os.environ.get('SOFT_RHO', '0'))
m = ltompc.BatchedMPC(T, N, 1, options=o); m.set_initial_guess(x)
s_end = T.s_max - 0.1 * N * 25.0  # horizon look-ahead at v_max
t_solve, ticks, fails, iters = 0.0, 0, 0, []
s_hist = []
while x[0, 0] < s_end and ticks < 2500:
    t0 = time.perf_counter(); u = m.make_step(x); t_solve += time.perf_counter() - t0
    fails += int(m.status[0] != 0); iters.append(int(m.iters[0]))
    x = m.plant_step(x, u); ticks += 1; s_hist.append(float(x[0, 0]))
    if ticks % 100 == 0: print(f"tick {ticks}: s = {x[0,0]:7.1f} m, vx = {x[0,3]:5.2f} m/s, n = {x[0,1]:5.2f}, status fails so far {fails}, mean iters {np.mean(iters[-100:]):.1f}", flush=True)
print(f"N = {N}: {ticks} ticks, s = {x[0,0]:.1f} of {T.s_max:.1f} m, simulated {0.1*ticks:.1f} s, solve wall {t_solve:.2f} s -> real-time factor {0.1*ticks/t_solve:.1f}, "
      f"non-converged ticks {fails}, iterations mean {np.mean(iters):.1f} max {max(iters)}")
