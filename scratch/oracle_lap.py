"""Closed-loop lap with the CPU oracle (one instance): does the loop pass the s ~ 405 m chicane?  SOFT_RHO env = options.soft_rho."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
from oracle import oracle as orc
T = ltompc.build_tables()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s_stop = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
o = orc.default_options(); o.max_iter = 300
o.soft_rho = float(os.environ.get('SOFT_RHO', '0'))
O = orc.Oracle(T.packed(), options=o)
x = ltompc.X0_REFERENCE[None].copy()
warm, st, up = None, None, np.zeros((1, 2))
s_end = min(T.s_max - 0.1 * N * 25.0, s_stop)
ticks, fails, iters = 0, 0, []
t0 = time.time()
while x[0, 0] < s_end and ticks < 2500:
    r = O.solve(x, N, up, warm, prev_status=st)
    warm, st, up = r, r["status"], r["u0"]
    fails += int(st[0] not in (0, 1)); iters.append(int(r["iters"][0]))
    x = O.plant_step(x, r["u0"], n_sub=100); ticks += 1
    if abs(x[0,1]) > 5: print('car left the track'); break
    if ticks % 50 == 0 or st[0] not in (0, 1):
        print(f"tick {ticks}: s {x[0,0]:7.1f} vx {x[0,3]:5.2f} n {x[0,1]:6.3f} T {x[0,7]:5.2f} st {st[0]} it {iters[-1]} fails {fails} mean it {np.mean(iters[-50:]):.1f}", flush=True)
print(f"N={N}: {ticks} ticks, s={x[0,0]:.1f} of {T.s_max:.1f}, sim {0.1*ticks:.1f}s, wall {time.time()-t0:.1f}s, fails {fails}, iters mean {np.mean(iters):.1f} max {max(iters)}")
