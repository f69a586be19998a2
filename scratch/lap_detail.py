import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N = 60
x = ltompc.X0_REFERENCE[None].copy()
o = ltompc.default_options(); o.max_iter = 500
m = ltompc.BatchedMPC(T, N, 1, options=o); m.set_initial_guess(x)
for tick in range(470):
    u = m.make_step(x)
    st = m.stats()
    if tick >= 395 and tick % 2 == 0 or st["status"][0] != 0:
        s = x[0, 0]
        nl, nr = np.interp(s, T.s_arc, T.n_left), np.interp(s, T.s_arc, T.n_right)
        vr = np.interp(s, T.s_arc, T.v_ref); kap = np.interp(s, T.s_kappa, T.kappa)
        print(f"tick {tick}: s {s:6.1f} n {x[0,1]:6.2f} mu {x[0,2]:6.3f} vx {x[0,3]:5.2f} vy {x[0,4]:5.2f} r {x[0,5]:5.2f} delta {x[0,6]:5.2f} T {x[0,7]:5.2f} | NL {nl:4.2f} NR {nr:4.2f} 0.6vref {0.6*vr:5.2f} kappa {kap:6.3f} | status {st['status'][0]} iters {st['iters'][0]} kkt {st['kkt'][0]:.1e} u {u[0,0]:6.2f} {u[0,1]:5.2f}")
    x = m.plant_step(x, u)
    if abs(x[0, 1]) > 8: break
