"""Restoration phase (elastic mode) on the GPU against the oracle: stall states, a sampled batch, the reference loop."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
from oracle import oracle as orc
T = ltompc.build_tables()
STALL = {20: ([226.623754, -0.545036120, -0.0112268024, 8.52329373, 0.122918012, 0.161888169, 0.0837443810, 0.371730909], [0.78837973, -0.00572868]),
         40: ([271.551631, -3.15996996e-03, -0.138116402, 9.84560464, 0.483360380, 0.717172238, 0.338696158, -0.336634617], [1.17640462, -0.99999992])}
O = orc.Oracle(T.packed())
for N, (x, up) in STALL.items():
    x = np.array([x]); 
    for mode in (1, 2):
        o = ltompc.default_options(); o.latency_mode = mode
        m = ltompc.BatchedMPC(T, N, 1, options=o)
        m.set_initial_guess(x)
        u = m.make_step(x); s = m.stats()
        r = O.solve(x, N)   # (cold start, uprev 0 on both sides)
        print(f"N={N} mode {mode}: gpu status {s['status'][0]} it {s['iters'][0]} resto {s['n_resto'][0]} viol {s['viol'][0]:.3e} u0 {u[0]} | oracle status {r['status'][0]} it {r['iters'][0]} viol {r['viol'][0]:.3e} u0 {r['u0'][0]}", flush=True)
        m.close()
B, N = 256, 40
x = ltompc.sample_x0(T, B)
m = ltompc.BatchedMPC(T, N, B)
m.set_initial_guess(x)
ref, up = None, np.zeros((B, 2))
for t in range(5):
    u = m.make_step(x); s = m.stats()
    ref = O.solve(x, N, up, ref, nthreads=8, prev_status=None if ref is None else ref["status"])
    both = (s["status"] == 0) & (ref["status"] == 0)
    print(f"tick {t}: gpu {np.bincount(s['status'], minlength=6)} oracle {np.bincount(ref['status'], minlength=6)} same status {(s['status']==ref['status']).mean():.4f} "
          f"|u0 diff| max {np.abs(u-ref['u0'])[both].max():.2e} iters equal(<=2) {(np.abs(s['iters']-ref['iters'])[both]<=2).mean():.3f} resto gpu {int(s['n_resto'].sum())} oracle {int(ref['n_resto'].sum())}", flush=True)
    x, up = O.plant_step(x, ref["u0"]), ref["u0"]
m.close()
# the reference loop: N = 10, 500 ticks, hard constraints (src/mpc.py:104-153)
m = ltompc.BatchedMPC(T, 10, 1)
x = ltompc.X0_REFERENCE[None].copy()
m.set_initial_guess(x)
hist, bad = {}, []
t0 = time.time()
for t in range(500):
    u = m.make_step(x)
    st = int(m.status[0]); hist[st] = hist.get(st, 0) + 1
    if st not in (0, 1): bad.append((t + 1, st, float(m.stats()["viol"][0])))
    x = m.plant_step(x, u, 100)
print("reference loop N=10, 500 ticks:", hist, "s =", x[0, 0], "wall", time.time() - t0)
print("non-converged ticks (tick, status, violation):", bad)
