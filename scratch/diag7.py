"""Closed loop as in bench.py; save the warm-start data of every solve that needed >= 90 iterations."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 150
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
prev_status = m.stats()["status"].copy()
out = dict(x0=[], uprev=[], iters=[], status=[], prev_status=[], tick=[], idx=[], X=[], C=[], U=[], L1=[], L2=[])
for tick in range(7):
    x0 = m.plant_step(x0, u0)
    warm = m.iterate(); uprev = u0.copy()
    t0 = time.time(); u0 = m.make_step(x0); dt = time.time() - t0
    st = m.stats(); it = st["iters"]
    sel = np.flatnonzero(it >= 90)
    print(f"tick {tick}: {dt*1e3:.1f} ms status {np.bincount(st['status'], minlength=5)} pct {np.percentile(it,[50,99,99.9,100])} slow {sel.tolist()} iters {it[sel].tolist()} status {st['status'][sel].tolist()} prev {prev_status[sel].tolist()}", flush=True)
    for j in sel:
        out["x0"].append(x0[j]); out["uprev"].append(uprev[j]); out["iters"].append(it[j]); out["status"].append(st["status"][j])
        out["prev_status"].append(prev_status[j]); out["tick"].append(tick); out["idx"].append(j)
        for k in ("X", "C", "U", "L1", "L2"): out[k].append(warm[k][j])
    prev_status = st["status"].copy()
np.savez("gpurun_out/slow2.npz", **{k: np.array(v) for k, v in out.items()})
