"""GPU diagnostic: the benchmark's closed loop (8192 sampled states, N = 40) with per-tick statistics by recovery path, and a dump
of the slowest instances' problems (x0, u_prev, warm start) for a replay in the oracle.
usage: python scratch/slow_dump.py [ticks] [threshold_iters] [B] [key=value options]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltompc
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 25
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
opts = ltompc.default_options()
for a in sys.argv[4:]:
    k, v = a.split("="); setattr(opts, k, type(getattr(opts, k))(float(v)))
N = 40
ltompc.build_library()
tables = ltompc.build_tables()
mpc = ltompc.BatchedMPC(tables, n_horizon=N, batch=B, options=opts)
x = ltompc.sample_x0(tables, B, seed=ltompc.scenarios.SEED)
mpc.set_initial_guess(x)
up = np.zeros((B, 2))
os.makedirs("gpurun_out", exist_ok=True)
dumped = 0
for t in range(ticks):
    warm = mpc.iterate() if t > 0 else None
    u0 = mpc.make_step(x)
    s = mpc.stats()
    st, it = s["status"], s["iters"]
    node0 = (st == 5) & (s["status_solver"] != 5)
    esc = s["n_resto"] >= 2
    tm = mpc.timing()
    print(f"tick {t}: status {np.bincount(st, minlength=6).tolist()} launched {tm['ip_iterations']} | iters mean {it.mean():.1f} p99 {np.percentile(it, 99):.0f} max {it.max()}"
          f" | node0-infeasible {int(node0.sum())} solver-infeasible {int((s['status_solver'] == 5).sum())} | shift {int((s['n_shift'] > 0).sum())} (solved w/o resto {int(((s['n_shift'] > 0) & (s['n_resto'] == 0) & (s['status_solver'] <= 1)).sum())})"
          f" resto {int((s['n_resto'] > 0).sum())} escalated {int(esc.sum())} (solved {int((esc & (s['status_solver'] <= 1)).sum())}) | iters of: shift-only {it[(s['n_shift'] > 0) & (s['n_resto'] == 0)].mean() if ((s['n_shift'] > 0) & (s['n_resto'] == 0)).any() else 0:.0f}"
          f" resto1 {it[s['n_resto'] == 1].mean() if (s['n_resto'] == 1).any() else 0:.0f} escalated {it[esc].mean() if esc.any() else 0:.0f} | >100: {int((it > 100).sum())} >200: {int((it > 200).sum())}", flush=True)
    slow = np.nonzero(it > thr)[0]
    for b in slow[:6]:
        if warm is None or dumped >= 24: break
        np.savez(f"gpurun_out/slow_t{t}_b{b}.npz", x0=x[b], up=up[b], status=st[b], iters=it[b], n_resto=s["n_resto"][b], n_shift=s["n_shift"][b],
                 n_reg=s["n_reg"][b], n_lsfail=s["n_lsfail"][b], viol=s["viol"][b], **{k: warm[k][b] for k in ("X", "C", "U", "L1", "L2")}, prev_status=prev_st[b])
        dumped += 1
        print(f"    dumped instance {b}: status {st[b]} iters {it[b]} n_resto {s['n_resto'][b]} n_shift {s['n_shift'][b]} n_reg {s['n_reg'][b]} n_lsfail {s['n_lsfail'][b]} viol {s['viol'][b]:.2e}")
    prev_st = s["status_solver"].copy()
    up = u0
    x = mpc.plant_step(x, u0, 100)
