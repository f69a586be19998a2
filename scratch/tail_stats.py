"""Who makes the tail of a tick: per-instance iteration counts by status / restoration, over the ticks of the bench workload."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, K = 8192, 40, int(sys.argv[1]) if len(sys.argv) > 1 else 25
o = ltompc.default_options()
for kv in os.environ.get('OPTS', '').split(','):
    if kv:
        k, v = kv.split('='); setattr(o, k, type(getattr(o, k))(float(v)))
m = ltompc.BatchedMPC(T, N, B, options=o)
x = ltompc.sample_x0(T, B)
m.set_initial_guess(x)
prev = None
for t in range(K):
    u = m.make_step(x)
    s = m.stats()
    it, st, nr = s["iters"], s["status"], s["n_resto"]
    if t >= K - 6 or t < 2:
        print(f"tick {t}: launched {m.timing()['ip_iterations']} status {np.bincount(st, minlength=6)} resto {int((nr>0).sum())}")
        for name, sel in (("solved, no resto", (st == 0) & (nr == 0)), ("solved via resto", (st == 0) & (nr > 0)), ("infeasible", st == 5), ("other", ~np.isin(st, (0, 5)))):
            if sel.any():
                q = it[sel]
                print(f"    {name:18s} n {sel.sum():5d} iters mean {q.mean():6.1f} p50 {np.percentile(q,50):4.0f} p90 {np.percentile(q,90):4.0f} p99 {np.percentile(q,99):4.0f} max {q.max():4d}  n_reg mean {s['n_reg'][sel].mean():.1f}")
        top = np.argsort(-it)[:8]
        print("    slowest:", [(int(b), int(it[b]), int(st[b]), int(nr[b]), int(s['n_reg'][b]), int(s['n_lsfail'][b])) for b in top])
        if prev is not None:
            again = (st == 5) & (prev == 5)
            print(f"    infeasible now {int((st==5).sum())}, of which infeasible last tick {int(again.sum())}; viol p50 {np.median(s['viol'][st==5]) if (st==5).any() else 0:.2e} max {s['viol'][st==5].max() if (st==5).any() else 0:.2e}")
    prev = st.copy()
    x = m.plant_step(x, u, 100)
