"""Who keeps the narrow launches busy: iteration counts of converged vs non-converged instances (warm ticks, B = 8192)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B, seed=ltompc.scenarios.SEED)
o = ltompc.default_options(); o.max_iter = 150
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
for tick in range(4):
    x0 = m.plant_step(x0, u0, 100)
    u0 = m.make_step(x0)
    s = m.stats(); it, st = s["iters"], s["status"]
    ok = st == 0
    print(f"tick {tick}: solved {ok.mean():.4f}; statuses {np.bincount(st, minlength=5)}")
    for thr in (30, 40, 60, 80, 100):
        late = it > thr
        print(f"   iterations > {thr:3d}: {late.sum():4d} instances, of which converge later {int((late & ok).sum()):4d}, fail {int((late & ~ok).sum()):4d}; "
              f"instance-iterations beyond {thr}: solved {int((it[late & ok] - thr).sum())}, failed {int((it[late & ~ok] - thr).sum())}")
    h = m.history()
    print("   poll history (it, active, width):", [tuple(r) for r in h[::3]][:16])
