import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
from oracle import oracle as orc
T = ltompc.TrackTables.load_npz("tests/golden/tables_buckmore_mx5_curvature.npz")
O = orc.Oracle(T.packed())
B, N = 64, 20
x0 = ltompc.sample_x0(T, B)
m = ltompc.BatchedMPC(T, N, B); m.set_initial_guess(x0)
u0 = m.make_step(x0); st = m.stats()
r = O.solve(x0, N, nthreads=8)
d = np.abs(u0 - r["u0"]).max(axis=1)
idx = np.argsort(-d)[:8]
print("worst", idx, d[idx]); print("gpu iters", st["iters"][idx], "orc iters", r["iters"][idx], "status", st["status"][idx], r["status"][idx])
print("kkt gpu", st["kkt"][idx], "orc", r["kkt"][idx]); print("obj diff", (st["obj"]-r["obj"])[idx])
print("iters equal frac", (st["iters"]==r["iters"]).mean())
m.close()
# timing
for B, N in ((1, 40), (1024, 40), (8192, 40)):
    x0 = ltompc.sample_x0(T, B) if B > 1 else ltompc.X0_REFERENCE[None]
    m = ltompc.BatchedMPC(T, N, B); m.set_initial_guess(x0)
    t0 = time.time(); u0 = m.make_step(x0); t1 = time.time()
    st = m.stats(); its = st["iters"]
    print(f"B={B} N={N} cold: {t1-t0:.3f}s  iters mean {its.mean():.1f} p90 {np.percentile(its,90):.0f} max {its.max()} status {np.bincount(st['status'], minlength=5)}  ip_iterations {m.timing()['ip_iterations']}")
    for tick in range(2):
        x0 = m.plant_step(x0, u0)
        m.set_profiling(tick == 1)
        t0 = time.time(); u0 = m.make_step(x0); t1 = time.time()
        st = m.stats(); its = st["iters"]
        print(f"   warm{tick}: {t1-t0:.3f}s  iters mean {its.mean():.1f} p90 {np.percentile(its,90):.0f} max {its.max()} status {np.bincount(st['status'], minlength=5)} ip_iterations {m.timing()['ip_iterations']}")
    tm = m.timing(); print("   profile ms:", {k: round(v,3) for k,v in tm["ms"].items()}, tm["launches_by_kernel"])
    m.close()
