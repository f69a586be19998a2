"""Who are the instances that do not converge in the bench's closed loop?"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 150
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
for tick in range(4):
    x0 = m.plant_step(x0, u0)
    u0 = m.make_step(x0)
    st = m.stats()
    bad = np.flatnonzero(st["status"] != 0)
    print(f"tick {tick}: status {np.bincount(st['status'], minlength=5)} iters pct {np.percentile(st['iters'],[50,90,99,99.9,100])}")
s_arc, nl, nr = np.asarray(T.s_arc), np.asarray(T.n_left), np.asarray(T.n_right)
X, U = m.prediction()
ine = None
print("   b status iters      kkt       mu  nreg nlsf |       s       n      mu      vx      vy       r   delta       T | NL(s) NR(s) | min margin over horizon")
for b in bad[:40]:
    x = x0[b]
    NLs, NRs = np.interp(x[0], s_arc, nl), np.interp(x[0], s_arc, nr)
    Xb = X[b]
    gl = Xb[:, 1] - 1.5 * np.sin(np.abs(Xb[:, 2])) + 1.15 * np.cos(Xb[:, 2]) - np.interp(Xb[:, 0], s_arc, nl)
    gr = -Xb[:, 1] + 1.5 * np.sin(np.abs(Xb[:, 2])) + 1.15 * np.cos(Xb[:, 2]) - np.interp(Xb[:, 0], s_arc, nr)
    print(f"{b:5d} {st['status'][b]:6d} {st['iters'][b]:5d} {st['kkt'][b]:8.1e} {st['mu'][b]:8.1e} {st['n_reg'][b]:5d} {st['n_lsfail'][b]:4d} | " +
          " ".join(f"{v:7.3f}" for v in x) + f" | {NLs:5.2f} {NRs:5.2f} | gL max {gl[1:].max():7.3f} gR max {gr[1:].max():7.3f} vx min {Xb[:,3].min():6.2f}")
print("n bad", bad.size, "of which |n| > 3:", int((np.abs(x0[bad, 1]) > 3).sum()), " vx<1:", int((x0[bad, 3] < 1).sum()))
good = np.flatnonzero(st["status"] == 0)
print("good: |n| mean", np.abs(x0[good, 1]).mean(), "bad: |n| mean", np.abs(x0[bad, 1]).mean(), " good vx mean", x0[good, 3].mean(), "bad vx mean", x0[bad, 3].mean())
