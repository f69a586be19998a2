"""k_tail (LTOMPC_TAIL) against the launch-per-phase path: controls, statuses, iteration counts bit for bit.
usage: python scratch/tail_cmp2.py B N ticks out.npz   (run once per LTOMPC_TAIL setting, then compare with `cmp a.npz b.npz`)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltompc
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        print(k, "identical" if same else f"DIFFERENT: {int((a[k] != b[k]).sum())} entries, max |diff| {np.abs(a[k].astype(float) - b[k].astype(float)).max():.3e}")
    sys.exit(0)
B, N, ticks = (int(v) for v in sys.argv[1:4])
T = ltompc.build_tables()
o = ltompc.default_options(); o.latency_mode = 2
m = ltompc.BatchedMPC(T, N, B, options=o)
x = ltompc.sample_x0(T, B, seed=3)
m.set_initial_guess(x)
U, S, I, E = [], [], [], []
t0 = time.perf_counter()
for t in range(ticks):
    u = m.make_step(x)
    s = m.stats()
    U.append(u.copy()), S.append(s["status"].copy()), I.append(s["iters"].copy()), E.append(s["kkt"].copy())
    print(f"tick {t}: status {np.bincount(s['status'], minlength=6).tolist()} iters max {s['iters'].max()} launched {m.timing()['ip_iterations']}", flush=True)
    x = m.plant_step(x, u, 100)
print(f"{(time.perf_counter() - t0) * 1e3:.1f} ms")
np.savez(sys.argv[4], U=np.array(U), S=np.array(S), I=np.array(I), E=np.array(E))
