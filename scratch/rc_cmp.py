"""Riccati output buffer of k_riccati1 against k_riccati8 after one iteration (bit-identical by design)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 4, 10
x0 = ltompc.sample_x0(T, B, seed=3)
def run():
    o = ltompc.default_options(); o.max_iter, o.latency_mode = 1, 2
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0); m.make_step(x0)
    rc = m.debug_fetch(1).copy(); dx = m.debug_fetch(5).copy(); m.close()
    return rc, dx
os.environ["LTOMPC_RIC1"] = "512"; a, da = run()
os.environ["LTOMPC_RIC1"] = "0"; b, db = run()
Bp = 64
A = a.reshape(N + 1, Bp // 8, 82, 8); Bb = b.reshape(N + 1, Bp // 8, 82, 8)
d = np.argwhere(A != Bb)
print("differences:", len(d), "dX equal:", np.array_equal(da, db))
names = [(0, "K"), (16, "Kv"), (20, "kff"), (22, "P"), (58, "Pxv"), (74, "pp")]
for k, grp, f, l in d[:40]:
    nm = [n for o_, n in names if o_ <= f][-1]
    print(f"  stage {k} instance {grp*8+l} field {f} ({nm}+{f - [o_ for o_, n in names if o_ <= f][-1]}): ric1 {A[k,grp,f,l]:.6g} ric8 {Bb[k,grp,f,l]:.6g}")
