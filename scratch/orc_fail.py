import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
sys.path.insert(0, "/root/repo/oracle"); import oracle as O
T = ltompc.build_tables()
O.build(); orc = O.Oracle(T.packed())
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
orc.o.max_iter = 150
t = time.time(); r = orc.solve(x0, N, nthreads=8); print("oracle cold", time.time() - t, "s")
st, it = r["status"], r["iters"]
print("status", np.bincount(st, minlength=5), "iters pct", np.percentile(it, [50, 90, 99, 99.9, 100]))
for lo, hi in ((0, 30), (30, 50), (50, 80), (80, 120), (120, 151)):
    sel = (it >= lo) & (it < hi)
    print(f"iters in [{lo},{hi}): n={sel.sum()} status {np.bincount(st[sel], minlength=5)}")
np.save("/tmp/orc_cold_status.npy", np.stack([st, it]))
bad = np.flatnonzero(st != 0)
print("bad idx", bad[:80].tolist())
slow = np.flatnonzero((st == 0) & (it > 45))
print("slow-but-solved idx", slow.tolist(), it[slow].tolist())
