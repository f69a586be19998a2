"""Compare the Riccati output (RC, dX, dU) of the wide path and of the tail kernel after the same number of iterations."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, MI = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x0 = ltompc.sample_x0(T, max(B, 2))[:B]
def run(tail):
    os.environ["LTOMPC_RIC1"] = str(tail); os.environ["LTOMPC_STEP1"] = str(tail)
    o = ltompc.default_options(); o.max_iter = MI
    m = ltompc.BatchedMPC(T, N, B, options=o)
    m.set_initial_guess(x0); u0 = m.make_step(x0)
    Bp = m.debug_fetch(12).size // 8 // 64 if False else None
    out = {w: m.debug_fetch(w).copy() for w in (0, 1, 5, 6, 12, 13)}
    out["u0"] = u0; out["it"] = m.iters.copy()
    m.close(); return out
a, b = run(0), run(1024)
print("iters", a["it"][:8], b["it"][:8])
for w in (0, 1, 5, 6, 12, 13, "u0"):
    d = a[w] != b[w]
    print(w, "size", a[w].size, "differ", int(d.sum()), "max abs diff", float(np.nanmax(np.abs(a[w].astype(float) - b[w]))) if d.any() else 0.0)
w = 1
d = np.flatnonzero(a[w] != b[w])
if d.size:
    Bp = a[12].size // 8  # ST planes... unknown count; print raw indices instead
    print("first differing flat indices in RC:", d[:20], "of", a[w].size)
    print("values wide:", a[w][d[:8]], "tail:", b[w][d[:8]])

si_a, si_b = a[13].reshape(-1, a[13].size // 13 if a[13].size % 13 == 0 else 1), None
print("SI wide:", a[13].reshape(13, -1)[:, :8].tolist() if a[13].size % 13 == 0 else a[13][:40])
print("SI tail:", b[13].reshape(13, -1)[:, :8].tolist() if b[13].size % 13 == 0 else b[13][:40])
