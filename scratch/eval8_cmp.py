"""Compare the wave-cooperative evaluation kernels (k_eval8 / k_expand8) with the thread-per-slot ones after MI iterations."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, MI = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x0 = ltompc.sample_x0(T, max(B, 2))[:B]
names = {0: "QP", 2: "RS", 3: "SP", 4: "LS", 7: "dC", 8: "dT", 9: "dNU", 10: "nL1", 11: "nL2", 12: "st"}
def run(mode):
    os.environ["LTOMPC_EVAL"] = mode
    o = ltompc.default_options(); o.max_iter = MI
    m = ltompc.BatchedMPC(T, N, B, options=o)
    m.set_initial_guess(x0); u0 = m.make_step(x0)
    out = {w: m.debug_fetch(w).copy() for w in names}
    out["u0"] = u0; out["it"] = m.iters.copy(); out["status"] = m.status.copy()
    m.close(); return out
a, b = run("slot"), run("wave")
print("iters", a["it"][:8], b["it"][:8], "status", a["status"][:8], b["status"][:8])
for w, nm in list(names.items()) + [("u0", "u0")]:
    x, y = a[w], b[w]
    fin = np.isfinite(x) & np.isfinite(y)
    d = np.abs(x - y)[fin]
    scale = np.maximum(np.abs(x), np.abs(y))[fin]
    rel = d / np.maximum(scale, 1e-3)
    print(f"{nm:4s} size {x.size:7d} nonfinite-mismatch {int((np.isfinite(x) != np.isfinite(y)).sum()):4d} max abs {d.max() if d.size else 0:.3e} max rel {rel.max() if rel.size else 0:.3e}  at {int(np.argmax(rel)) if rel.size else -1}")
