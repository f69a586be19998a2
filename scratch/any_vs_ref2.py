import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 300, 20
x0 = ltompc.sample_x0(T, B, seed=9)
def run():
    o = ltompc.default_options(); o.max_iter = 1
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    m.make_step(x0)
    out = {w: m.debug_fetch(i).copy() for w, i in (("SP", 3), ("dC", 7), ("dT", 8), ("dNU", 9), ("nL1", 10), ("nL2", 11))}
    Bp = m.debug_fetch(12).size // 16
    m.close()
    return out, Bp
os.environ.pop("LTOMPC_BOUNDS", None)
a, Bp = run(); a2, _ = run()
os.environ["LTOMPC_BOUNDS"] = "expand"
b, _ = run(); b2, _ = run()
for k in a:
    A, Bb = a[k].reshape(-1, N, Bp), b[k].reshape(-1, N, Bp)
    d = np.abs(A - Bb)[:, :, :B]
    d[np.isnan(d)] = 1e300
    f, kk, bb = np.where(d > 0)
    print(k, "fields", A.shape[0], "ndiff", len(f), "ref repeatable", np.array_equal(a[k], a2[k], equal_nan=True), "any repeatable", np.array_equal(b[k], b2[k], equal_nan=True))
    if len(f):
        print("   fields that differ:", np.unique(f), "intervals:", np.unique(kk), "n instances", len(np.unique(bb)))
        i = 0
        print("   e.g. field %d k %d b %d: ref %.17g any %.17g" % (f[i], kk[i], bb[i], A[f[i], kk[i], bb[i]], Bb[f[i], kk[i], bb[i]]))
