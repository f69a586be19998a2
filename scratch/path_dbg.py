import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N, B = 40, 600
x0 = ltompc.sample_x0(T, B, seed=4242 + N)
def run(env, mi=90):
    for k in ("LTOMPC_SWEEPS_W",): os.environ.pop(k, None)
    os.environ.update(env)
    o = ltompc.default_options(); o.latency_mode, o.max_iter = 2, mi
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    u = m.make_step(x0); s = m.stats(); c = m.counters() if hasattr(m, "counters") else None
    it = m.iterate(); m.close(); return u, s, it
ua, sa, ia = run({})
ub, sb, ib = run({"LTOMPC_SWEEPS_W": "512"})
d = np.where((ua != ub).any(1) | (sa["status"] != sb["status"]) | (sa["iters"] != sb["iters"]))[0]
print("differing instances (tick 0):", len(d))
for j in d[:10]:
    print(j, "status", sa["status"][j], sb["status"][j], "iters", sa["iters"][j], sb["iters"][j], "n_reg", sa["n_reg"][j], sb["n_reg"][j], "kkt", sa["kkt"][j], sb["kkt"][j], "|dX|", np.abs(ia["X"][j] - ib["X"][j]).max())
for mi in (88, 89, 90, 91, 92):
    row = []
    for w in ("0", "16", "512"):
        u, s, it = run({"LTOMPC_SWEEPS_W": w}, mi)
        row.append((int(s["status"][311]), int(s["iters"][311]), int(s["n_reg"][311]), float(s["kkt"][311])))
    print("max_iter", mi, "SWEEPS_W 0/16/512:", row)
