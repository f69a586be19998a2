"""Which kernel path makes an instance's bits differ?  B = 520 (wide launches first, narrow ones later), 3 ticks, several N."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B = int(os.environ.get("BB", 520))
def run(N, env, x0):
    for k in ("LTOMPC_RIC1", "LTOMPC_STEP1", "LTOMPC_PACK", "LTOMPC_COMPACT", "LTOMPC_SWEEPS_W"): os.environ.pop(k, None)
    os.environ.update(env)
    o = ltompc.default_options(); o.max_iter, o.latency_mode = 120, 2
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    x, out = x0.copy(), []
    for t in range(3):
        u = m.make_step(x); s = m.stats(); out.append((u.copy(), s["status"].copy(), s["iters"].copy())); x = m.plant_step(x, u, 100)
    m.close(); return out
for N in (40, 60):
    x0 = ltompc.sample_x0(T, B, seed=12345)
    a = run(N, {}, x0)
    for name, env in (("RIC1=0", {"LTOMPC_RIC1": "0"}), ("STEP1=0", {"LTOMPC_STEP1": "0"}), ("SWEEPS_W=0", {"LTOMPC_SWEEPS_W": "0"}), ("PACK=0", {"LTOMPC_PACK": "0"}),
                      ("COMPACT=0", {"LTOMPC_COMPACT": "0"})):
        b = run(N, env, x0)
        line = f"N={N} {name:10s}:"
        for t in range(3):
            same = np.array_equal(a[t][0], b[t][0])
            nd = int((a[t][0] != b[t][0]).any(1).sum())
            line += f" tick {t} {'same' if same else f'DIFF ({nd} instances, iters differ for {int((a[t][2] != b[t][2]).sum())})'}"
            if not same and t == 1:
                j = np.where((a[t][0] != b[t][0]).any(1))[0][:4]
                line += f" e.g. {[(int(q), int(a[t][2][q]), int(b[t][2][q]), int(a[0][1][q]), int(a[0][2][q])) for q in j]}"
        print(line, flush=True)
