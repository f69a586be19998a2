"""Poll interval, re-pack threshold and second-phase line-search width are scheduling only: same bits."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N, B = 40, 3000
x0 = ltompc.sample_x0(T, B, seed=77)
def run(poll, env):
    for k in ("LTOMPC_PACK_NUM", "LTOMPC_LSW"): os.environ.pop(k, None)
    os.environ.update(env)
    o = ltompc.default_options(); o.latency_mode, o.max_iter = 2, 130
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_poll_every(poll); m.set_initial_guess(x0)
    x, out = x0.copy(), []
    for t in range(3):
        u = m.make_step(x); s = m.stats(); out.append((u.copy(), s["status"].copy(), s["iters"].copy())); x = m.plant_step(x, u, 50)
    m.close(); return out
ref = run(4, {})
print("statuses tick 0:", np.bincount(ref[0][1], minlength=6).tolist())
for poll, env in ((1, {}), (2, {}), (3, {}), (7, {}), (16, {}), (4, {"LTOMPC_PACK_NUM": "2"}), (4, {"LTOMPC_PACK_NUM": "7"}), (4, {"LTOMPC_LSW": "64"}), (4, {"LTOMPC_LSW": "2048"})):
    got = run(poll, env)
    print(f"poll_every {poll} {env}:", "same" if all(np.array_equal(got[t][q], ref[t][q]) for t in range(3) for q in range(3)) else "DIFF")
