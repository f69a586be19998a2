"""Bit-identity of an instance's result across kernel paths and batch compositions, for several option sets
(3 ticks each; u0, status, iterations).  Paths: narrow kernels off, in-launch sweep retries off, compaction / packing off,
a sub-batch, a permuted batch."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
ENV = ("LTOMPC_RIC1", "LTOMPC_STEP1", "LTOMPC_PACK", "LTOMPC_COMPACT", "LTOMPC_SWEEPS_W", "LTOMPC_RIC1Q")
def run(N, B, x0, opts, params, env, ticks=3):
    for k in ENV: os.environ.pop(k, None)
    os.environ.update(env)
    o = ltompc.default_options(); o.latency_mode = 2
    for k, v in opts.items(): setattr(o, k, v)
    p = ltompc.default_params()
    for k, v in params.items(): setattr(p, k, v)
    m = ltompc.BatchedMPC(T, N, B, options=o, params=p); m.set_initial_guess(x0)
    x, out = x0.copy(), []
    for t in range(ticks):
        u = m.make_step(x); s = m.stats(); out.append((u.copy(), s["status"].copy(), s["iters"].copy())); x = m.plant_step(x, u, 100)
    m.close(); return out
cases = [
    (40, 600, dict(max_iter=90), {}),
    (20, 700, dict(max_iter=1000, soft_rho=50.0), {}),
    (20, 700, dict(max_iter=200, resto_sticky=2), {}),
    (12, 900, dict(max_iter=150, n_linesearch=4, stall_iter=5), {}),
    (30, 600, dict(max_iter=150, periodic_tables=1, soft_rho=100.0), {}),
    (20, 600, dict(max_iter=200), dict(ell_penalty=100.0, ell_rho=1.0, ell_D_f=4000.0, ell_D_r=4000.0)),
    (20, 600, dict(max_iter=200), dict(ptv=3000.0)),
    (60, 530, dict(max_iter=100, warm_reset_on_fail=0), {}),
]
bad = 0
for N, B, opts, params in cases:
    x0 = ltompc.sample_x0(T, B, seed=4242 + N)
    if opts.get("periodic_tables"): x0[: B // 2, 0] += 700.0
    ref = run(N, B, x0, opts, params, {})
    line = f"N={N} B={B} {opts} {params}: statuses tick0 {np.bincount(ref[0][1], minlength=6).tolist()} |"
    for name, env in (("RIC1=0,STEP1=0", {"LTOMPC_RIC1": "0", "LTOMPC_STEP1": "0"}), ("RIC1Q=0", {"LTOMPC_RIC1Q": "0"}), ("RIC1Q=512", {"LTOMPC_RIC1Q": "512"}), ("SWEEPS_W=0", {"LTOMPC_SWEEPS_W": "0"}), ("SWEEPS_W=512", {"LTOMPC_SWEEPS_W": "512"}),
                      ("COMPACT=0", {"LTOMPC_COMPACT": "0"}), ("PACK=0", {"LTOMPC_PACK": "0"})):
        got = run(N, B, x0, opts, params, env)
        same = all(np.array_equal(got[t][q], ref[t][q]) for t in range(3) for q in range(3))
        line += f" {name}: {'same' if same else 'DIFF'}"; bad += not same
    # a sub-batch and a permutation of the batch
    sub = np.arange(0, B, 3)
    got = run(N, len(sub), x0[sub], opts, params, {})
    same = all(np.array_equal(got[t][q], ref[t][q][sub]) for t in range(3) for q in range(3))
    line += f" sub-batch: {'same' if same else 'DIFF'}"; bad += not same
    perm = np.random.default_rng(5).permutation(B)
    got = run(N, B, x0[perm], opts, params, {})
    same = all(np.array_equal(got[t][q], ref[t][q][perm]) for t in range(3) for q in range(3))
    line += f" permuted: {'same' if same else 'DIFF'}"; bad += not same
    print(line, flush=True)
print("differences:", bad)
