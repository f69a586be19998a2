"""Configurations off the beaten path (batch sizes that are not multiples of 64, short / long horizons, option
combinations) against the oracle on a sample of the instances: cold start + 2 warm ticks each."""
import sys, os, itertools, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
from oracle import oracle as orc
T = ltompc.build_tables()
rng = np.random.default_rng(1)
cases = [
    dict(B=1, N=2), dict(B=3, N=3), dict(B=65, N=5, latency_mode=2), dict(B=65, N=5, latency_mode=1),
    dict(B=700, N=10), dict(B=1500, N=20, soft_rho=50.0), dict(B=1500, N=20, periodic_tables=1, soft_rho=100.0),
    dict(B=2000, N=12, warm_shift=1, mu_init_warm=1e-3), dict(B=1111, N=33, warm_reset_on_fail=0),
    dict(B=520, N=60), dict(B=900, N=8, n_linesearch=1), dict(B=900, N=8, n_linesearch=4, stall_iter=5),
    dict(B=4100, N=10, max_iter=60),
    # round 3: the recovery steps and their options
    dict(B=1300, N=40), dict(B=800, N=40, soft_rho=1e5), dict(B=600, N=20, resto_rho=1e4, resto_rho_max=1e6, resto_rho_factor=10.0),
    dict(B=900, N=30, infeasible_sticky=0, resto_shift_retry=0), dict(B=900, N=30, dual_inf_max=0.0, max_mu_stay=30),
    dict(B=700, N=25, node0_check=0, resto_sticky=2), dict(B=513, N=40, warm_shift=1, mu_init_warm=1e-3, warm_fallback_iter=6),
]
bad = 0
TICKS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for c in cases:
    c = dict(c); B, N = c.pop("B"), c.pop("N")
    o, oo = ltompc.default_options(), orc.default_options()
    o.max_iter = oo.max_iter = 200
    for k, v in c.items():
        setattr(o, k, v)
        if k != "latency_mode": setattr(oo, k, v)
    x0 = ltompc.sample_x0(T, B, seed=int(rng.integers(1 << 30)))
    if c.get("periodic_tables"): x0[: B // 2, 0] += 700.0   # into the second lap
    sel = np.sort(rng.choice(B, size=min(B, 48), replace=False))
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    O = orc.Oracle(T.packed(), options=oo)
    x, ref, up = x0.copy(), None, np.zeros((len(sel), 2))
    line = f"B={B:5d} N={N:3d} {c}:"
    for tick in range(TICKS):
        u0 = m.make_step(x)
        ref = O.solve(x[sel], N, uprev=up, warm=ref, nthreads=8, prev_status=None if ref is None else ref["status"])
        both = (m.status[sel] == 0) & (ref["status"] == 0)
        same_status = (m.status[sel] == ref["status"]).mean()
        err = np.abs(u0[sel] - ref["u0"])[both].max() if both.any() else 0.0
        iters_close = (np.abs(m.iters[sel] - ref["iters"])[both] <= 2).mean() if both.any() else 1.0
        flag = "" if (err < 1e-5 and same_status > 0.85 and iters_close > 0.85 and np.isfinite(u0).all()) else "  <-- CHECK"
        bad += bool(flag)
        line += f" [t{tick}: ok {both.mean():.2f} same-status {same_status:.2f} err {err:.1e} iters~ {iters_close:.2f}{flag}]"
        # continue from the GPU's own controls for all instances; the oracle follows the same states for its sample
        xn = m.plant_step(x, u0, 100)
        up = u0[sel].copy()
        # the oracle's warm start is its own previous solution of the same sample
        x = xn
    m.close()
    print(line, flush=True)
print("cases to check:", bad)
