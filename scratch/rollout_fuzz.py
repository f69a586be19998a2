"""Rollout with free-running instances against the lock-step loop for several option sets (u0, status, iteration logs, final x)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables(); dev = torch.device("cuda", 0)
cases = [
    (20, 400, 6, 50, dict(max_iter=60), {}),
    (40, 300, 5, 100, dict(max_iter=90), {}),
    (10, 700, 8, 25, dict(max_iter=1000, soft_rho=50.0), {}),
    (20, 500, 6, 50, dict(max_iter=300, resto_sticky=2), {}),
    (20, 300, 5, 50, dict(max_iter=200), dict(ell_penalty=100.0, ell_rho=1.0, ell_D_f=4000.0, ell_D_r=4000.0)),
    (30, 300, 5, 50, dict(max_iter=150, periodic_tables=1, soft_rho=100.0), {}),
    (12, 900, 6, 50, dict(max_iter=150, n_linesearch=4, stall_iter=5), {}),
    (20, 64, 6, 50, dict(max_iter=100), {}),
    (20, 7, 6, 50, dict(max_iter=100), {}),
]
bad = 0
for N, B, K, NSUB, opts, params in cases:
    o = ltompc.default_options(); o.latency_mode = 2
    for k, v in opts.items(): setattr(o, k, v)
    p = ltompc.default_params()
    for k, v in params.items(): setattr(p, k, v)
    x0 = ltompc.sample_x0(T, B, seed=99 + N + B)
    if opts.get("periodic_tables"): x0[: B // 2, 0] += 700.0
    a = ltompc.BatchedMPC(T, N, B, options=o, params=p); a.set_initial_guess(x0)
    x, U, S, I = x0.copy(), [], [], []
    for t in range(K):
        u = a.make_step(x); s = a.stats(); U.append(u.copy()), S.append(s["status"].copy()), I.append(s["iters"].copy()); x = a.plant_step(x, u, NSUB)
    U, S, I = np.stack(U, 1), np.stack(S, 1), np.stack(I, 1)
    b = ltompc.BatchedMPC(T, N, B, options=o, params=p)
    xb = torch.from_numpy(x0).to(dev); b.set_initial_guess_dev(xb.data_ptr())
    ul = torch.zeros(B, K, 2, dtype=torch.float64, device=dev); sl = torch.full((B, K), -1, dtype=torch.int32, device=dev); il = torch.zeros(B, K, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    info = b.rollout_dev(xb.data_ptr(), K, NSUB, ul.data_ptr(), sl.data_ptr(), il.data_ptr())
    same = (np.array_equal(U, ul.cpu().numpy()), np.array_equal(S, sl.cpu().numpy()), np.array_equal(I, il.cpu().numpy()), np.array_equal(x, xb.cpu().numpy()))
    bad += not all(same)
    print(f"N={N} B={B} K={K} {opts} {params}: statuses {np.bincount(S.ravel(), minlength=6).tolist()} u/status/iters/x identical: {same} {info}", flush=True)
    a.close(); b.close()
print("differences:", bad)
