import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")); import oracle as O
T = ltompc.build_tables(); O.build(); orc = O.Oracle(T.packed()); orc.o.max_iter = 300
B, N = 256, 40
x0 = ltompc.sample_x0(T, B, seed=3)
o = ltompc.default_options(); o.max_iter = 300
for mode in (2, 1):
    o.latency_mode = mode
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    x, ref, uprev = x0.copy(), None, np.zeros((B, 2))
    for tick in range(3):
        u0 = m.make_step(x)
        ref = orc.solve(x, N, uprev=uprev, warm=ref, nthreads=16, prev_status=None if ref is None else ref["status"])
        both = (m.status == 0) & (ref["status"] == 0)
        d = np.abs(u0 - ref["u0"])[both].max(axis=1)
        print(f"latency_mode {mode} tick {tick}: both solved {both.sum()}/{B}, |u0 - oracle| median {np.median(d):.1e} p99 {np.percentile(d, 99):.1e} max {d.max():.1e}; "
              f"iteration counts equal {np.mean((m.iters == ref['iters'])[both]):.3f}, within 1: {np.mean((np.abs(m.iters - ref['iters']) <= 1)[both]):.3f}")
        x, uprev = m.plant_step(x, ref["u0"]), ref["u0"]
    m.close()
