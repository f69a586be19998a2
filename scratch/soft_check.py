"""soft_rho on the GPU vs the oracle on a small batch (prints what the parity test asserts), then the full lap."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
from oracle import oracle as orc
T = ltompc.build_tables()
B, N = 48, 20
x0 = ltompc.sample_x0(T, B, seed=5)
oo = orc.default_options(); oo.soft_rho = 100.0
O = orc.Oracle(T.packed(), options=oo)
for mode in (2, 1):
    o = ltompc.default_options(); o.soft_rho, o.latency_mode = 100.0, mode
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    x, ref, up = x0.copy(), None, np.zeros((B, 2))
    for tick in range(3):
        u0 = m.make_step(x)
        ref = O.solve(x, N, uprev=up, warm=ref, nthreads=8, prev_status=None if ref is None else ref["status"])
        both = (m.status == 0) & (ref["status"] == 0)
        print(f"mode {mode} tick {tick}: both {both.mean():.3f} gpu ok {(m.status==0).mean():.3f} orc ok {(ref['status']==0).mean():.3f} "
              f"max|du0| {np.abs(u0-ref['u0'])[both].max():.2e} iters equal {(m.iters==ref['iters'])[both].mean():.3f} "
              f"obj err {np.abs(m.stats()['obj']-ref['obj'])[both].max():.2e} mean it {m.iters.mean():.1f}", flush=True)
        x, up = O.plant_step(x, ref["u0"]), ref["u0"]
    m.close()
