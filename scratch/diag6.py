"""Save the warm-start data of the slowest instances of a warm closed-loop tick (for an oracle trace on the CPU)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 150
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
for tick in range(3):
    x0 = m.plant_step(x0, u0)
    if tick == 2:
        warm = m.iterate(); uprev = u0.copy(); prev_status = m.stats()["status"].copy()
    u0 = m.make_step(x0)
st = m.stats()
it = st["iters"]
print("status", np.bincount(st["status"], minlength=5), "pct", np.percentile(it, [50, 90, 99, 99.9, 100]))
for lo, hi in ((0, 30), (30, 40), (40, 50), (50, 70), (70, 100), (100, 151)):
    sel = (it >= lo) & (it < hi)
    print(f"iters in [{lo},{hi}): n={sel.sum()} status {np.bincount(st['status'][sel], minlength=5)} prev-tick status {np.bincount(prev_status[sel], minlength=5)}")
slow = np.argsort(-it)[:48]
gen = np.arange(1024)
np.savez("gpurun_out/gen.npz", idx=gen, x0=x0[gen], uprev=uprev[gen], iters=it[gen], status=st["status"][gen], prev_status=prev_status[gen], u0=u0[gen], **{k: v[gen] for k, v in warm.items() if k in ("X","C","U","L1","L2")})
print("slowest", slow.tolist(), it[slow].tolist(), st["status"][slow].tolist())
np.savez("gpurun_out/slow.npz", idx=slow, x0=x0[slow], uprev=uprev[slow], iters=it[slow], status=st["status"][slow], prev_status=prev_status[slow],
         u0=u0[slow], **{k: v[slow] for k, v in warm.items()})
