"""Which instances make the tail of a tick?  The benchmark's closed loop on one handle; for the ticks 5 .. T-1 every instance with more than
`thr` passes (iters + regularisation retries) with its statistics, and a histogram by recovery path.
usage: python scratch/tail_table.py [ticks] [thr] [key=value options]"""
import sys, os, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltompc
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 25
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 70
opts = ltompc.default_options()
for a in sys.argv[3:]:
    k, v = a.split("="); setattr(opts, k, type(getattr(opts, k))(float(v)))
B, N = 8192, 40
tables = ltompc.build_tables()
mpc = ltompc.BatchedMPC(tables, n_horizon=N, batch=B, options=opts)
x = ltompc.sample_x0(tables, B, seed=ltompc.scenarios.SEED)
mpc.set_initial_guess(x)
hist = collections.Counter(); rows = []
sum_passes = collections.Counter()
for t in range(ticks):
    u0 = mpc.make_step(x)
    s = mpc.stats()
    it, reg = s["iters"], s["n_reg"]
    passes = it + reg
    if t >= 5:
        for b in np.nonzero(passes > thr)[0]:
            path = ("shift " if s["n_shift"][b] else "") + (f"resto{s['n_resto'][b]} " if s["n_resto"][b] else "") + ("fallback " if s["n_fallback"][b] else "")
            path = path.strip() or "plain"
            key = (path, int(s["status_solver"][b]))
            hist[key] += 1; sum_passes[key] += int(passes[b])
            rows.append((t, b, int(s["status"][b]), int(s["status_solver"][b]), int(it[b]), int(reg[b]), int(s["n_lsfail"][b]), int(s["n_soc"][b]) if "n_soc" in s else -1, path, float(s["g0"][b]), float(s["penalty"][b])))
        top = np.argsort(-passes)[:3]
        print(f"tick {t}: launched {mpc.timing()['ip_iterations']}; top passes " + ", ".join(f"{int(passes[b])} (b {b}, it {int(it[b])} reg {int(reg[b])})" for b in top), flush=True)
    x = mpc.plant_step(x, u0, 100)
print(f"\ninstances with more than {thr} passes, ticks 5..{ticks - 1}: {len(rows)} ({len(rows) / (ticks - 5):.1f} per tick)")
print(f"{'path':28s} {'solver status':>13s} {'count':>6s} {'mean passes':>12s}")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(f"{k[0]:28s} {k[1]:13d} {v:6d} {sum_passes[k] / v:12.1f}")
rows.sort(key=lambda r: -(r[4] + r[5]))
print("\nthe 40 longest: tick b status solver_status iters n_reg n_lsfail n_soc path g0 penalty")
for r in rows[:40]:
    print(" ".join(str(q) if not isinstance(q, float) else f"{q:.2e}" for q in r))
