"""Per kernel class and launch width: number of launches and average device time, for one warm tick.  usage: launch_table.py [ticks] [B]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = (int(sys.argv[2]) if len(sys.argv) > 2 else 8192), 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options()
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for tick in range(NT):
    x0 = m.plant_step(x0, u0)
    m.set_profiling(tick == NT - 1)
    t0 = time.perf_counter(); u0 = m.make_step(x0); dt = time.perf_counter() - t0
kind, width, ms = m.launch_log()
names = ["eval", "riccati", "expand", "linesearch", "pick", "update", "riccati1", "step1"]
print(f"tick {dt*1e3:.1f} ms, sum of kernel time {ms.sum():.1f} ms, launches {len(ms)}")
bins = [(B, B), (B // 2 + 1, B - 1), (513, B // 2), (65, 512), (17, 64), (1, 16)]
print("kernel       " + "".join(f"{f'{lo}..{hi}':>22s}" for hi, lo in [(b[1], b[0]) for b in bins]))
for q, nm in enumerate(names):
    row = f"{nm:12s} "
    for lo, hi in bins:
        sel = (kind == q) & (width >= lo) & (width <= hi)
        row += f"{int(sel.sum()):5d} x {ms[sel].mean()*1e3 if sel.any() else 0:7.1f}us ={ms[sel].sum():6.1f}"
    print(row)
tot = [ms[(width >= lo) & (width <= hi)].sum() for lo, hi in bins]
print("total ms     " + "".join(f"{t:22.1f}" for t in tot))
