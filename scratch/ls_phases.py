"""k_linesearch / k_pick: phase 0 vs phase 1 launch times at full width (alternating launches of the same class)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 150
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
u0 = m.make_step(x0)
for tick in range(3):
    x0 = m.plant_step(x0, u0)
    m.set_profiling(tick == 2)
    u0 = m.make_step(x0)
kind, width, ms = m.launch_log()
for q, nm in ((3, "linesearch"), (4, "pick")):
    sel = np.where((kind == q) & (width == B))[0]
    print(nm, "phase 0: %.1f us, phase 1: %.1f us (%d launches each)" % (ms[sel[0::2]].mean() * 1e3, ms[sel[1::2]].mean() * 1e3, len(sel) // 2))
