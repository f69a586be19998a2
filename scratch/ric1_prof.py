"""Section cycle counts of k_riccati1 for one instance (LTOMPC_DBG counters)."""
import sys, os, time, numpy as np
os.environ["LTOMPC_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
x0 = ltompc.X0_REFERENCE[None].copy()
o = ltompc.default_options(); o.latency_mode = int(os.environ.get("MODE", "0"))
m = ltompc.BatchedMPC(T, N, 1, options=o); m.set_initial_guess(x0)
prev = np.zeros(13)
for tick in range(3):
    m.set_profiling(True)
    t0 = time.perf_counter(); u0 = m.make_step(x0); dt = time.perf_counter() - t0
    c = m.debug_fetch(14)[:13].copy(); d = c - prev; prev = c
    tm = m.timing()
    print(f"tick {tick}: {dt*1e3:.2f} ms wall, iters {m.iters[0]}, riccati1 launches {int(d[4])}: cycles/launch head {d[0]/d[4]:.0f} staging {d[1]/d[4]:.0f} backward {d[2]/d[4]:.0f} forward {d[3]/d[4]:.0f}")
    if d[12] > 0:
        print(f"   step1 launches {int(d[12])}: cycles/launch own line search {d[8]/d[12]:.0f} wait for the others {d[9]/d[12]:.0f} pick {d[10]/d[12]:.0f} update {d[11]/d[12]:.0f}")
    print("   event ms per launch:", {k: round(v / max(1, tm['launches_by_kernel'][k]) * 1e3, 1) for k, v in tm["ms"].items()}, flush=True)
    x0 = m.plant_step(x0, u0)
