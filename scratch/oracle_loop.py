"""The benchmark's closed loop in the ORACLE (CPU): per tick the statuses, iteration statistics and the longest solves (iterations +
repeated factorisations), e.g. to compare a variant of the algorithm (ORACLE_DWFB=1 with scratch/dw_feedback_oracle.patch applied).
usage: python scratch/oracle_loop.py B ticks [key=value options of the oracle, e.g. max_soc=4]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module("lap-time-optimization_amd")
from oracle import oracle as orc
orc.build()
B, ticks = int(sys.argv[1]), int(sys.argv[2])
N = 40
tables = pkg.TrackTables.load_npz(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tables_buckmore_mx5_curvature.npz"))
opts = orc.default_options()
for a in sys.argv[3:]:
    k, v = a.split("="); setattr(opts, k, type(getattr(opts, k))(float(v)))
O = orc.Oracle(tables.packed(), options=opts)
x = pkg.sample_x0(tables, 8192)[:B]
warm, st, up = None, None, np.zeros((B, 2))
for tick in range(ticks):
    t0 = time.time()
    r = O.solve(x, N, up, warm, prev_status=st, nthreads=8)
    warm, st, up = r, r["status"], r["u0"]
    ps = r["iters"] + r["n_reg"]
    print(f"tick {tick}: status {np.bincount(st, minlength=6).tolist()} solver {np.bincount(r['status_solver'], minlength=6).tolist()} iters mean {r['iters'].mean():.2f} p99 {np.percentile(r['iters'], 99):.0f} max {r['iters'].max()} | passes max {ps.max()} top5 {sorted(ps)[-5:]} >70: {(ps > 70).sum()} | shift {int((r['n_shift']>0).sum())} resto {int((r['n_resto']>0).sum())} n_reg sum {int(r['n_reg'].sum())} n_soc sum {int(r['n_soc'].sum())} {time.time()-t0:.1f}s", flush=True)
    x = O.plant_step(x, r["u0"], n_sub=100)
