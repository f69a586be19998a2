import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); import ltompc
sys.path.insert(0, "/root/repo/oracle"); import oracle as O
T = ltompc.build_tables(); O.build()
for n in ("slow", "gen", "slow2"):
    d = np.load("/root/repo/gpurun_out/%s.npz" % n)
    warm = {k: d[k] for k in ("X", "C", "U", "L1", "L2")}
    orc = O.Oracle(T.packed()); orc.o.max_iter = 150
    for k, v in [a.split("=") for a in sys.argv[1:]]: setattr(orc.o, k, type(getattr(orc.o, k))(float(v)))
    r = orc.solve(d["x0"], 40, uprev=d["uprev"], warm=warm, nthreads=8, prev_status=d["prev_status"])
    it, st = r["iters"], r["status"]
    print(f"[{n:5s}] n={len(it):4d} solved {int((st == 0).sum()):4d} status {np.bincount(st, minlength=5)} iters mean {it.mean():6.2f} p50 {np.percentile(it,50):.0f} p99 {np.percentile(it,99):6.1f} max {it.max():3d} sum-over-40 {int(np.maximum(it-40,0).sum())} nlsfail {int(r['n_lsfail'].sum())} nreg {int(r['n_reg'].sum())}")
