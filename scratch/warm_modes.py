import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); import ltompc
sys.path.insert(0, "/root/repo/oracle"); import oracle as O
T = ltompc.build_tables(); O.build(); orc = O.Oracle(T.packed()); orc.o.max_iter = 150
d = np.load("/root/repo/gpurun_out/%s.npz" % (sys.argv[1] if len(sys.argv) > 1 else "gen"))
warm = {k: d[k] for k in ("X", "C", "U", "L1", "L2")}
def rep(name, r):
    it, st = r["iters"], r["status"]
    print(f"{name:28s} iters mean {it.mean():6.2f} pct50/90/99/max {np.percentile(it,[50,90,99,100])} status {np.bincount(st, minlength=5)}")
rw = orc.solve(d["x0"], 40, uprev=d["uprev"], warm=warm, nthreads=8); rep("warm (primal+dual)", rw)
w2 = dict(warm); w2["L1"] = np.zeros_like(warm["L1"]); w2["L2"] = np.zeros_like(warm["L2"])
r2 = orc.solve(d["x0"], 40, uprev=d["uprev"], warm=w2, nthreads=8); rep("primal-only", r2)
print("both solved:", ((rw["status"] == 0) & (r2["status"] == 0)).sum(), "u0 agree (<1e-6):", (np.abs(rw["u0"] - r2["u0"]).max(1) < 1e-6)[(rw["status"] == 0) & (r2["status"] == 0)].sum())
for ws, mw in ((1, 1e-3), (1, 1e-2), (0, 1e-2)):
    orc.o.warm_shift = ws; orc.o.mu_init_warm = mw
    rep(f"shift={ws} mu_w={mw} keep", orc.solve(d["x0"], 40, uprev=d["uprev"], warm=warm, nthreads=8))
    rep(f"shift={ws} mu_w={mw} primal-only", orc.solve(d["x0"], 40, uprev=d["uprev"], warm=w2, nthreads=8))
print("---- auto: reset duals only when the previous solve failed")
bad = d["prev_status"] != 0
w3 = dict(warm); w3["L1"] = warm["L1"].copy(); w3["L2"] = warm["L2"].copy(); w3["L1"][bad] = 0; w3["L2"][bad] = 0
for ws, mw in ((0, 0.0), (1, 1e-3), (1, 1e-2)):
    orc.o.warm_shift = ws; orc.o.mu_init_warm = mw
    rep(f"shift={ws} mu_w={mw} auto", orc.solve(d["x0"], 40, uprev=d["uprev"], warm=w3, nthreads=8))
