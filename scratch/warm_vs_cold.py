import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); import ltompc
sys.path.insert(0, "/root/repo/oracle"); import oracle as O
T = ltompc.build_tables(); O.build(); orc = O.Oracle(T.packed()); orc.o.max_iter = 150
d = np.load("/root/repo/gpurun_out/slow.npz")
warm = {k: d[k] for k in ("X", "C", "U", "L1", "L2")}
rw = orc.solve(d["x0"], 40, uprev=d["uprev"], warm=warm, nthreads=8)
rc = orc.solve(d["x0"], 40, uprev=d["uprev"], warm=None, nthreads=8)
print("idx   prev gpu_it gpu_st | warm_it st | cold_it st")
for j in range(len(d["idx"])):
    print(f"{d['idx'][j]:5d} {d['prev_status'][j]:4d} {d['iters'][j]:6d} {d['status'][j]:6d} | {rw['iters'][j]:6d} {rw['status'][j]:3d} | {rc['iters'][j]:6d} {rc['status'][j]:3d}  du0 {np.abs(rw['u0'][j]-rc['u0'][j]).max():.1e}")
w2 = dict(warm); w2["L1"] = np.zeros_like(warm["L1"]); w2["L2"] = np.zeros_like(warm["L2"])
r2 = orc.solve(d["x0"], 40, uprev=d["uprev"], warm=w2, nthreads=8)
print("primal-only warm start: iters", r2["iters"].tolist()); print("status", r2["status"].tolist())
sel = d["prev_status"] != 0
print("prev failed: warm iters mean", rw["iters"][sel].mean(), "solved", (rw["status"][sel] == 0).sum(), "| cold", rc["iters"][sel].mean(), (rc["status"][sel] == 0).sum(), "| primal-only", r2["iters"][sel].mean(), (r2["status"][sel] == 0).sum(), "of", sel.sum())
sel = ~sel
print("prev ok:     warm iters mean", rw["iters"][sel].mean(), "solved", (rw["status"][sel] == 0).sum(), "| cold", rc["iters"][sel].mean(), (rc["status"][sel] == 0).sum(), "| primal-only", r2["iters"][sel].mean(), (r2["status"][sel] == 0).sum(), "of", sel.sum())
