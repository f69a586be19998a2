import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); import ltompc
sys.path.insert(0, "/root/repo/oracle"); import oracle as O
T = ltompc.build_tables(); O.build()
def load(name):
    d = np.load("/root/repo/gpurun_out/%s.npz" % name)
    return d, {k: d[k] for k in ("X", "C", "U", "L1", "L2")}
sets = {n: load(n) for n in ("slow", "gen")}
base = {}
for ls in (8, 4, 3, 2):
    for si in (15, 10, 6):
        row = f"max_ls_fail={ls} stall_iter={si:2d}:"
        for n, (d, warm) in sets.items():
            orc = O.Oracle(T.packed()); orc.o.max_iter = 150; orc.o.max_ls_fail = ls; orc.o.stall_iter = si
            r = orc.solve(d["x0"], 40, uprev=d["uprev"], warm=warm, nthreads=8, prev_status=d["prev_status"])
            it, st = r["iters"], r["status"]
            if (ls, si) == (8, 15): base[n] = st.copy()
            lost = int(((base[n] == 0) & (st != 0)).sum())
            row += f"  [{n}] solved {int((st == 0).sum()):4d} lost {lost:2d} max-iters {it.max():3d} p99 {np.percentile(it, 99):5.1f} failed-iters-mean {it[st != 0].mean() if (st != 0).any() else 0:5.1f} sum-iters-over-40 {int(np.maximum(it - 40, 0).sum()):5d}"
        print(row, flush=True)
