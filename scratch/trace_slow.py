import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); import ltompc
sys.path.insert(0, "/root/repo/oracle"); import oracle as O
T = ltompc.build_tables(); O.build(); orc = O.Oracle(T.packed()); orc.o.max_iter = 150
d = np.load("/root/repo/gpurun_out/slow.npz")
j = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "warm"
warm = {k: d[k][j:j+1] for k in ("X", "C", "U", "L1", "L2")}
print("idx", d["idx"][j], "gpu iters", d["iters"][j], "status", d["status"][j], "prev", d["prev_status"][j], "x0", np.round(d["x0"][j], 3), file=sys.stderr)
r = orc.solve(d["x0"][j:j+1], 40, uprev=d["uprev"][j:j+1], warm=None if mode == "cold" else warm)
print("oracle: status", r["status"], "iters", r["iters"], "kkt", r["kkt"], "u0", r["u0"], "gpu u0", d["u0"][j], file=sys.stderr)
