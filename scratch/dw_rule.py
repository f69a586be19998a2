"""For slow / failing warm solves: how long does the regularisation stay huge, and would a rule on it have false positives?"""
import sys, os, re, subprocess, numpy as np
sets = ("slow", "slow2")
code = r'''
import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); import ltompc
sys.path.insert(0, "/root/repo/oracle"); import oracle as O
T = ltompc.build_tables(); O.build(); orc = O.Oracle(T.packed()); orc.o.max_iter = 150
d = np.load("/root/repo/gpurun_out/%s.npz" % sys.argv[1]); j = int(sys.argv[2])
warm = {k: d[k][j:j+1].copy() for k in ("X", "C", "U", "L1", "L2")}
r = orc.solve(d["x0"][j:j+1], 40, uprev=d["uprev"][j:j+1], warm=warm, prev_status=d["prev_status"][j:j+1])
print("RESULT", int(r["status"][0]), int(r["iters"][0]), file=sys.stderr)
'''
open("/tmp/one.py", "w").write(code)
for name in sets:
    d = np.load("/root/repo/gpurun_out/%s.npz" % name)
    rows = []
    for j in range(len(d["idx"])):
        out = subprocess.run([sys.executable, "/tmp/one.py", name, str(j)], env=dict(os.environ, ORACLE_TRACE="1"), capture_output=True, text=True).stderr
        dws = [float(m) for m in re.findall(r" dw ([0-9.e+-]+) ", out)]
        res = re.search(r"RESULT (\d+) (\d+)", out)
        st, it = int(res.group(1)), int(res.group(2))
        def longest(th):
            best = cur = 0
            for v in dws:
                cur = cur + 1 if v >= th else 0
                best = max(best, cur)
            return best
        first = next((i for i in range(len(dws)) if all(v >= 1e3 for v in dws[i:i + 8]) and len(dws[i:i + 8]) == 8), -1)
        rows.append((st, it, longest(1e2), longest(1e3), longest(1e4), first))
    print(name, "(status, iters, longest run of dw >= 1e2 / 1e3 / 1e4, first iteration of 8 consecutive dw >= 1e3)")
    for r in sorted(rows): print("   ", r)
