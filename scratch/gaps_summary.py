"""Gaps between consecutive kernels on the solver queue from a rocprofv3 kernel trace (last third of the run)."""
import csv, sys, collections, numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("ltompc::", "").replace("void ", "").split("(")[0].split("<")[0]) for r in rows)
ev = ev[len(ev) * 2 // 3:]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(ev, ev[1:]):
    gap[(a[2], b[2])].append((b[0] - a[1]) / 1e3)
for e in ev: dur[e[2]].append((e[1] - e[0]) / 1e3)
print("kernel durations (us):", {k: round(float(np.median(v)), 1) for k, v in dur.items() if len(v) > 5})
tot = 0
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  gap {k[0]:>24s} -> {k[1]:<24s} n {len(v):4d} median {np.median(v):7.1f} us, sum {sum(v)/1e3:7.2f} ms")
    tot += sum(v)
print("all gaps %.2f ms of %.2f ms" % (sum(sum(v) for v in gap.values()) / 1e3, (ev[-1][1] - ev[0][0]) / 1e6))
