"""Two-lane rollout from a rocprofv3 kernel trace: per queue, kernels / busy time / passes; gaps of the solver queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"].replace("ltompc::", "").replace("void ", "").split("(")[0].split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r["Grid_Size_X"]), r["Queue_Id"]))
ev.sort()
begins = [i for i, e in enumerate(ev) if e[2] == "k_roll_begin"]
ev = ev[begins[-1]:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print("wall %.1f ms" % ((t1 - t0) / 1e6))
byq = collections.defaultdict(list)
for e in ev: byq[e[4]].append(e)
for q, es in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in es) / 1e6
    names = collections.Counter(e[2] for e in es)
    ks = collections.defaultdict(float)
    for e in es: ks[e[2]] += (e[1] - e[0]) / 1e6
    span = (es[-1][1] - es[0][0]) / 1e6
    print(f"queue {q}: {len(es)} kernels, busy {busy:.1f} ms over a span of {span:.1f} ms; passes (k_roll_finish) {names.get('k_roll_finish', 0)}; top: " +
          ", ".join(f"{k} {v:.0f} ms/{names[k]}" for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:7]))
# lane 1 pass period while lane 0 is at full width
