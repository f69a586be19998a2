import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file", glob.glob(d + "/**/*", recursive=True)[:10]); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
print("columns", list(rows[0].keys()))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    for c, v in cs.items():
        v2 = sorted(v, reverse=True)
        print(f"{k:28s} {c:12s} n={len(v):5d} max={v2[0]:.0f} top10-mean={sum(v2[:10])/min(10,len(v2)):.0f} mean={sum(v)/len(v):.0f}")
