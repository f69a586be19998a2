"""Per-pass timeline of a rollout from a rocprofv3 kernel trace: a pass starts at every k_roll_mark; passes grouped by the grid of k_eval."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"].replace("ltompc::", "").replace("void ", "").split("(")[0].split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))))
ev.sort()
# split into rollouts: k_roll_begin marks a start; use the last one
begins = [i for i, e in enumerate(ev) if e[2] == "k_roll_begin"]
ev = ev[begins[-1]:]
passes = []; cur = None
for s, e, n, g in ev:
    if n == "k_roll_mark":
        if cur: passes.append(cur)
        cur = {"t0": s, "k": collections.defaultdict(float), "w": 0, "end": e}
    if cur is None: continue
    if n != "k_roll_plant":
        cur["k"][n] += (e - s) / 1e3; cur["end"] = max(cur["end"], e)
    if n == "k_eval": cur["w"] = g // 40
if cur: passes.append(cur)
print("passes", len(passes), "wall", (passes[-1]["end"] - passes[0]["t0"]) / 1e6, "ms")
buckets = [(8192, 8192), (4097, 8191), (1025, 4096), (513, 1024), (65, 512), (1, 64)]
for lo, hi in buckets:
    ps = [(i, p) for i, p in enumerate(passes) if lo <= p["w"] <= hi]
    if not ps: continue
    # wall per pass = start of the next pass - start of this one
    wall = sum((passes[i + 1]["t0"] if i + 1 < len(passes) else p["end"]) - p["t0"] for i, p in ps) / 1e3
    ksum = collections.defaultdict(float)
    for _, p in ps:
        for k, v in p["k"].items(): ksum[k] += v
    tot = sum(ksum.values())
    top = sorted(ksum.items(), key=lambda kv: -kv[1])[:9]
    print(f"width {lo:5d}..{hi:5d}: {len(ps):5d} passes, wall {wall/1e3:8.1f} ms ({wall/len(ps):7.1f} us per pass), kernel sum {tot/1e3:8.1f} ms | " + ", ".join(f"{k} {v/len(ps):.0f}" for k, v in top))
# drift of the full-width kernels over the rollout (sustained load: clocks?)
fw = [p for p in passes if p["w"] == 8192]
for a, b in ((0, 10), (10, 30), (30, 60), (60, 120), (120, 240), (240, 400)):
    ps = fw[a:b]
    if not ps: continue
    ks = collections.defaultdict(float)
    for p in ps:
        for k, v in p["k"].items(): ks[k] += v / len(ps)
    print(f"full-width passes {a:3d}..{b:3d}: " + ", ".join(f"{k} {ks[k]:.0f}" for k in ("k_riccati8", "k_eval", "k_expand", "k_linesearch", "k_update", "k_pick")))
