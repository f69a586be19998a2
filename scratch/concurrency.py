"""Concurrency picture of a multi-handle run from a rocprofv3 kernel trace (CSV): per stream (queue) busy time, time with
0 / 1 / 2 / ... kernels in flight, kernel time by class alone vs overlapped.   usage: concurrency.py <trace dir> [t_skip_frac]"""
import csv, sys, glob, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(f)))
ev = np.array([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows], dtype=np.int64)
name = np.array([r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("ltompc::", "") for r in rows])
queue = np.array([r.get("Queue_Id", "0") for r in rows])
grid = np.array([int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) for r in rows])
t0, t1 = ev[:, 0].min(), ev[:, 1].max()
lo = t0 + int((t1 - t0) * skip)      # steady state: the last part of the run
m = ev[:, 0] >= lo
ev, name, queue, grid = ev[m], name[m], queue[m], grid[m]
span = (ev[:, 1].max() - ev[:, 0].min()) / 1e6
print(f"{m.sum()} launches in the last {span:.1f} ms of the trace; queues: {sorted(set(queue))}")
# in-flight histogram
pts = np.concatenate([np.stack([ev[:, 0], np.ones(len(ev), dtype=np.int64)], 1), np.stack([ev[:, 1], -np.ones(len(ev), dtype=np.int64)], 1)])
pts = pts[np.argsort(pts[:, 0], kind="stable")]
level = np.cumsum(pts[:, 1])
dt = np.diff(pts[:, 0])
hist = collections.Counter()
for l, d in zip(level[:-1], dt): hist[int(l)] += d
tot = sum(hist.values())
print("kernels in flight -> share of time: " + "  ".join(f"{k}: {100 * v / tot:.1f}%" for k, v in sorted(hist.items())))
# the same for 'wide' kernels only (grid >= 65536 threads) and narrow only
for tag, sel in (("wide (>= 65536 threads)", grid >= 65536), ("narrow (< 65536 threads)", grid < 65536)):
    e = ev[sel]
    pts = np.concatenate([np.stack([e[:, 0], np.ones(len(e), dtype=np.int64)], 1), np.stack([e[:, 1], -np.ones(len(e), dtype=np.int64)], 1)])
    pts = pts[np.argsort(pts[:, 0], kind="stable")]
    level = np.cumsum(pts[:, 1]); dt = np.diff(pts[:, 0])
    h = collections.Counter()
    for l, d in zip(level[:-1], dt): h[int(l)] += d
    covered = sum(v for k, v in h.items() if k > 0)
    print(f"{tag}: in flight {100 * covered / tot:.1f}% of the time; levels " + "  ".join(f"{k}: {100 * v / tot:.1f}%" for k, v in sorted(h.items()) if k > 0))
# per queue: busy share and gaps
for q in sorted(set(queue)):
    e = ev[queue == q]
    busy = (e[:, 1] - e[:, 0]).sum()
    print(f"queue {q}: {len(e)} launches, busy {100 * busy / tot:.1f}% of the span")
# per class: mean duration
print(f"{'kernel':28s} {'n':>7s} {'total ms':>9s} {'mean us':>8s} {'median us':>9s}")
for k in sorted(set(name), key=lambda k: -(ev[name == k][:, 1] - ev[name == k][:, 0]).sum()):
    d = (ev[name == k][:, 1] - ev[name == k][:, 0]) / 1e3
    if d.sum() < 100: continue
    print(f"{k:28s} {len(d):7d} {d.sum() / 1e3:9.2f} {d.mean():8.1f} {np.median(d):9.1f}")
