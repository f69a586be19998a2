"""Two handles of 4096 on own streams / threads, free-running ticks; per-tick wall times with and without the wide-phase gate
(LTOMPC_GATE=1 in the environment; needs a library built with scratch/wide_gate.patch)."""
import sys, os, time, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N = 40
dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 8192 // P
x0_all = ltompc.sample_x0(T, 8192)
t_origin = time.perf_counter()
def make(lo, n):
    st = torch.cuda.Stream(dev)
    m = ltompc.BatchedMPC(T, N, n); m.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        x = torch.from_numpy(x0_all[lo:lo + n]).to(dev); xn = torch.empty_like(x); u = torch.zeros(n, 2, dtype=torch.float64, device=dev)
    st.synchronize()
    m.set_initial_guess_dev(x.data_ptr())
    return dict(m=m, st=st, x=x, xn=xn, u=u, t=[], s=[])
def tick(h):
    t0 = time.perf_counter()
    h["m"].make_step_dev(h["x"].data_ptr(), h["u"].data_ptr())
    h["m"].plant_step_dev(h["x"].data_ptr(), h["u"].data_ptr(), h["xn"].data_ptr(), 100)
    h["x"], h["xn"] = h["xn"], h["x"]
    h["st"].synchronize()
    h["s"].append((t0 - t_origin) * 1e3); h["t"].append((time.perf_counter() - t0) * 1e3)
hs = [make(i * n, n) for i in range(P)]
def phase(k):
    th = [threading.Thread(target=lambda h=h: [tick(h) for _ in range(k)]) for h in hs]
    for t in th: t.start()
    for t in th: t.join()
phase(5)
t0 = time.perf_counter(); phase(20); dt = time.perf_counter() - t0
print(f"P={P} gate={os.environ.get('LTOMPC_GATE','0')}: {dt / 20 * 1e3:.2f} ms per tick of 8192, {8192 * 20 / dt:.0f} attempted solves/s")
for i, h in enumerate(hs):
    print(i, "start", " ".join(f"{s - hs[0]['s'][-20]:6.0f}" for s in h["s"][-20:]))
    print(i, "dur  ", " ".join(f"{s:6.1f}" for s in h["t"][-20:]))
