"""Throughput of S concurrent handles (B/S instances each, own HIP stream, own host thread, ticks not synchronised)."""
import sys, os, time, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, K, Wm = 8192, 40, 16, 5
dev = torch.device("cuda", 0)
x0_all = ltompc.sample_x0(T, B)
def run(S, tuned=False, prio=False):
    o = ltompc.default_options()
    if tuned: o.warm_shift, o.mu_init_warm = 1, 1e-3
    n = B // S
    hs = []
    for s in range(S):
        st = torch.cuda.Stream(dev, priority=(-1 if (prio and s % 2) else 0))
        m = ltompc.BatchedMPC(T, N, n, options=o); m.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            x = torch.from_numpy(x0_all[s * n:(s + 1) * n]).to(dev); xn = torch.empty_like(x); u = torch.zeros(n, 2, dtype=torch.float64, device=dev)
        st.synchronize()
        m.set_initial_guess_dev(x.data_ptr())
        hs.append(dict(m=m, st=st, x=x, xn=xn, u=u))
    def ticks(h, k):
        for _ in range(k):
            h["m"].make_step_dev(h["x"].data_ptr(), h["u"].data_ptr())
            h["m"].plant_step_dev(h["x"].data_ptr(), h["u"].data_ptr(), h["xn"].data_ptr(), 100)
            h["x"], h["xn"] = h["xn"], h["x"]
        h["st"].synchronize()
    def phase(k):
        th = [threading.Thread(target=ticks, args=(h, k)) for h in hs]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(dev)
    phase(Wm)
    t0 = time.perf_counter(); phase(K); dt = time.perf_counter() - t0
    solved = sum(int((h["m"].stats()["status"] == 0).sum()) for h in hs)
    print(f"S={S:2d} tuned={int(tuned)} prio={int(prio)}: {B * K / dt:9.0f} solves/s, {dt / K * 1e3:7.1f} ms per tick of {B}, solved {solved / B:.4f}", flush=True)
    for h in hs: h["m"].close()
for S in (1, 2, 3, 4):
    run(S)
run(2, prio=True); run(4, prio=True)
