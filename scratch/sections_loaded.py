"""Section timers of the one-instance kernels (LTOMPC_DBG=1: k_riccati1 / k_riccati1q head, staging, backward, forward; k_step1 line
search, wait, pick, update) for a handle of 8 instances, alone and while a handle of 4096 ticks beside it: which part of a narrow
pass gets longer on a loaded chip?"""
import sys, os, time, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N = 40
dev = torch.device("cuda", 0)
x0_all = ltompc.sample_x0(T, 8192)
def make(lo, n, dbg):
    if dbg: os.environ["LTOMPC_DBG"] = "1"
    else: os.environ.pop("LTOMPC_DBG", None)
    st = torch.cuda.Stream(dev)
    m = ltompc.BatchedMPC(T, N, n); m.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        x = torch.from_numpy(x0_all[lo:lo + n]).to(dev); xn = torch.empty_like(x); u = torch.zeros(n, 2, dtype=torch.float64, device=dev)
    st.synchronize()
    m.set_initial_guess_dev(x.data_ptr())
    return dict(m=m, st=st, x=x, xn=xn, u=u)
def tick(h):
    h["m"].make_step_dev(h["x"].data_ptr(), h["u"].data_ptr())
    h["m"].plant_step_dev(h["x"].data_ptr(), h["u"].data_ptr(), h["xn"].data_ptr(), 100)
    h["x"], h["xn"] = h["xn"], h["x"]
    h["st"].synchronize()
small = make(0, 8, True); wide = make(4096, 4096, False)
for h in (small, wide):
    for _ in range(4): tick(h)
def measure(tag, k=20):
    c0 = small["m"].debug_fetch(14)[:13].copy()
    t0 = time.perf_counter()
    for _ in range(k): tick(small)
    dt = time.perf_counter() - t0
    d = small["m"].debug_fetch(14)[:13] - c0
    print(f"{tag}: {dt / k * 1e3:.2f} ms per tick | riccati1 launches {int(d[4])}: cycles per launch head(+staging) {d[0]/d[4]:.0f} (staging {d[1]/d[4]:.0f}) backward {d[2]/d[4]:.0f} forward {d[3]/d[4]:.0f}"
          f" | step1 {int(d[12])}: line search {d[8]/d[12]:.0f} wait {d[9]/d[12]:.0f} pick {d[10]/d[12]:.0f} update {d[11]/d[12]:.0f}", flush=True)
measure("alone          ")
stop = threading.Event()
th = threading.Thread(target=lambda: [tick(wide) for _ in iter(lambda: stop.is_set(), True)])
th.start()
time.sleep(0.2)
measure("beside 4096    ")
stop.set(); th.join()
measure("alone again    ")
