"""How much does a narrow handle (<= 512 instances: k_riccati1 / k_step1 launches only) slow down while a wide handle's
kernels run beside it on another stream, and vice versa?  Per-tick wall time of each, alone and together."""
import sys, os, time, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N = 40
dev = torch.device("cuda", 0)
x0_all = ltompc.sample_x0(T, 8192)  # narrow handle: rows from 0, wide handle: the last ww rows
def make(lo, n):
    assert lo + n <= x0_all.shape[0], 'rows out of range'
    st = torch.cuda.Stream(dev)
    m = ltompc.BatchedMPC(T, N, n); m.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        x = torch.from_numpy(x0_all[lo:lo + n]).to(dev); xn = torch.empty_like(x); u = torch.zeros(n, 2, dtype=torch.float64, device=dev)
    st.synchronize()
    m.set_initial_guess_dev(x.data_ptr())
    return dict(m=m, st=st, x=x, xn=xn, u=u, t=[], prof=None)
def tick(h):
    t0 = time.perf_counter()
    h["m"].make_step_dev(h["x"].data_ptr(), h["u"].data_ptr())
    h["m"].plant_step_dev(h["x"].data_ptr(), h["u"].data_ptr(), h["xn"].data_ptr(), 100)
    h["x"], h["xn"] = h["xn"], h["x"]
    h["st"].synchronize()
    h["t"].append((time.perf_counter() - t0) * 1e3)
def run(hs, k, background=()):
    stop = threading.Event()
    def bg(h):
        while not stop.is_set(): tick(h)
    def fg(h):
        for _ in range(k): tick(h)
    tb = [threading.Thread(target=bg, args=(h,)) for h in background]
    tf = [threading.Thread(target=fg, args=(h,)) for h in hs]
    for t in tb + tf: t.start()
    for t in tf: t.join()
    stop.set()
    for t in tb: t.join()
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ww = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
A = make(0, nw); W = make(8192 - ww, ww)
for h in (A, W):
    for _ in range(4): tick(h)
def rep(tag, h, n):
    t = np.array(h["t"][-n:]); print(f"{tag:40s} {t.mean():8.2f} ms per tick (min {t.min():.2f} max {t.max():.2f})", flush=True)
run([A], 12); rep(f"narrow {nw} alone", A, 12)
run([W], 8); rep(f"wide {ww} alone", W, 8)
run([A], 12, background=[W]); rep(f"narrow {nw} beside wide {ww}", A, 12)
run([W], 8, background=[A]); rep(f"wide {ww} beside narrow {nw}", W, 8)
A2 = make(512, nw)
for _ in range(4): tick(A2)
run([A], 12, background=[A2]); rep(f"narrow {nw} beside narrow {nw}", A, 12)
