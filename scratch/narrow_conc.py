"""Do concurrent NARROW chains slow each other down?  n handles of `nw` instances each (own stream, own host thread), per-tick wall time of
handle 0, for n = 1, 2, 4, 8; then the same with one wide handle (4096 instances) ticking beside them."""
import sys, os, time, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
N = 40
dev = torch.device("cuda", 0)
x0_all = ltompc.sample_x0(T, 8192)
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def make(lo, n):
    assert lo + n <= x0_all.shape[0]
    st = torch.cuda.Stream(dev)
    m = ltompc.BatchedMPC(T, N, n); m.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        x = torch.from_numpy(x0_all[lo:lo + n]).to(dev); xn = torch.empty_like(x); u = torch.zeros(n, 2, dtype=torch.float64, device=dev)
    st.synchronize()
    m.set_initial_guess_dev(x.data_ptr())
    return dict(m=m, st=st, x=x, xn=xn, u=u, t=[], it=[])
def tick(h):
    t0 = time.perf_counter()
    h["m"].make_step_dev(h["x"].data_ptr(), h["u"].data_ptr())
    h["m"].plant_step_dev(h["x"].data_ptr(), h["u"].data_ptr(), h["xn"].data_ptr(), 100)
    h["x"], h["xn"] = h["xn"], h["x"]
    h["st"].synchronize()
    h["t"].append((time.perf_counter() - t0) * 1e3); h["it"].append(h["m"].timing()["ip_iterations"])
def run(fg, k, background=()):
    stop = threading.Event()
    def bg(h):
        while not stop.is_set(): tick(h)
    tb = [threading.Thread(target=bg, args=(h,)) for h in background]
    tf = [threading.Thread(target=lambda h=h: [tick(h) for _ in range(k)]) for h in fg]
    for t in tb + tf: t.start()
    for t in tf: t.join()
    stop.set()
    for t in tb: t.join()
hs = [make(64 * i, nw) for i in range(8)]       # the SAME instances in every configuration for handle 0
wide = make(4096, 4096)
for h in hs + [wide]:
    for _ in range(4): tick(h)
K = 15
# every handle repeats the same ticks? no: states evolve; compare per-PASS times (ms per launched iteration) instead of per tick
def rep(tag, h):
    t, it = np.array(h["t"][-K:]), np.array(h["it"][-K:])
    print(f"{tag:46s} {t.sum() / it.sum() * 1e3:7.1f} us per launched pass ({t.mean():6.2f} ms per tick, {it.mean():5.1f} passes)", flush=True)
for n in (1, 2, 4, 8):
    run(hs[:n], K); rep(f"{n} narrow handle(s) of {nw}", hs[0])
for n in (1, 2, 4):
    run(hs[:n], K, background=[wide]); rep(f"{n} narrow handle(s) of {nw} beside a wide one", hs[0])
