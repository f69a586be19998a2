"""Three handles driven from three host threads at once (own streams) give the results of running them one after the other."""
import sys, os, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables(); dev = torch.device("cuda", 0)
cfg = [(40, 1500, 11), (20, 300, 12), (10, 64, 13)]
def loop(N, B, seed, out, stream=None):
    o = ltompc.default_options(); o.max_iter = 200
    m = ltompc.BatchedMPC(T, N, B, options=o)
    if stream is not None: m.set_stream(stream.cuda_stream)
    x = torch.from_numpy(ltompc.sample_x0(T, B, seed=seed)).to(dev); xn = torch.empty_like(x); u = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)
    m.set_initial_guess_dev(x.data_ptr())
    res = []
    for t in range(6):
        m.make_step_dev(x.data_ptr(), u.data_ptr()); m.plant_step_dev(x.data_ptr(), u.data_ptr(), xn.data_ptr(), 50)
        (stream or torch.cuda.current_stream(dev)).synchronize()
        res.append(u.cpu().numpy().copy()); x, xn = xn, x
    st = m.stats(); m.close()
    out.append((np.stack(res), st["status"].copy(), st["iters"].copy()))
seq = []
for c in cfg:
    o = []; loop(*c, o); seq.append(o[0])
par = [[] for _ in cfg]
streams = [torch.cuda.Stream(dev) for _ in cfg]
th = [threading.Thread(target=loop, args=(*c, par[i], streams[i])) for i, c in enumerate(cfg)]
for t in th: t.start()
for t in th: t.join()
for i, c in enumerate(cfg):
    print(c, "identical:", all(np.array_equal(a, b) for a, b in zip(seq[i], par[i][0])))
