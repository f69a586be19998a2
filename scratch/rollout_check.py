"""Closed-loop rollout (free-running instances) against the synchronous make_step + plant_step loop: bit-identical logs; timing."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, K, NSUB = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 100
dev = torch.device("cuda", 0)
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.latency_mode = 2
for kv in os.environ.get('OPTS', '').split(','):
    if kv:
        k, v = kv.split('='); setattr(o, k, type(getattr(o, k))(float(v)))
# synchronous loop
a = ltompc.BatchedMPC(T, N, B, options=o)
a.set_stream(torch.cuda.current_stream(dev).cuda_stream)
xa = torch.from_numpy(x0).to(dev); xn = torch.empty_like(xa); ua = torch.zeros(B, 2, dtype=torch.float64, device=dev)
a.set_initial_guess_dev(xa.data_ptr())
U, S, I = [], [], []
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(K):
    a.make_step_dev(xa.data_ptr(), ua.data_ptr())
    a.plant_step_dev(xa.data_ptr(), ua.data_ptr(), xn.data_ptr(), NSUB)
    xa, xn = xn, xa
    if B <= 4096:
        s = a.stats(); U.append(ua.cpu().numpy().copy()); S.append(s["status"].copy()); I.append(s["iters"].copy())
torch.cuda.synchronize(); ts = time.perf_counter() - t0
# rollout
b = ltompc.BatchedMPC(T, N, B, options=o)
b.set_stream(torch.cuda.current_stream(dev).cuda_stream)
xb = torch.from_numpy(x0).to(dev)
ul = torch.zeros(B, K, 2, dtype=torch.float64, device=dev); sl = torch.full((B, K), -1, dtype=torch.int32, device=dev); il = torch.zeros(B, K, dtype=torch.int32, device=dev)
b.set_initial_guess_dev(xb.data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
info = b.rollout_dev(xb.data_ptr(), K, NSUB, ul.data_ptr(), sl.data_ptr(), il.data_ptr())
torch.cuda.synchronize(); tr = time.perf_counter() - t0
print(f"B={B} N={N} K={K}: sync loop {ts*1e3:.1f} ms ({B*K/ts:.0f} solves/s incl. stats reads) | rollout {tr*1e3:.1f} ms ({B*K/tr:.0f} solves/s), {info}")
tot = il.cpu().numpy().sum(1) + K   # (+1 pass per solve: the pass that finds it converged)
print(f"passes per instance over the {K} ticks: median {np.median(tot):.0f}, p99 {np.percentile(tot, 99):.0f}, max {tot.max()} (the critical path of the rollout); "
      f"sum {tot.sum()} = {tot.sum() / B:.0f} full-width iterations")
sl_h = sl.cpu().numpy(); print("rollout status histogram", np.bincount(sl_h.ravel() + 1, minlength=8)[1:], "iters mean", il.cpu().numpy().mean())
if U:
    U, S, I = np.stack(U, 1), np.stack(S, 1), np.stack(I, 1)
    print("u logs identical:", np.array_equal(U, ul.cpu().numpy()), "status identical:", np.array_equal(S, sl_h), "iters identical:", np.array_equal(I, il.cpu().numpy()),
          "final x identical:", np.array_equal(xa.cpu().numpy(), xb.cpu().numpy()), "max |du|", np.abs(U - ul.cpu().numpy()).max())
