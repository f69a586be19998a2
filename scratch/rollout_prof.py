"""Rollout only (for rocprofv3 --kernel-trace --stats): W warm-up ticks, then K timed ticks; prints the poll history."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, K, W = (int(a) for a in sys.argv[1:5]); NSUB = int(os.environ.get("NSUB", 100))
dev = torch.device("cuda", 0)
o = ltompc.default_options(); o.latency_mode = 2
for kv in os.environ.get('OPTS', '').split(','):
    if kv:
        k, v = kv.split('='); setattr(o, k, type(getattr(o, k))(float(v)))
b = ltompc.BatchedMPC(T, N, B, options=o)
b.set_stream(torch.cuda.current_stream(dev).cuda_stream)
xb = torch.from_numpy(ltompc.sample_x0(T, B)).to(dev)
b.set_initial_guess_dev(xb.data_ptr())
def run(K):
    ul = torch.zeros(B, K, 2, dtype=torch.float64, device=dev); sl = torch.full((B, K), -1, dtype=torch.int32, device=dev); il = torch.zeros(B, K, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = b.rollout_dev(xb.data_ptr(), K, NSUB, ul.data_ptr(), sl.data_ptr(), il.data_ptr())
    torch.cuda.synchronize(); return time.perf_counter() - t0, info, sl.cpu().numpy(), il.cpu().numpy()
run(W)
tr, info, sl, il = run(K)
conv = np.isin(sl, (0, 1)).sum()
tot = il.sum(1) + K
print(f"B={B} N={N} K={K}: rollout {tr*1e3:.1f} ms, {conv/tr:.0f} converged solves/s, {info}; passes per instance median {np.median(tot):.0f} p99 {np.percentile(tot,99):.0f} max {tot.max()}, "
      f"full-width equivalent {tot.sum()/B:.0f}")
h = b.history()
# time-weighted width profile (polls are every poll_every passes)
print("polls (pass, instances with ticks left, launch width):")
for i in range(0, len(h), max(1, len(h) // 40)): print("  ", h[i])
