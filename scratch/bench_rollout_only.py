"""The rollout extra of bench.py in bench.py's process layout (main handle alive, second handle for the rollout)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables(); dev = torch.device("cuda", 0)
B, N, K, W = 8192, 40, 20, 5
o = ltompc.default_options(); o.latency_mode = 2
main = ltompc.BatchedMPC(T, N, B, options=o); main.set_stream(torch.cuda.current_stream(dev).cuda_stream)
x = torch.from_numpy(ltompc.sample_x0(T, B)).to(dev); u = torch.zeros(B, 2, dtype=torch.float64, device=dev); xn = torch.empty_like(x)
main.set_initial_guess_dev(x.data_ptr())
for t in range(3):
    main.make_step_dev(x.data_ptr(), u.data_ptr()); main.plant_step_dev(x.data_ptr(), u.data_ptr(), xn.data_ptr(), 100); x, xn = xn, x
torch.cuda.synchronize()
mr = ltompc.BatchedMPC(T, N, B, options=o); mr.set_stream(torch.cuda.current_stream(dev).cuda_stream)
xb = torch.from_numpy(ltompc.sample_x0(T, B)).to(dev); mr.set_initial_guess_dev(xb.data_ptr())
def run(K):
    sl = torch.full((B, K), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mr.rollout_dev(xb.data_ptr(), K, 100, 0, sl.data_ptr(), 0)
    torch.cuda.synchronize(); return time.perf_counter() - t0, sl.cpu().numpy()
run(W); tr, sl = run(K)
print(f"plant streams {os.environ.get('LTOMPC_PLANT_STREAMS', 'default')}, GPU_MAX_HW_QUEUES {os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: rollout {tr*1e3:.1f} ms, {np.isin(sl, (0, 1)).sum()/tr:.0f} converged solves/s")
