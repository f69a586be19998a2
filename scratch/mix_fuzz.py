"""Sequences that mix the entry points must equal the plain lock-step loop: rollout then make_step, make_step then rollout then make_step,
accessors in between, host-buffer and device-pointer calls."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables(); dev = torch.device("cuda", 0)
N, B, NSUB = 20, 700, 50
o = ltompc.default_options(); o.latency_mode, o.max_iter = 2, 200
x0 = ltompc.sample_x0(T, B, seed=31)
# plain loop: 9 ticks
a = ltompc.BatchedMPC(T, N, B, options=o); a.set_initial_guess(x0)
x, U = x0.copy(), []
for t in range(9):
    u = a.make_step(x); U.append(u.copy()); x = a.plant_step(x, u, NSUB)
U = np.stack(U, 1); xa = x
# mixed: 2 ticks make_step (host buffers), stats(), rollout 3 ticks, prediction(), 1 tick make_step_dev, rollout 2 ticks, 1 tick make_step
b = ltompc.BatchedMPC(T, N, B, options=o); b.set_initial_guess(x0)
x, V = x0.copy(), []
for t in range(2):
    u = b.make_step(x); V.append(u.copy()); x = b.plant_step(x, u, NSUB)
_ = b.stats()
xd = torch.from_numpy(x).to(dev)
ul = torch.zeros(B, 3, 2, dtype=torch.float64, device=dev)
torch.cuda.synchronize(); b.rollout_dev(xd.data_ptr(), 3, NSUB, ul.data_ptr(), 0, 0); torch.cuda.synchronize()
V += [ul[:, t].cpu().numpy() for t in range(3)]
_ = b.prediction() if hasattr(b, "prediction") else None
ud = torch.zeros(B, 2, dtype=torch.float64, device=dev); xn = torch.empty_like(xd)
b.make_step_dev(xd.data_ptr(), ud.data_ptr()); b.plant_step_dev(xd.data_ptr(), ud.data_ptr(), xn.data_ptr(), NSUB); torch.cuda.synchronize()
V.append(ud.cpu().numpy().copy()); xd = xn.clone()
ul = torch.zeros(B, 2, 2, dtype=torch.float64, device=dev)
b.rollout_dev(xd.data_ptr(), 2, NSUB, ul.data_ptr(), 0, 0); torch.cuda.synchronize()
V += [ul[:, t].cpu().numpy() for t in range(2)]
x = xd.cpu().numpy()
u = b.make_step(x); V.append(u.copy()); x = b.plant_step(x, u, NSUB)
V = np.stack(V, 1)
print("controls identical tick by tick:", [bool(np.array_equal(U[:, t], V[:, t])) for t in range(9)], "final state identical:", np.array_equal(x, xa))
