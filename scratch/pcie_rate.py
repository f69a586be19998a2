"""Host-buffer entry point (x0 in, u0 out over PCIe) against the device-pointer one, B = 8192, N = 40, warm ticks."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8192, 40
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 150
for mode in ("host", "dev"):
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    x = x0.copy()
    if mode == "dev":
        dev = torch.device("cuda", 0); m.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        xd = torch.from_numpy(x).to(dev); xn = torch.empty_like(xd); ud = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    ts = []
    for tick in range(6):
        if mode == "host":
            t0 = time.perf_counter(); u = m.make_step(x); ts.append(time.perf_counter() - t0)
            x = m.plant_step(x, u, 100)
        else:
            torch.cuda.synchronize(); t0 = time.perf_counter(); m.make_step_dev(xd.data_ptr(), ud.data_ptr()); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            m.plant_step_dev(xd.data_ptr(), ud.data_ptr(), xn.data_ptr(), 100); xd, xn = xn, xd
    print(f"{mode}: make_step per tick {np.mean(ts[2:])*1e3:.2f} ms -> {B/np.mean(ts[2:]):.0f} solves/s")
    m.close()
