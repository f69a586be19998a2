"""Bit-level regression of a refactoring: save / compare u0, iterations and status of a fixed batch (both modes)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
out = {}
for name, B, N, mode in (("slot", 600, 40, 2), ("wave", 16, 20, 1)):
    x0 = ltompc.sample_x0(T, B, seed=77)
    o = ltompc.default_options(); o.max_iter = 200; o.latency_mode = mode
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    u0 = m.make_step(x0); x1 = m.plant_step(x0, u0); u1 = m.make_step(x1)
    out[name + "_u0"], out[name + "_u1"], out[name + "_it"], out[name + "_st"] = u0, u1, m.iters.copy(), m.status.copy()
    m.close()
f = "gpurun_out/regress_ref.npz" if sys.argv[1] == "save" else "scratch/regress_ref.npz"
if sys.argv[1] == "save":
    np.savez(f, **out); print("saved", {k: v.shape for k, v in out.items()})
else:
    ref = np.load(f)
    for k in out:
        same = np.array_equal(ref[k], out[k])
        print(k, "identical" if same else f"DIFFERENT: max abs {np.abs(ref[k].astype(float) - out[k]).max():.3e}, {int((ref[k] != out[k]).sum())} entries")
