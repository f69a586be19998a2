import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 100, 40
x0 = ltompc.sample_x0(T, 8192)[:B]
def run(tag):
    o = ltompc.default_options(); o.max_iter = 300
    m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0)
    u = m.make_step(x0); s = m.stats(); m.close()
    return u, s
u1, s1 = run("a"); u2, s2 = run("b")
print("tail env", os.environ.get("LTOMPC_TAIL"), "repeat identical:", np.array_equal(u1, u2), np.array_equal(s1["iters"], s2["iters"]), "max diff", np.abs(u1-u2).max())
np.save("gpurun_out/det_u_%s.npy" % os.environ.get("LTOMPC_TAIL", "dflt"), u1); np.save("gpurun_out/det_it_%s.npy" % os.environ.get("LTOMPC_TAIL", "dflt"), s1["iters"])
