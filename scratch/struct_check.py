"""Check the block structure of the stage matrices (A block upper triangular in the groups (s,n,mu | vx,vy,r | delta,T))."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N = 8, 10
x0 = ltompc.sample_x0(T, B)
o = ltompc.default_options(); o.max_iter = 3; o.latency_mode = 2
m = ltompc.BatchedMPC(T, N, B, options=o); m.set_initial_guess(x0); m.make_step(x0)
qp = m.debug_fetch(0).reshape(N + 1, 8, 215, 8)[:N]   # [k][b/8][field][b%8], Bp = 64
A = qp[:, 0, 0:64, :].reshape(N, 8, 8, 8)     # [k][i][j][b]
Bm = qp[:, 0, 64:80, :].reshape(N, 8, 2, 8)
np.set_printoptions(linewidth=200, precision=3, suppress=True)
print("max |A| by (i,j) over stages and instances:\n", np.abs(A).max(axis=(0, 3)))
print("max |B| by (i,c):\n", np.abs(Bm).max(axis=(0, 3)).T)
print("A[k=3,b=0]:\n", A[3, :, :, 0])
