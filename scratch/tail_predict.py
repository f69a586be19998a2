"""How predictable are a tick's stragglers from the previous tick?  (groundwork for a second, narrow lane that starts them early)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
B, N, K = 8192, 40, int(sys.argv[1]) if len(sys.argv) > 1 else 25
o = ltompc.default_options()
m = ltompc.BatchedMPC(T, N, B, options=o)
x = ltompc.sample_x0(T, B)
m.set_initial_guess(x)
IT, ST, NREG = [], [], []
for t in range(K):
    u = m.make_step(x)
    s = m.stats()
    IT.append(s["iters"].copy()), ST.append(s["status"].copy()), NREG.append(s["n_reg"].copy())
    x = m.plant_step(x, u, 100)
IT, ST, NREG = np.stack(IT), np.stack(ST), np.stack(NREG)
P = IT + NREG  # passes (iterations + sweep retries)
np.save(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "tail_predict_passes.npy"), np.stack([IT, ST, NREG]).astype(np.int16))
for thr in (28, 32, 40):
    print(f"--- stragglers: passes > {thr}")
    for t in range(5, K):
        S = P[t] > thr
        for name, pred in (("prev passes>thr or status>=4", (P[t-1] > thr) | (ST[t-1] >= 4)), ("prev passes>24 or status>=4", (P[t-1] > 24) | (ST[t-1] >= 4)),
                           ("any of last 3 ticks >thr", (P[t-3:t] > thr).any(0) | (ST[t-1] >= 4))):
            rest = P[t][~pred]
            if t in (5, 10, 15, 20, 24):
                print(f"tick {t:2d} max passes {P[t].max():4d} | stragglers {S.sum():4d} | predictor '{name}': size {pred.sum():4d}, recall {(S & pred).sum() / max(S.sum(),1):.2f}, "
                      f"max passes outside the predicted set {rest.max():4d}, p99.9 outside {np.percentile(rest, 99.9):.0f}")
