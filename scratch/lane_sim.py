"""Discrete-event estimate: rollout of K ticks with one lane (as built) vs a wide lane + a narrow fast lane for the laggards.
Input: per-instance passes per tick from scratch/tail_predict.py (gpurun_out/tail_predict_passes.npy)."""
import numpy as np, sys
d = np.load("gpurun_out/tail_predict_passes.npy").astype(np.int64)
P = (d[0] + d[2])[5:25].T + 3          # [B, K] passes per tick: iterations + sweep retries + (convergence pass, plant, init)
B, K = P.shape
cum = np.cumsum(P, 1)                   # passes needed to finish tick t
total = cum[:, -1]
print("passes per instance: median", np.median(total), "p99", np.percentile(total, 99), "max", total.max(), "full-width equivalent", total.sum() / B)
def t_wide(w):   # ms per pass at launch width w (measured: 1.28 at 8192, ~0.3 at <= 512)
    return 0.3 if w <= 512 else 0.3 + 0.98 * (w - 512) / 7680
def one_lane():
    done = np.zeros(B, np.int64); t = 0.0
    while True:
        left = done < total
        w = left.sum()
        if w == 0: return t
        # 4 passes between polls
        t += 4 * t_wide(w); done[left] += 4
def two_lanes(L=256, t_fast_loaded=0.40, poll=4, ratio_cap=16):
    done = np.zeros(B, np.int64); t = 0.0
    while True:
        left = done < total
        n = left.sum()
        if n == 0: return t
        # laggards: the L instances with the most passes still to do would be ideal; a scheduler sees only the past:
        # rank by ticks completed (fewer first), then by passes spent in the current solve (more first)
        ticks_done = (cum <= done[:, None]).sum(1)
        in_solve = done - np.where(ticks_done > 0, np.take_along_axis(cum, np.maximum(ticks_done - 1, 0)[:, None], 1)[:, 0], 0)
        key = np.where(left, ticks_done * 10000 - in_solve, 1 << 60)
        if n <= 512:
            t += poll * t_wide(n); done[left] += poll; continue
        fast = np.argsort(key, kind="stable")[:L]
        isfast = np.zeros(B, bool); isfast[fast] = True; isfast &= left
        wide = left & ~isfast
        dt = poll * t_wide(wide.sum())
        nf = min(int(dt / t_fast_loaded), ratio_cap * poll)
        t += dt; done[wide] += poll; done[isfast] += nf
print("one lane  : %.0f ms -> %.0f solves/s" % (one_lane(), B * K / one_lane() * 1e3))
for L in (64, 128, 256, 512):
    for tf in (0.35, 0.45):
        tt = two_lanes(L, tf)
        print(f"two lanes, {L:4d} in the fast lane at {tf} ms per pass: {tt:.0f} ms -> {B * K / tt * 1e3:.0f} solves/s")
