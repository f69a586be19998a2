#!/usr/bin/env python3
"""bench.py — MPC solves/s of the batched make_step path on N MI355X (one process per GPU).

Contract (driver):  python bench.py --gpus N --steps K --warmup W     (N > 1 via torch.distributed.run)
One "step" = one control tick of the reference's closed loop (src/mpc.py:140-153) for every instance of the
batch: make_step (one NLP solve per instance, warm-started from the previous solution) followed by the plant
step that produces the next x0.  Inputs are resident in HBM when the timed region starts.
Workload: BASELINE.json's metric config — horizon N=40, 8192 instances per GPU sampled along buckmore
(SURVEY.md §8d C3/C4), weak scaling (per-GPU batch fixed; instances are independent, no data-path collective).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# plant between two ticks (workload generation, not the hot path): RK4 with 100 sub-steps of 1 ms, ~4e-8 accurate
# (the parity tests use 400 sub-steps: 1e-10, the reference's CVODES tolerance)
PLANT_SUBSTEPS = 100
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Algorithmic bytes per (instance, interval, IP iteration), SURVEY.md §8(d): stage QP blocks written by the
# evaluation kernel and read by the Riccati kernel (153 words), Riccati outputs (36), evaluation inputs (46).
WORDS_QP, WORDS_RIC_OUT, WORDS_EVAL_IN = 153, 36, 46
BYTES_PER_STAGE_ITER = 8 * (2 * WORDS_QP + WORDS_RIC_OUT + WORDS_EVAL_IN)  # 3104
BYTES_BY_KERNEL = {  # share of the 3104 B each kernel class moves (algorithmic, not measured traffic)
    "eval": 8 * (WORDS_QP + WORDS_EVAL_IN), "riccati": 8 * (WORDS_QP + WORDS_RIC_OUT),
    "expand": 8 * (WORDS_EVAL_IN + WORDS_RIC_OUT), "linesearch": 8 * WORDS_EVAL_IN, "pick": 0, "update": 8 * 2 * WORDS_EVAL_IN,
    "riccati1": 0,  # one-wavefront-per-instance sweep of the narrow launches: latency-bound by construction, no roofline claim
    "step1": 0,     # fused line-search / pick / update kernel of the narrow launches: likewise
}


def load_pmc_traffic(kernel, B, N):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this same command (separate rocprofv3 --pmc
    passes, profiles/pmc_traffic.py); None when there is none for this batch / horizon."""
    import glob
    import json as _json
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*pmc_traffic.json")), reverse=True):
        try:
            d = _json.load(open(f))
        except Exception:
            continue
        k = d.get("kernels", {}).get("k_" + kernel)
        if k and d.get("batch") == B and d.get("horizon") == N:
            return dict(k, source=os.path.relpath(f, ROOT))
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8192, help="MPC instances per GPU")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--max-iter", type=int, default=150, help="interior-point iteration budget per solve")
    ap.add_argument("--soft-rho", type=float, default=0.0, help="options.soft_rho for the timed run and the CPU baseline (extension: "
                    "softened track constraints; 0 = the reference's hard constraints)")
    ap.add_argument("--poll-every", type=int, default=4, help="iterations between two read-backs of the number of unfinished instances")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="instances solved by the CPU oracle for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra measurements after the timed region (batch 1, N = 60, tuned warm start)")
    args = ap.parse_args()

    import torch
    import ltompc

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    ltompc.build_library()
    tables = ltompc.build_tables()  # buckmore / MX-5 / curvature race line (the only one the reference MPC runs on)
    B, N = args.batch, args.horizon
    opts = ltompc.default_options()
    opts.max_iter, opts.soft_rho = args.max_iter, args.soft_rho
    mpc = ltompc.BatchedMPC(tables, n_horizon=N, batch=B, options=opts, device=local_rank)
    stream = torch.cuda.current_stream(dev)
    mpc.set_stream(stream.cuda_stream)
    mpc.set_poll_every(args.poll_every)

    x0_host = ltompc.sample_x0(tables, B, seed=ltompc.scenarios.SEED + rank)
    x = torch.from_numpy(x0_host).to(dev)
    xn = torch.empty_like(x)
    u = torch.zeros(B, 2, dtype=torch.float64, device=dev)

    def tick():
        nonlocal x, xn
        mpc.make_step_dev(x.data_ptr(), u.data_ptr())
        mpc.plant_step_dev(x.data_ptr(), u.data_ptr(), xn.data_ptr(), PLANT_SUBSTEPS)
        x, xn = xn, x

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])

    # ---- warm-up: cold start (do_mpc set_initial_guess) + W ticks, untimed.  Every launch is bracketed by HIP events
    #      in the last of them: which kernel class takes the most device time (the one the roofline is quoted for) and the per-class
    #      totals come from these ticks; the timed ticks bracket the launches of that class only (two events per
    #      iteration instead of seven: the full bracketing costs 4 % of the throughput at this speed, one class 1 %).
    mpc.set_initial_guess_dev(x.data_ptr())
    for w in range(args.warmup):
        if w == args.warmup - 1:
            mpc.set_profiling(not args.no_profile)  # the last warm-up tick (a warm one if W >= 2), like the timed ticks
        tick()
    torch.cuda.synchronize(dev)
    tm_warm = mpc.timing() if (not args.no_profile and args.warmup > 0) else None
    dom = None
    if tm_warm is not None:
        dom = max((k for k in tm_warm["ms"] if BYTES_BY_KERNEL[k] > 0), key=lambda k: tm_warm["ms"][k])  # dominant wide (HBM-streaming) kernel

    # ---- timed region: exactly K ticks
    mpc.set_profiling(not args.no_profile, only=dom)
    iters_sum, solved, ip_launch_iters = 0, 0, 0
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    per_tick = []
    for _ in range(args.steps):
        tick()
        per_tick.append(mpc.timing()["ip_iterations"])
    torch.cuda.synchronize(dev)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    st = mpc.stats()  # last tick
    tm = mpc.timing()
    log = mpc.launch_log() if not args.no_profile else None
    mpc.set_profiling(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([float((st["status"] == 0).sum()), float(st["iters"].sum())], dtype=torch.float64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        n_solved_last, iters_last = agg.tolist()
    else:
        n_solved_last, iters_last = float((st["status"] == 0).sum()), float(st["iters"].sum())

    total_solves = B * world * args.steps
    value = total_solves / elapsed

    # ---- roofline of the dominant kernel (rank 0; HIP events on the launch stream, accumulated over the timed ticks)
    roofline = None
    if not args.no_profile and rank == 0:
        ms, ln = tm["ms"], tm["launches_by_kernel"]
        if dom is None:  # no warm-up ticks: every launch of the timed region was bracketed
            dom = max((k for k in ms if BYTES_BY_KERNEL[k] > 0), key=lambda k: ms[k])
        avg_ms = ms[dom] / max(1, ln[dom])
        # instances still iterating, averaged over launches (finished instances idle inside a launch)
        active_per_launch = float(st["iters"].sum()) * args.steps / max(1, sum(per_tick))  # last tick's distribution
        bytes_per_launch = active_per_launch * N * BYTES_BY_KERNEL[dom]
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "kernel_ms_total": {k: round(v, 3) for k, v in (tm_warm or tm)["ms"].items()},
                    "launches": (tm_warm or tm)["launches_by_kernel"],
                    "kernel_ms_total_from": ("the last warm-up tick, every launch bracketed"
                                             if tm_warm else "the timed ticks, every launch bracketed"),
                    "timed_region_events": ("launches of k_" + dom + " only") if tm_warm else "every launch"}
        # the same kernel over its full-width launches only (every instance of the batch still iterating or idle in
        # its wavefront; the launches after the first re-packing are sized for a few stragglers and latency-bound)
        pmc = load_pmc_traffic(dom, B, N)
        if pmc:
            roofline["traffic"] = pmc["traffic_avg_all_launches"]
            roofline["traffic_source"] = pmc["source"]
        names = list(ms.keys())
        kind, width, lms = log
        full = (kind == names.index(dom)) & (width == B)
        if full.any():
            fw_ms = float(lms[full].mean())
            fw_bytes = B * N * BYTES_BY_KERNEL[dom]
            roofline["full_width"] = {"launches": int(full.sum()), "avg_launch_ms": fw_ms, "algorithmic_bytes_per_launch": fw_bytes,
                                      "achieved": fw_bytes / (fw_ms * 1e-3) / 1e9, "frac": fw_bytes / (fw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            by_width = {}
            for w in sorted(set(int(x) for x in width[kind == names.index(dom)]), reverse=True)[:12]:
                sel = (kind == names.index(dom)) & (width == w)
                by_width[str(w)] = [int(sel.sum()), round(float(lms[sel].mean()), 4)]
            roofline["avg_launch_ms_by_width"] = by_width  # width -> [launches, avg ms]
            if pmc:
                roofline["full_width"]["traffic"] = pmc["traffic"]

    # ---- batch = 1 latency (second handle, same stream), reported as an extra
    extras = {}
    if rank == 0 and not args.no_extras:
        m1 = ltompc.BatchedMPC(tables, n_horizon=N, batch=1, options=opts, device=local_rank)
        m1.set_stream(stream.cuda_stream)
        x1 = ltompc.X0_REFERENCE[None].copy()
        m1.set_initial_guess(x1)
        u1 = m1.make_step(x1)
        x1 = m1.plant_step(x1, u1)
        torch.cuda.synchronize(dev)
        nb, t_solve = 10, 0.0
        for _ in range(nb):  # closed loop; only the solves are timed (host x0 in, host u0 out: PCIe included)
            tb = time.perf_counter()
            u1 = m1.make_step(x1)
            t_solve += time.perf_counter() - tb
            x1 = m1.plant_step(x1, u1)
        extras["batch1_solves_per_s"] = nb / t_solve
        extras["batch1_ms_per_solve"] = 1e3 * t_solve / nb
        extras["batch1_iters_last"] = int(m1.iters[0])
        m1.close()
        # BASELINE config 5 (closed loop from the reference's x0, horizon N = 60): real-time factor over the first 300
        # ticks (30 s of driving, s = 0 .. ~290 m; the formulation does not get through the narrowest section of the
        # track at s ~ 416 m, so a whole lap is not a meaningful workload)
        o60 = ltompc.default_options()
        o60.max_iter = 300
        m60 = ltompc.BatchedMPC(tables, n_horizon=60, batch=1, options=o60, device=local_rank)
        m60.set_stream(stream.cuda_stream)
        x60 = ltompc.X0_REFERENCE[None].copy()
        m60.set_initial_guess(x60)
        t60, bad60, it60 = 0.0, 0, 0
        for _ in range(300):
            tb = time.perf_counter()
            u60 = m60.make_step(x60)
            t60 += time.perf_counter() - tb
            bad60 += int(m60.status[0] != 0)
            it60 += int(m60.iters[0])
            x60 = m60.plant_step(x60, u60)
        extras["closed_loop_n60"] = {"ticks": 300, "simulated_s": 30.0, "solve_wall_s": t60, "real_time_factor": 30.0 / t60,
                                     "s_reached_m": float(x60[0, 0]), "non_converged_ticks": bad60, "ip_iters_mean": it60 / 300.0}
        m60.close()
        # ... and the whole lap with the track constraints softened (options.soft_rho = 100, do_mpc's soft_constraint /
        # penalty_term_cons; an extension: the reference's hard constraints stop the loop part-way): until the horizon
        # reaches the end of the tables
        o60.soft_rho = 100.0
        m60 = ltompc.BatchedMPC(tables, n_horizon=60, batch=1, options=o60, device=local_rank)
        m60.set_stream(stream.cuda_stream)
        x60 = ltompc.X0_REFERENCE[None].copy()
        m60.set_initial_guess(x60)
        s_end = tables.s_max - 0.1 * 60 * 25.0
        t60, bad60, it60, n60 = 0.0, 0, 0, 0
        while x60[0, 0] < s_end and n60 < 1500:
            tb = time.perf_counter()
            u60 = m60.make_step(x60)
            t60 += time.perf_counter() - tb
            bad60 += int(m60.status[0] != 0)
            it60 += int(m60.iters[0])
            x60 = m60.plant_step(x60, u60)
            n60 += 1
        extras["closed_loop_lap_soft_n60"] = {"options": {"soft_rho": 100.0}, "ticks": n60, "simulated_s": 0.1 * n60, "solve_wall_s": t60,
                                              "real_time_factor": 0.1 * n60 / t60, "s_reached_m": float(x60[0, 0]),
                                              "s_target_m": float(s_end), "non_converged_ticks": bad60, "ip_iters_mean": it60 / max(n60, 1)}
        m60.close()

    # ---- same workload with the warm start tuned for MPC (extension, not the reference's solver settings): previous
    #      solution shifted by one interval, barrier restarted at 1e-3 instead of IPOPT's 0.1.  Same NLP, same
    #      tolerance; reported as an extra, the headline `value` keeps do_mpc/IPOPT's defaults.
    if rank == 0 and not args.no_extras:
        to = ltompc.default_options()
        to.max_iter, to.warm_shift, to.mu_init_warm = args.max_iter, 1, 1e-3
        mt = ltompc.BatchedMPC(tables, n_horizon=N, batch=B, options=to, device=local_rank)
        mt.set_stream(stream.cuda_stream)
        xt = torch.from_numpy(x0_host).to(dev)
        xtn, ut = torch.empty_like(xt), torch.zeros(B, 2, dtype=torch.float64, device=dev)
        mt.set_initial_guess_dev(xt.data_ptr())
        launched = []
        for s in range(args.warmup + args.steps):
            if s == args.warmup:
                torch.cuda.synchronize(dev)
                tt = time.perf_counter()
            mt.make_step_dev(xt.data_ptr(), ut.data_ptr())
            mt.plant_step_dev(xt.data_ptr(), ut.data_ptr(), xtn.data_ptr(), PLANT_SUBSTEPS)
            xt, xtn = xtn, xt
            launched.append(mt.timing()["ip_iterations"])
        torch.cuda.synchronize(dev)
        tt = time.perf_counter() - tt
        stt = mt.stats()
        extras["tuned_warm_start"] = {"options": {"warm_shift": 1, "mu_init_warm": 1e-3}, "solves_per_s": B * args.steps / tt,
                                      "ms_per_step": 1e3 * tt / args.steps, "solved_frac_last_tick": float((stt["status"] == 0).mean()),
                                      "ip_iters_mean_last_tick": float(stt["iters"].mean()),
                                      "ip_iterations_launched_per_tick": launched[args.warmup:]}
        mt.close()

    # ---- CPU baseline: the oracle (a port of the same NLP + algorithm) on the host cores, bounded sample
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        ncores = os.cpu_count() or 1
        nthreads = min(ncores, 16)
        S = min(args.cpu_sample, B)
        oo = orc.default_options()
        oo.max_iter, oo.soft_rho = args.max_iter, args.soft_rho
        O = orc.Oracle(tables.packed(), options=oo)
        xs = x0_host[:S]
        r = O.solve(xs, N, nthreads=nthreads)  # cold start, untimed (creates the warm start)
        xs1 = O.plant_step(xs, r["u0"])
        tc = time.perf_counter()
        r2 = O.solve(xs1, N, uprev=r["u0"], warm=r, nthreads=nthreads)
        tcpu = time.perf_counter() - tc
        cpu = {"value": S / tcpu, "unit": "MPC solves/s", "cores": nthreads, "kind": "port",
               "sample": f"{S} instances of the same batch, N={N}, one warm tick, OpenMP over instances "
                         f"({nthreads} threads of {ncores} host cores), {tcpu:.1f}s wall; oracle/ltompc_oracle.c "
                         "(do_mpc/IPOPT itself is not installable offline)",
               "iters_mean": float(r2["iters"].mean()), "solved_frac": float((r2["status"] == 0).mean())}
        # the reference's own mode of operation is one process, one instance at a time: same oracle, one thread
        S1 = min(64, S)
        w1 = {k: r[k][:S1] for k in ("X", "C", "U", "L1", "L2")}
        tc = time.perf_counter()
        O.solve(xs1[:S1], N, uprev=r["u0"][:S1], warm=w1, nthreads=1)
        cpu["single_thread_value"] = S1 / (time.perf_counter() - tc)
        cpu["single_thread_sample"] = f"first {S1} instances of the same tick, 1 thread"

    if rank == 0:
        out = {
            "metric": "MPC solves/sec (N=40, nx=8 [7 + progress s], nu=2)", "value": value, "unit": "MPC solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batch={B} per GPU x {world} GPU, horizon N={N}, closed-loop warm ticks "
                                   f"(buckmore / MX-5 / curvature tables, x0 sampled along the lap, seed {ltompc.scenarios.SEED})",
                       "batch_per_gpu": B, "horizon": N, "max_iter": args.max_iter, "tol": opts.tol, "soft_rho": args.soft_rho,
                       "parallelism": f"instances sharded over {world} GPU(s), no data-path collective"},
            "solved_frac_last_tick": n_solved_last / (B * world),
            "ip_iters_mean_last_tick": iters_last / (B * world),
            "ip_iterations_launched_per_tick": per_tick,
            "roofline": roofline, "cpu_baseline": cpu, "extras": extras,
        }
        print(json.dumps(out))
    mpc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
