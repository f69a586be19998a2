#!/usr/bin/env python3
"""bench.py — MPC solves/s of the batched make_step path on N MI355X (one process per GPU).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1: this process is the one rank.  N > 1 and no RANK in the environment: this process starts
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py ...` as a CHILD (before
  anything touches the GPU) and relays rank 0's JSON line; under torchrun (RANK set) it is one of the N ranks.
One "step" = one control tick of the reference's closed loop (src/mpc.py:140-153) for every instance of the batch:
make_step (one NLP solve per instance, warm-started from the previous solution, the reference's solver settings incl.
max_iter = 1000) followed by the plant step that produces the next x0.  Inputs are resident in HBM when the timed region starts.
Workload: BASELINE.json's metric config — horizon N = 40, 8192 instances per GPU sampled along buckmore (SURVEY.md §8d
C3/C4), weak scaling (per-GPU batch fixed; instances are independent NLPs, no data-path collective): that is `value`.
With N > 1 the same command ALSO measures BASELINE config 4 as stated, 8192 instances in TOTAL sharded over the N GPUs
(`config4_total_8192`, strong scaling, 8192 / N per GPU), and every tick gathers the controls on all ranks over RCCL
(`sharding.gather_rows`, the one collective north_star names; its cost is reported as `gather_ms`).  `--total-batch T` makes
the strong-scaling shape the headline instead (`"scaling": "strong"`).
`value` counts CONVERGED solves only (status SOLVED / ACCEPTABLE); `solves_attempted_per_s` has every instance.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import subprocess
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
    "riccati1": 8 * (WORDS_QP + WORDS_RIC_OUT),  # the one-wavefront-per-instance sweep of the narrow launches: same words as "riccati"
    "step1": 8 * (WORDS_EVAL_IN + 2 * WORDS_EVAL_IN),  # fused line search + pick + update of the narrow launches: their words together
}
KERNEL_OF_CLASS = {"eval": "k_eval", "riccati": "k_riccati8", "expand": "k_expand", "linesearch": "k_linesearch", "pick": "k_pick",
                   "update": "k_update", "riccati1": "k_riccati1", "step1": "k_step1"}
STATUS_NAMES = ("solved", "acceptable", "max_iter", "numerical", "stalled", "infeasible", "6", "other")


def load_pmc_traffic(kernel, B, N):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this same command (separate rocprofv3 --pmc
    passes, profiles/pmc_traffic.py); None when there is none for this batch / horizon."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        k = d.get("kernels", {}).get(kernel)
        if k and d.get("batch") == B and d.get("horizon") == N:
            return dict(k, source=os.path.relpath(f, ROOT))
    return None


def host_cpu_share():
    """(threads, description): the CPUs this process can actually use = min(affinity mask, cgroup CPU quota).  On the GPU
    box os.cpu_count() is 256 but the container's cgroup quota is 16 CPUs: 256 OpenMP threads on a 16-CPU quota run at
    half the rate of 16 threads (measured: 758 vs 1414 solves/s)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    desc = f"affinity mask {n} CPUs"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = max(1, int(round(int(quota) / int(period))))
            desc += f", cgroup cpu.max {quota}/{period} = {q} CPUs"
            n = min(n, q)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                q = max(1, int(round(quota / period)))
                desc += f", cgroup cfs quota {quota}/{period} = {q} CPUs"
                n = min(n, q)
        except (OSError, ValueError):
            pass
    return n, desc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8192, help="MPC instances per GPU")
    ap.add_argument("--parts", type=int, default=0, help="handles the batch of a GPU is split into, each on its own HIP stream and host thread, ticking at "
                    "its own pace (SplitMPC: one part's narrow tail overlaps the others' full-width launches; results bit-identical to one "
                    "handle).  0 = 4 for a GPU with 4096 instances or more, else 1")
    ap.add_argument("--narrow-width", type=int, default=-1, help="split handles: widest launch that uses the one-instance kernels (ltompc_set_narrow_width); -1 = SplitMPC's default (128 for three or more parts)")
    ap.add_argument("--total-batch", type=int, default=0, help="strong scaling: this many instances in TOTAL, sharded over the GPUs in contiguous "
                    "blocks (BASELINE config 4: 8192); 0 = weak scaling with --batch instances per GPU")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--max-iter", type=int, default=1000, help="interior-point iteration budget per solve (reference: ipopt.max_iter = 1000, controller.py:18)")
    ap.add_argument("--soft-rho", type=float, default=0.0, help="options.soft_rho for the timed run and the CPU baseline (extension: "
                    "softened track constraints; 0 = the reference's hard constraints)")
    ap.add_argument("--resto-sticky", type=int, default=0, help="options.resto_sticky for the timed run and the CPU baseline (extension: instances "
                    "that needed the restoration phase start their next solves in elastic mode; 0 = every solve starts on the hard constraints)")
    ap.add_argument("--poll-every", type=int, default=4, help="iterations between two read-backs of the number of unfinished instances")
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances solved by the CPU oracle for cpu_baseline (0: sized for ~10 s on the host's cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra measurements after the timed region (batch 1, N = 60, max_iter 150, tuned warm start)")
    ap.add_argument("--selftest-stub", action="store_true", help="CPU-only plumbing test of the multi-rank path (tests/test_distributed_cpu.py): gloo "
                    "instead of RCCL and a stand-in for the solver; the line it prints says data = stub and is not a measurement")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """--gpus N > 1 outside torchrun: start the N ranks as a child process group and pass rank 0's JSON line through.
    Nothing in this process has touched the GPU (no torch.cuda / HIP call): it only waits for the child."""
    port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:  # relay: the last JSON object line rank 0 printed is the result
        out = out.rstrip("\n")
        if out.startswith("{") and out.endswith("}"):
            line = out
        else:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        n = json.loads(line).get("n_gpus")
        if n != args.gpus:
            print(f"bench.py: asked for {args.gpus} ranks, the result reports {n}", file=sys.stderr)
            rc = rc or 1
        print(line)
    return rc if rc or line is not None else 1


def stub_main(args):
    """Plumbing test of the multi-rank path without a GPU (never a measurement): the same launcher, rendezvous, barrier,
    max-over-ranks timing and sum-over-ranks aggregation as the real run, with gloo and a stand-in for the solver."""
    import torch
    import torch.distributed as dist
    from importlib import import_module
    shard = import_module("lap-time-optimization_amd.sharding")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    def shape(total):
        """one timed region: total = 0 -> args.batch per rank (weak), else `total` instances sharded over the ranks (strong)"""
        lo, hi = shard.shard_range(total, rank, world) if total else (rank * args.batch, (rank + 1) * args.batch)
        Bl, n_all = hi - lo, (total if total else args.batch * world)
        for _ in range(args.warmup):
            time.sleep(0.001)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        converged, checksum = 0, 0.0
        for k in range(args.steps):
            time.sleep(0.002 * (1 + rank))  # ranks differ: the slowest one sets the time
            converged += Bl - rank  # rank r "fails" r instances per tick
            u = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1).repeat(1, 2) + k   # stand-in for the controls of the shard
            g = shard.gather_rows(u, n_all, rank, world)  # the per-tick collective of the real run
            assert g.shape == (n_all, 2) and float(g[-1, 0]) == n_all - 1 + k
            checksum += float(g.sum())
        elapsed = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        _, _, elapsed_max = shard.reduce_stats(0, 0, elapsed)
        tot = torch.tensor([float(converged)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        return dict(value=float(tot.item()) / elapsed_max, ms_per_step=1e3 * elapsed_max / args.steps, converged_solves=float(tot.item()),
                    batch_per_gpu=Bl, total_batch=n_all, gathered_checksum=checksum)
    head = shape(args.total_batch)
    extra = shape(8192) if (world > 1 and not args.total_batch) else None
    if rank == 0:
        out = {"metric": "MPC solves/sec (N=40, nx=8 [7 + progress s], nu=2)", "value": head["value"], "unit": "MPC solves/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "strong" if args.total_batch else "weak", "vs_baseline": None, "dtype": "f64", "data": "stub",
               "config": {"workload": "plumbing self-test, no solver", "batch_per_gpu": head["batch_per_gpu"], "total_batch": head["total_batch"]},
               "converged_solves": head["converged_solves"], "gathered_checksum": head["gathered_checksum"]}
        if extra:
            out["config4_total_8192"] = dict(extra, scaling="strong")
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.selftest_stub:
        return stub_main(args)

    import torch
    import ltompc

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}")
    # Rehearsal of the N > 1 path on a box with ONE GPU (development only, never a measurement: the line says so):
    # LTOMPC_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo for the collectives (RCCL refuses two ranks on one device).
    rehearsal = os.environ.get("LTOMPC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    ltompc.build_library()
    tables = ltompc.build_tables()  # buckmore / MX-5 / curvature race line (the only one the reference MPC runs on)
    shard = importlib.import_module("lap-time-optimization_amd.sharding")
    N = args.horizon
    lo, hi = shard.shard_range(args.total_batch, rank, world) if args.total_batch else (rank * args.batch, (rank + 1) * args.batch)
    B, n_total = hi - lo, (args.total_batch if args.total_batch else args.batch * world)
    opts = ltompc.default_options()
    opts.max_iter, opts.soft_rho, opts.resto_sticky = args.max_iter, args.soft_rho, args.resto_sticky
    n_parts = args.parts if args.parts > 0 else (4 if B >= 4096 else 1)
    split = n_parts > 1
    stream = torch.cuda.current_stream(dev)
    if split:
        mpc = ltompc.SplitMPC(tables, n_horizon=N, batch=B, n_parts=n_parts, options=opts, device=local_rank,  # (each part on its own stream)
                              narrow_width=None if args.narrow_width < 0 else args.narrow_width)
    else:
        mpc = ltompc.BatchedMPC(tables, n_horizon=N, batch=B, options=opts, device=local_rank)
        mpc.set_stream(stream.cuda_stream)
    mpc.set_poll_every(args.poll_every)
    prof = mpc.parts[0] if split else mpc   # the handle whose launches are logged (roofline accounting): B_acc instances
    B_acc = prof.B

    # weak scaling: every rank samples its own 8192 states; strong scaling: ONE batch of n_total states, rank r holds rows [lo, hi)
    x0_host = ltompc.sample_x0(tables, n_total, seed=ltompc.scenarios.SEED)[lo:hi] if args.total_batch else ltompc.sample_x0(tables, B, seed=ltompc.scenarios.SEED + rank)
    x = torch.from_numpy(np.ascontiguousarray(x0_host)).to(dev)
    xn = torch.empty_like(x)
    u = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    u_ring = [u, torch.zeros_like(u)]  # split handles on several ranks: tick t's controls go to u_ring[t % 2] (see run_split)

    gather_ev = []  # (start, end) events around the gather of every timed tick

    def run_split(k, after_tick=None, timed=False):
        """k ticks of every part at its own pace (SplitMPC.run_ticks); the states end in x.
        With several ranks the controls of the whole batch are still gathered on every rank once per tick (north_star: results
        gathered over RCCL / xGMI), by THIS thread, so that every rank issues the same sequence of collectives: the parts write
        tick t's controls to u_ring[t % 2]; when all parts of the rank have finished tick t its block is gathered, and a part starts
        tick t + 2 (which writes the same buffer again) only after that gather has completed - the parts run at most one tick
        ahead of the gather, and nothing of a tick's data path waits for another rank."""
        nonlocal x, xn
        torch.cuda.synchronize(dev)
        if world == 1:
            mpc.run_ticks(x.data_ptr(), u.data_ptr(), xn.data_ptr(), k, PLANT_SUBSTEPS, after_tick=after_tick)
        else:
            import threading
            pipe = shard.TickPipeline(n_parts, depth=len(u_ring))

            def after(pi, t):
                if after_tick is not None:
                    after_tick(pi, t)
                mpc.parts[pi].synchronize()  # tick t's controls are in u_ring[t % 2]
                pipe.after_tick(pi, t)

            def ticks():
                try:
                    mpc.run_ticks(x.data_ptr(), [q.data_ptr() for q in u_ring], xn.data_ptr(), k, PLANT_SUBSTEPS, after_tick=after,
                                  before_tick=pipe.before_tick)
                except BaseException as e:  # (a part failed: let the gather loop below end instead of waiting for ever)
                    pipe.fail(e)

            def gather(t):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                shard.gather_rows(u_ring[t % len(u_ring)], n_total, rank, world)
                e1.record(stream)
                stream.synchronize()  # the slot may be written again once this returns
                if timed:
                    gather_ev.append((e0, e1))

            th = threading.Thread(target=ticks)
            th.start()
            try:
                pipe.consume(k, gather)
            finally:
                th.join()
            if pipe.failure is not None:
                raise pipe.failure
        if k % 2:
            x, xn = xn, x

    def tick(timed=False):
        nonlocal x, xn
        mpc.make_step_dev(x.data_ptr(), u.data_ptr())
        mpc.plant_step_dev(x.data_ptr(), u.data_ptr(), xn.data_ptr(), PLANT_SUBSTEPS)
        x, xn = xn, x
        if world > 1:
            # the controls of the whole batch on every rank (north_star: results gathered over RCCL / xGMI): one all_gather of
            # 16 B per instance, on the stream the solver runs on; nothing in the next tick depends on it
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            shard.gather_rows(u, n_total, rank, world)
            e1.record(stream)
            if timed:
                gather_ev.append((e0, e1))

    def barrier():
        if world > 1:
            dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])

    # ---- warm-up: cold start (do_mpc set_initial_guess) + W ticks, untimed.  Every launch is bracketed by HIP events
    #      in the last of them: which kernel class takes the most device time (the one the roofline is quoted for; all 8
    #      classes compete) and the per-class totals come from that tick; the timed ticks bracket the launches of that class
    #      only (two events per iteration instead of seven: the full bracketing costs 4 % of the throughput, one class 1 %).
    torch.cuda.synchronize(dev)
    mpc.set_initial_guess_dev(x.data_ptr())
    if split:
        if args.warmup > 1:
            run_split(args.warmup - 1)
        if args.warmup > 0:
            mpc.set_profiling(not args.no_profile)
            run_split(1)
    else:
        for w in range(args.warmup):
            if w == args.warmup - 1:
                mpc.set_profiling(not args.no_profile)  # the last warm-up tick (a warm one if W >= 2), like the timed ticks
            tick()
    torch.cuda.synchronize(dev)
    tm_warm = prof.timing() if (not args.no_profile and args.warmup > 0) else None
    log_warm = prof.launch_log(with_iterations=True) if tm_warm is not None else None  # every launch of that tick, by class and width
    dom = max(tm_warm["ms"], key=lambda k: tm_warm["ms"][k]) if tm_warm is not None else None

    # ---- timed region: exactly K ticks.  After every tick: the device-side histogram of the statuses (one tiny kernel
    #      and a 72-byte copy, inside the timed region) and, when profiling, the per-iteration counts of unfinished instances.
    mpc.set_profiling(not args.no_profile, only=dom)
    per_tick_iters, per_tick_counts, per_tick_itersum, active_hist = [], [], [], []
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    per_tick_solver = []
    extra_inst_iters = 0.0  # instance-iterations of the parts whose launches are not logged (tick-level roofline)
    if split:
        # every part runs its K ticks at its own pace; after each of ITS ticks (in its thread): the status histograms and the
        # per-iteration counts of unfinished instances, exactly what the single handle records
        rec = [[None] * args.steps for _ in range(n_parts)]
        def after(pi, t):
            q = mpc.parts[pi]
            c, isum = q.status_counts()
            rec[pi][t] = (c, isum, q.solver_status_counts(), q.timing()["ip_iterations"], q.active_history() if not args.no_profile else None)
        run_split(args.steps, after_tick=after, timed=True)
        for t in range(args.steps):
            per_tick_counts.append(sum(rec[pi][t][0] for pi in range(n_parts)))
            per_tick_itersum.append(sum(rec[pi][t][1] for pi in range(n_parts)))
            per_tick_solver.append(sum(rec[pi][t][2] for pi in range(n_parts)))
            per_tick_iters.append(max(rec[pi][t][3] for pi in range(n_parts)))
            if not args.no_profile:
                active_hist.append(rec[0][t][4])
                extra_inst_iters += sum(mpc.parts[pi].B + float(rec[pi][t][4][:max(0, rec[pi][t][3] - 1)].sum()) for pi in range(1, n_parts))
    else:
        for _ in range(args.steps):
            tick(timed=True)
            c, isum = mpc.status_counts()
            per_tick_counts.append(c), per_tick_itersum.append(isum)
            per_tick_solver.append(mpc.solver_status_counts())
            per_tick_iters.append(mpc.timing()["ip_iterations"])
            if not args.no_profile:
                active_hist.append(mpc.active_history())
    torch.cuda.synchronize(dev)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    tm = prof.timing()
    log = prof.launch_log(with_iterations=True) if not args.no_profile else None
    mpc.set_profiling(False)

    counts = np.array(per_tick_counts, dtype=np.float64)  # (K, 8)
    scounts = np.array(per_tick_solver, dtype=np.float64)  # (K, 8): the solver's own statuses (before the node-0 rule)
    itersum = np.array(per_tick_itersum, dtype=np.float64)
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_ev])) if gather_ev else 0.0
    if world > 1:
        t = torch.tensor([elapsed, gather_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, gather_ms = float(t[0].item()), float(t[1].item())
        agg = torch.from_numpy(np.concatenate([counts.ravel(), scounts.ravel(), itersum])).to(dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        agg = agg.cpu().numpy()
        counts, scounts, itersum = agg[:counts.size].reshape(counts.shape), agg[counts.size:2 * counts.size].reshape(counts.shape), agg[2 * counts.size:]
    n_all = n_total
    converged_per_tick = counts[:, 0] + counts[:, 1]
    value = float(converged_per_tick.sum()) / elapsed
    attempted = n_all * args.steps / elapsed
    solver_value = float((scounts[:, 0] + scounts[:, 1]).sum()) / elapsed

    # ---- N > 1, weak-scaling headline: BASELINE config 4 as stated in the same run - 8192 instances in TOTAL over the N GPUs
    config4 = None
    if world > 1 and not args.total_batch:
        T4 = 8192
        l4, h4 = shard.shard_range(T4, rank, world)
        m4 = ltompc.BatchedMPC(tables, n_horizon=N, batch=h4 - l4, options=opts, device=local_rank)
        m4.set_stream(stream.cuda_stream)
        m4.set_poll_every(args.poll_every)
        x4 = torch.from_numpy(np.ascontiguousarray(ltompc.sample_x0(tables, T4, seed=ltompc.scenarios.SEED)[l4:h4])).to(dev)
        x4n, u4 = torch.empty_like(x4), torch.zeros(h4 - l4, 2, dtype=torch.float64, device=dev)
        m4.set_initial_guess_dev(x4.data_ptr())
        ev4, c4 = [], []
        for s4 in range(args.warmup + args.steps):
            if s4 == args.warmup:
                barrier(); torch.cuda.synchronize(dev)
                t4 = time.perf_counter()
            m4.make_step_dev(x4.data_ptr(), u4.data_ptr())
            m4.plant_step_dev(x4.data_ptr(), u4.data_ptr(), x4n.data_ptr(), PLANT_SUBSTEPS)
            x4, x4n = x4n, x4
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); shard.gather_rows(u4, T4, rank, world); e1.record(stream)
            if s4 >= args.warmup:
                ev4.append((e0, e1))
                c4.append(m4.status_counts()[0])
        torch.cuda.synchronize(dev); barrier()
        t4 = time.perf_counter() - t4
        r4 = torch.tensor([t4, float(np.mean([a.elapsed_time(b) for a, b in ev4]))], dtype=torch.float64, device=dev)
        dist.all_reduce(r4, op=dist.ReduceOp.MAX)
        a4 = torch.from_numpy(np.array(c4, dtype=np.float64)).to(dev)
        dist.all_reduce(a4, op=dist.ReduceOp.SUM)
        a4 = a4.cpu().numpy()
        config4 = {"workload": f"BASELINE config 4: batch={T4} in total, contiguous shards of {T4 // world} per GPU over {world} GPUs, horizon N={N}",
                   "scaling": "strong", "value": float((a4[:, 0] + a4[:, 1]).sum()) / float(r4[0].item()), "unit": "MPC solves/s",
                   "ms_per_step": 1e3 * float(r4[0].item()) / args.steps, "batch_per_gpu": h4 - l4, "total_batch": T4,
                   "gather_ms": float(r4[1].item()), "solved_frac_last_tick": float(a4[-1, 0] + a4[-1, 1]) / T4}
        m4.close()

    # ---- roofline (rank 0; HIP events on the launch stream, accumulated over the timed ticks)
    roofline = None
    if not args.no_profile and rank == 0:
        ms, ln = tm["ms"], tm["launches_by_kernel"]
        names = list(ms.keys())
        if dom is None:  # no warm-up ticks: every launch of the timed region was bracketed
            dom = max(ms, key=lambda k: ms[k])
        kind, width, lms, lit = log
        # unfinished instances a launch of iteration `it` of tick `tk` works on: those that passed the previous
        # iteration's termination test (all B in iteration 0), at most the launch width
        tick_of_launch = np.zeros(kind.size, dtype=np.int64)
        tk = 0
        for i in range(1, kind.size):
            if lit[i] < lit[i - 1]:
                tk += 1
            tick_of_launch[i] = tk
        def active_in(tk, it):
            a = active_hist[min(tk, len(active_hist) - 1)]
            return B_acc if it == 0 else (int(a[it - 1]) if it - 1 < len(a) else 0)
        sel = kind == names.index(dom)
        act = np.array([min(active_in(int(tick_of_launch[i]), int(lit[i])), int(width[i])) for i in np.where(sel)[0]], dtype=np.float64)
        avg_ms = float(lms[sel].mean())
        bytes_per_launch = float(act.mean()) * N * BYTES_BY_KERNEL[dom]
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": KERNEL_OF_CLASS[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms, "launches_timed": int(sel.sum()),
                    "active_instances_per_launch": float(act.mean()), "algorithmic_bytes_per_launch": bytes_per_launch,
                    "algorithmic_bytes_per_instance_interval": BYTES_BY_KERNEL[dom],
                    "kernel_ms_total": {k: round(v, 3) for k, v in (tm_warm or tm)["ms"].items()},
                    "launches": (tm_warm or tm)["launches_by_kernel"],
                    "kernel_ms_total_from": ("the last warm-up tick, every launch bracketed"
                                             if tm_warm else "the timed ticks, every launch bracketed"),
                    "timed_region_events": ("launches of " + KERNEL_OF_CLASS[dom] + " only") if tm_warm else "every launch"}
        pmc = load_pmc_traffic(KERNEL_OF_CLASS[dom], B_acc, N)
        roofline["accounting"] = (f"launches of ONE of the {n_parts} handles the batch is split into ({B_acc} instances; 'full width' = all of them), "
                                  "measured while the other handle's kernels run beside them" if split else f"the one handle of {B} instances")
        if pmc:
            roofline["traffic"] = pmc.get("traffic_avg_all_launches")
            roofline["traffic_source"] = pmc["source"]
        if dom == "riccati1":
            roofline["kernel_note"] = "class riccati1 = k_riccati1 (one wavefront per instance) and k_riccati1q (four, launches of at most 16 instances): both counted"
        # the whole tick against the same roof: every (instance, interval, iteration) moves 3104 algorithmic bytes
        inst_iters = float(np.mean([B_acc + float(a[:max(0, len(a) - 1)].sum()) for a in active_hist])) + extra_inst_iters / args.steps
        tick_bytes = inst_iters * N * BYTES_PER_STAGE_ITER
        ms_per_step = 1e3 * elapsed / args.steps
        roofline["tick"] = {"algorithmic_bytes_per_tick": tick_bytes, "instance_iterations_per_tick": inst_iters, "ms_per_step": ms_per_step,
                            "achieved": tick_bytes / (ms_per_step * 1e-3) / 1e9, "frac": tick_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # the widest launches of the dominant class (every instance of the batch still iterating or idle in its wavefront)
        full = sel & (width == B_acc)
        if full.any():
            fw_ms = float(lms[full].mean())
            fw_bytes = B_acc * N * BYTES_BY_KERNEL[dom]
            roofline["full_width"] = {"launches": int(full.sum()), "avg_launch_ms": fw_ms, "algorithmic_bytes_per_launch": fw_bytes,
                                      "achieved": fw_bytes / (fw_ms * 1e-3) / 1e9, "frac": fw_bytes / (fw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if pmc:
                roofline["full_width"]["traffic"] = pmc.get("traffic")
        # every class at FULL width (all B instances in the launch), from the warm-up tick whose launches were all bracketed:
        # what the thread-per-(interval, instance) kernels and the 8-instances-per-wavefront sweep reach when the chip is full
        if log_warm is not None:
            wk, ww, wms, _ = log_warm
            fwc = {}
            for ci, cname in enumerate(names):
                s3 = (wk == ci) & (ww == B_acc)
                if s3.any() and BYTES_BY_KERNEL[cname] > 0:
                    t_ms, by = float(wms[s3].mean()), B_acc * N * BYTES_BY_KERNEL[cname]
                    fwc[KERNEL_OF_CLASS[cname]] = {"launches": int(s3.sum()), "avg_launch_ms": round(t_ms, 4), "algorithmic_bytes_per_launch": by,
                                                   "achieved": round(by / (t_ms * 1e-3) / 1e9, 1), "frac": round(by / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            # (split handles: these launches shared the chip with the other handles' kernels - a statement about the mix, not about
            #  the kernels; the same table for ONE handle of all B instances with nothing beside it is added below from the side run)
            roofline["full_width_by_class_beside_the_other_handles" if split else "full_width_by_class"] = fwc
        by_width = {}
        for w in sorted(set(int(v) for v in width[sel]), reverse=True)[:12]:
            s2 = sel & (width == w)
            by_width[str(w)] = [int(s2.sum()), round(float(lms[s2].mean()), 4)]
        roofline["avg_launch_ms_by_width"] = by_width  # width -> [launches, avg ms]

    # ---- extras (rank 0, after the timed region)
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:  # (single-GPU runs only: the other ranks would wait or leave)
        # batch = 1 latency (second handle, same stream)
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
        # ... and with the tuned warm start (an option: previous solution shifted by one interval, barrier restarted at 1e-3, safeguarded)
        ot1 = ltompc.default_options(); ot1.max_iter, ot1.warm_shift, ot1.mu_init_warm = args.max_iter, 1, 1e-3
        m1 = ltompc.BatchedMPC(tables, n_horizon=N, batch=1, options=ot1, device=local_rank)
        m1.set_stream(stream.cuda_stream)
        x1 = ltompc.X0_REFERENCE[None].copy()
        m1.set_initial_guess(x1)
        u1 = m1.make_step(x1)
        x1 = m1.plant_step(x1, u1)
        torch.cuda.synchronize(dev)
        t_solve, its1 = 0.0, []
        for _ in range(nb):
            tb = time.perf_counter()
            u1 = m1.make_step(x1)
            t_solve += time.perf_counter() - tb
            its1.append(int(m1.iters[0]))
            x1 = m1.plant_step(x1, u1)
        extras["batch1_tuned_warm_start"] = {"options": {"warm_shift": 1, "mu_init_warm": 1e-3}, "ms_per_solve": 1e3 * t_solve / nb, "iters": its1}
        m1.close()
        # BASELINE config 5: closed loop from the reference's x0, horizon N = 60, the reference's hard track constraints,
        # until the horizon reaches the end of the tables (one lap minus the look-ahead)
        o60 = ltompc.default_options()
        m60 = ltompc.BatchedMPC(tables, n_horizon=60, batch=1, options=o60, device=local_rank)
        m60.set_stream(stream.cuda_stream)
        x60 = ltompc.X0_REFERENCE[None].copy()
        m60.set_initial_guess(x60)
        s_end = tables.s_max - 0.1 * 60 * 25.0
        t60, it60, n60, h60 = 0.0, 0, 0, {}
        while x60[0, 0] < s_end and n60 < 1500:
            tb = time.perf_counter()
            u60 = m60.make_step(x60)
            t60 += time.perf_counter() - tb
            h60[int(m60.status[0])] = h60.get(int(m60.status[0]), 0) + 1
            it60 += int(m60.iters[0])
            x60 = m60.plant_step(x60, u60)
            n60 += 1
        extras["closed_loop_lap_n60"] = {"constraints": "hard (reference)", "ticks": n60, "simulated_s": 0.1 * n60, "solve_wall_s": t60,
                                         "real_time_factor": 0.1 * n60 / t60, "s_reached_m": float(x60[0, 0]), "s_target_m": float(s_end),
                                         "status_histogram": {STATUS_NAMES[k]: v for k, v in sorted(h60.items())}, "ip_iters_mean": it60 / max(n60, 1)}
        m60.close()

        def side_run(o, label, note):
            mt = ltompc.BatchedMPC(tables, n_horizon=N, batch=B, options=o, device=local_rank)
            mt.set_stream(stream.cuda_stream)
            xt = torch.from_numpy(x0_host).to(dev)
            xtn, ut = torch.empty_like(xt), torch.zeros(B, 2, dtype=torch.float64, device=dev)
            mt.set_initial_guess_dev(xt.data_ptr())
            launched, conv = [], 0
            for s in range(args.warmup + args.steps):
                if s == args.warmup:
                    torch.cuda.synchronize(dev)
                    tt = time.perf_counter()
                mt.make_step_dev(xt.data_ptr(), ut.data_ptr())
                mt.plant_step_dev(xt.data_ptr(), ut.data_ptr(), xtn.data_ptr(), PLANT_SUBSTEPS)
                xt, xtn = xtn, xt
                if s >= args.warmup:
                    c, _ = mt.status_counts()
                    conv += int(c[0] + c[1])
                    launched.append(mt.timing()["ip_iterations"])
            torch.cuda.synchronize(dev)
            tt = time.perf_counter() - tt
            c, isum = mt.status_counts()
            extras[label] = {"options": note, "converged_solves_per_s": conv / tt, "ms_per_step": 1e3 * tt / args.steps,
                             "converged_frac_last_tick": float(c[0] + c[1]) / B, "ip_iters_mean_last_tick": isum / B,
                             "ip_iterations_launched_per_tick": launched}
            if label == "single_handle" and not args.no_profile:
                # one more tick with every launch bracketed: the kernel classes at FULL width (all B instances in the launch) with
                # nothing else on the GPU - the kernels' own efficiency, where the timed region's launches share the chip with the
                # other handle's kernels
                mt.set_profiling(True)
                mt.make_step_dev(xt.data_ptr(), ut.data_ptr())
                torch.cuda.synchronize(dev)
                wk, ww, wms = mt.launch_log()
                mt.set_profiling(False)
                fwc, names1 = {}, list(mt.timing()["ms"].keys())
                for ci, cname in enumerate(names1):
                    s3 = (wk == ci) & (ww == B)
                    if s3.any() and BYTES_BY_KERNEL.get(cname, 0) > 0:
                        t_ms, by = float(wms[s3].mean()), B * N * BYTES_BY_KERNEL[cname]
                        fwc[KERNEL_OF_CLASS[cname]] = {"launches": int(s3.sum()), "avg_launch_ms": round(t_ms, 4), "algorithmic_bytes_per_launch": by,
                                                       "achieved": round(by / (t_ms * 1e-3) / 1e9, 1), "frac": round(by / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                extras[label]["full_width_by_class"] = fwc
            mt.close()
        def side_run_split(o, label, note):
            """the headline's handle layout (SplitMPC, n_parts handles ticking at their own pace) with other options"""
            ms_ = ltompc.SplitMPC(tables, n_horizon=N, batch=B, n_parts=n_parts, options=o, device=local_rank)
            xt = torch.from_numpy(x0_host).to(dev)
            xtn, ut = torch.empty_like(xt), torch.zeros(B, 2, dtype=torch.float64, device=dev)
            torch.cuda.synchronize(dev)
            ms_.set_initial_guess_dev(xt.data_ptr())
            if args.warmup:
                ms_.run_ticks(xt.data_ptr(), ut.data_ptr(), xtn.data_ptr(), args.warmup, PLANT_SUBSTEPS)
                if args.warmup % 2:
                    xt, xtn = xtn, xt
            recs = [[None] * args.steps for _ in range(n_parts)]
            def after_s(pi, t):
                c, _ = ms_.parts[pi].status_counts()
                recs[pi][t] = (int(c[0] + c[1]), ms_.parts[pi].timing()["ip_iterations"])
            torch.cuda.synchronize(dev)
            tt = time.perf_counter()
            ms_.run_ticks(xt.data_ptr(), ut.data_ptr(), xtn.data_ptr(), args.steps, PLANT_SUBSTEPS, after_tick=after_s)
            torch.cuda.synchronize(dev)
            tt = time.perf_counter() - tt
            c, isum = ms_.status_counts()
            extras[label] = {"options": note, "converged_solves_per_s": sum(r[0] for q in recs for r in q) / tt, "ms_per_step": 1e3 * tt / args.steps,
                             "converged_frac_last_tick": float(c[0] + c[1]) / B, "ip_iters_mean_last_tick": isum / B,
                             "ip_iterations_launched_per_tick": [max(recs[pi][t][1] for pi in range(n_parts)) for t in range(args.steps)]}
            ms_.close()
        # the same K ticks per instance as a closed-loop ROLLOUT with free-running instances (ltompc_rollout_dev: converged
        # instances start their next tick inside the running batch; bit-identical controls, tests/test_gpu_parity.py)
        mr = ltompc.BatchedMPC(tables, n_horizon=N, batch=B, options=opts, device=local_rank)
        mr.set_stream(stream.cuda_stream)
        xr = torch.from_numpy(x0_host).to(dev)
        xrn, ur = torch.empty_like(xr), torch.zeros(B, 2, dtype=torch.float64, device=dev)
        mr.set_initial_guess_dev(xr.data_ptr())
        for s_ in range(args.warmup):
            mr.make_step_dev(xr.data_ptr(), ur.data_ptr())
            mr.plant_step_dev(xr.data_ptr(), ur.data_ptr(), xrn.data_ptr(), PLANT_SUBSTEPS)
            xr, xrn = xrn, xr
        sl = torch.full((B, args.steps), -1, dtype=torch.int32, device=dev)
        il = torch.zeros(B, args.steps, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        tr = time.perf_counter()
        info = mr.rollout_dev(xr.data_ptr(), args.steps, PLANT_SUBSTEPS, 0, sl.data_ptr(), il.data_ptr())
        torch.cuda.synchronize(dev)
        tr = time.perf_counter() - tr
        slh, ilh = sl.cpu().numpy(), il.cpu().numpy()
        passes = ilh.sum(1) + args.steps
        extras["closed_loop_rollout"] = {"what": f"{args.steps} ticks per instance after the same {args.warmup} warm-up ticks, instances free-running (no lockstep)",
                                         "converged_solves_per_s": float(np.isin(slh, (0, 1)).sum()) / tr, "wall_ms": 1e3 * tr,
                                         "converged_frac": float(np.isin(slh, (0, 1)).mean()), "ip_iterations_launched": info["iterations"],
                                         "passes_per_instance": {"median": float(np.median(passes)), "p99": float(np.percentile(passes, 99)), "max": int(passes.max())},
                                         "full_width_iterations_equivalent": float(passes.sum()) / B}
        mr.close()
        if split:  # the same workload on ONE handle of all B instances (what `value` was in rounds 1 and 2)
            side_run(opts, "single_handle", {"parts": 1})
            if roofline is not None and "full_width_by_class" in extras.get("single_handle", {}):
                roofline["full_width_by_class"] = extras["single_handle"]["full_width_by_class"]
                roofline["full_width_by_class_from"] = f"extras.single_handle: one handle of all {B} instances, nothing running beside its launches"
        # the same workload at round 1's iteration budget (150 instead of the reference's 1000)
        o150 = ltompc.default_options(); o150.max_iter, o150.soft_rho = 150, args.soft_rho
        side_run(o150, "max_iter_150", {"max_iter": 150})
        # ... and with the warm start tuned for MPC (extension, not the reference's solver settings): previous solution
        # shifted by one interval, barrier restarted at 1e-3 instead of IPOPT's 0.1.  Same NLP, same tolerance.
        to = ltompc.default_options(); to.max_iter, to.warm_shift, to.mu_init_warm = args.max_iter, 1, 1e-3
        side_run(to, "tuned_warm_start", {"warm_shift": 1, "mu_init_warm": 1e-3})
        if split:  # ... and in the headline's handle layout
            side_run_split(to, "tuned_warm_start_split", {"warm_shift": 1, "mu_init_warm": 1e-3, "parts": n_parts})

    # ---- CPU baseline: the oracle (a port of the same NLP + algorithm) on ALL host cores, on the states, warm starts and
    #      previous controls the GPU handle holds after the timed region (i.e. the tick the GPU would solve next)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (on rank 0 at N = 1 only)
        from oracle import oracle as orc
        ncores = os.cpu_count() or 1
        nthreads, share = host_cpu_share()
        S = args.cpu_sample if args.cpu_sample > 0 else min(B, max(256, 600 * nthreads))  # ~60 solves/s per core: ~10 s
        S = min(S, B)
        oo = orc.default_options()
        oo.max_iter, oo.soft_rho = args.max_iter, args.soft_rho
        O = orc.Oracle(tables.packed(), options=oo)
        it = mpc.iterate()                      # the GPU's last solutions: the oracle's warm start
        st = mpc.stats()
        xs = x.cpu().numpy()[:S]                # states after the last timed tick
        up = u.cpu().numpy()[:S]                # controls applied last (u_prev of the next solve)
        warm = {k: it[k][:S] for k in ("X", "C", "U", "L1", "L2")}
        tc = time.perf_counter()
        r2 = O.solve(xs, N, uprev=up, warm=warm, nthreads=nthreads, prev_status=st["status"][:S])
        tcpu = time.perf_counter() - tc
        conv = int(np.isin(r2["status"], (0, 1)).sum())
        cpu = {"value": conv / tcpu, "unit": "MPC solves/s", "cores": nthreads, "kind": "port",
               "sample": f"{S} instances of the same batch: the tick after the timed region (the GPU's states, warm starts and u_prev), N={N}, "
                         f"OpenMP over instances, {nthreads} threads = all the CPU time this process may use ({share}; os.cpu_count() = {ncores}), "
                         f"{tcpu:.1f} s wall; converged solves counted; "
                         "oracle/ltompc_oracle.c (do_mpc/IPOPT itself is not installable offline)",
               "attempted_per_s": S / tcpu, "iters_mean": float(r2["iters"].mean()), "converged_frac": conv / S}
        # the reference's own mode of operation is one process, one instance at a time: same oracle, one thread
        S1 = min(64, S)
        w1 = {k: warm[k][:S1] for k in warm}
        tc = time.perf_counter()
        r1 = O.solve(xs[:S1], N, uprev=up[:S1], warm=w1, nthreads=1, prev_status=st["status"][:S1])
        cpu["single_thread_value"] = int(np.isin(r1["status"], (0, 1)).sum()) / (time.perf_counter() - tc)
        cpu["single_thread_sample"] = f"first {S1} instances of the same tick, 1 thread"

    if rank == 0:
        hist_last = {STATUS_NAMES[k]: int(v) for k, v in enumerate(counts[-1]) if v}
        out = {
            "metric": "MPC solves/sec (N=40, nx=8 [7 + progress s], nu=2)", "value": value, "unit": "MPC solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.total_batch else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL: all ranks on one GPU, gloo collectives - not a measurement",
            "config": {"workload": (f"batch={n_total} in TOTAL sharded over {world} GPU(s) ({B} on rank 0)" if args.total_batch else f"batch={B} per GPU x {world} GPU")
                                   + (f" (as {n_parts} handles of {B_acc}, each on its own stream, ticking at its own pace)" if split else "")
                                   + f", horizon N={N}, closed-loop warm ticks "
                                   f"(buckmore / MX-5 / curvature tables, x0 sampled along the lap, seed {ltompc.scenarios.SEED})",
                       "batch_per_gpu": B, "total_batch": n_total, "parts_per_gpu": n_parts, "horizon": N, "max_iter": args.max_iter, "tol": opts.tol, "soft_rho": args.soft_rho,
                       "resto_rho": opts.resto_rho, "resto_sticky": args.resto_sticky, "parallelism": f"instances sharded over {world} GPU(s), no data-path collective; controls gathered per tick (RCCL all_gather)" if world > 1 else "1 GPU",
                       "resto_rho_max": opts.resto_rho_max, "dual_inf_max": opts.dual_inf_max, "node0_check": opts.node0_check, "resto_shift_retry": opts.resto_shift_retry},
            "value_counts": "converged solves only (status solved / acceptable, after the node-0 rule: an instance whose measured state is outside "
                            "the track band counts as INFEASIBLE like the reference's NLP, whatever its solve did)",
            "solves_attempted_per_s": attempted,
            "solver_converged_solves_per_s": solver_value,
            "solver_status_histogram_last_tick": {STATUS_NAMES[k]: int(v) for k, v in enumerate(scounts[-1]) if v},
            "gather_ms": gather_ms, "config4_total_8192": config4,
            "solved_frac_per_tick": [round(float(v) / n_all, 5) for v in converged_per_tick],
            "solved_frac_last_tick": float(converged_per_tick[-1]) / n_all,
            "status_histogram_last_tick": hist_last,
            "ip_iters_mean_per_tick": [round(float(v) / n_all, 3) for v in itersum],
            "ip_iterations_launched_per_tick": per_tick_iters,
            "roofline": roofline, "cpu_baseline": cpu, "extras": extras,
        }
        print(json.dumps(out))
    mpc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
