"""N > 1 path on CPU: two gloo ranks shard a batch (contiguous blocks, no data-path collective), each solves its
shard, results are gathered and must equal the single-process solve of the whole batch bit for bit."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

WORKER = textwrap.dedent('''
    import os, sys, importlib
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["LTOMPC_ROOT"])
    pkg = importlib.import_module("lap-time-optimization_amd")
    shard = importlib.import_module("lap-time-optimization_amd.sharding")
    from oracle import oracle as orc
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    tables = pkg.TrackTables.load_npz(os.path.join(os.environ["LTOMPC_ROOT"], "tests", "golden", "tables_buckmore_mx5_curvature.npz"))
    B, N = 7, 10                                      # ragged: 7 instances over 2 ranks
    x0 = pkg.sample_x0(tables, B, seed=5)
    lo, hi = shard.shard_range(B, rank, world)
    r = orc.Oracle(tables.packed()).solve(x0[lo:hi], N, nthreads=1)   # stands in for the per-rank GPU solve
    u0 = shard.gather_rows(torch.from_numpy(r["u0"]), B, rank, world)
    stats = shard.reduce_stats(int(r["iters"].max()), int((r["status"] != 0).sum()), 0.0)
    if rank == 0:
        np.save(os.environ["LTOMPC_OUT"], u0.numpy())
        print("STATS", stats)
    dist.destroy_process_group()
''')


def test_two_rank_sharding_matches_single_process(tmp_path, oracle, pkg, tables):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "u0.npy"
    env = dict(os.environ, LTOMPC_ROOT=ROOT, LTOMPC_OUT=str(out), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env, timeout=300)
    u0 = np.load(out)
    ref = oracle.solve(pkg.sample_x0(tables, 7, seed=5), 10, nthreads=1)
    assert u0.shape == (7, 2) and np.array_equal(u0, ref["u0"])


def test_shard_ranges_cover_the_batch(pkg):
    shard = __import__("importlib").import_module("lap-time-optimization_amd.sharding")
    for B in (1, 7, 8, 8192, 8193):
        for world in (1, 2, 3, 8):
            r = [shard.shard_range(B, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launcher_starts_the_ranks_itself(tmp_path):
    """`python bench.py --gpus 2 ...` (the driver's command, no torchrun around it): bench.py starts the two ranks as a child
    process group, every rank joins the rendezvous, the barrier-bracketed time is the max over ranks, the counts are summed
    over ranks, and rank 0's one JSON line comes back through the parent with n_gpus = world size.  The solver is replaced by
    a stand-in (--selftest-stub, gloo): this covers the plumbing, not the numbers."""
    import json
    env = dict(os.environ, MASTER_PORT="29547", OMP_NUM_THREADS="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "100",
                          "--selftest-stub"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["data"] == "stub"
    assert r["converged_solves"] == 3 * (100 + 99)            # summed over the ranks (rank 1 "fails" one instance per tick)
    assert r["ms_per_step"] >= 4.0                              # the slower rank (2 x 2 ms per tick) sets the time
    assert r["value"] == pytest.approx(r["converged_solves"] / (r["ms_per_step"] * 3e-3), rel=1e-9)
    # the same command also measured BASELINE config 4's shape (8192 in TOTAL over the ranks: strong scaling), with the per-tick
    # gather of the controls over the process group
    c4 = r["config4_total_8192"]
    assert r["scaling"] == "weak" and c4["scaling"] == "strong" and c4["total_batch"] == 8192 and c4["batch_per_gpu"] == 4096
    assert c4["converged_solves"] == 3 * (4096 + 4095)
    assert r["gathered_checksum"] == sum(2.0 * (sum(range(200)) + 200 * k) for k in range(3))     # every rank's rows arrived, in order
    # --total-batch makes that shape the headline
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--total-batch", "101",
                          "--selftest-stub"], env=dict(env, MASTER_PORT="29549"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r2 = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r2["scaling"] == "strong" and r2["config"]["total_batch"] == 101 and r2["config"]["batch_per_gpu"] == 51 and "config4_total_8192" not in r2
    assert r2["converged_solves"] == 2 * (51 + 49)
    # one rank: no child process, same line format
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--batch", "10", "--selftest-stub"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def _fake_parts(pipe, bounds, n_ticks, ring, rng_seed, log=None):
    """Stand-in for SplitMPC.run_ticks: every part in its own thread, random tick lengths, tick t of part p writes the value
    1000 t + row to its rows of ring[t % depth] - with the pipeline's hooks around each tick."""
    import threading, time
    def body(pi, lo, hi):
        rng = np.random.default_rng(rng_seed + pi)
        try:
            for t in range(n_ticks):
                pipe.before_tick(pi, t)
                if log is not None:
                    log.append(("start", pi, t))
                time.sleep(float(rng.uniform(0.0, 0.004)))
                ring[t % len(ring)][lo:hi, 0] = 1000.0 * t + np.arange(lo, hi)
                ring[t % len(ring)][lo:hi, 1] = -(1000.0 * t + np.arange(lo, hi))
                pipe.after_tick(pi, t)
        except BaseException as e:
            pipe.fail(e)
    th = [threading.Thread(target=body, args=(pi, lo, hi)) for pi, (lo, hi) in enumerate(bounds)]
    for q in th:
        q.start()
    return th


def test_tick_pipeline_hands_every_tick_over_exactly_once(pkg):
    """Free-running parts beside a per-tick consumer (bench.py with several ranks: the gather of the controls): the consumer sees
    tick t's rows of ALL parts, complete and not yet overwritten, in tick order; no part starts tick t + depth before tick t has
    been consumed."""
    shard = __import__("importlib").import_module("lap-time-optimization_amd.sharding")
    for depth in (1, 2, 3):
        B, n_ticks, bounds = 23, 40, [(0, 6), (6, 12), (12, 18), (18, 23)]
        ring = [np.full((B, 2), np.nan) for _ in range(depth)]
        pipe = shard.TickPipeline(len(bounds), depth=depth)
        log, seen = [], []
        th = _fake_parts(pipe, bounds, n_ticks, ring, 3, log)
        def consume(t):
            seen.append(ring[t % depth].copy())
            log.append(("consumed", -1, t))
        assert pipe.consume(n_ticks, consume)
        for q in th:
            q.join()
        assert pipe.failure is None and len(seen) == n_ticks
        for t, a in enumerate(seen):
            assert np.array_equal(a[:, 0], 1000.0 * t + np.arange(B)) and np.array_equal(a[:, 1], -a[:, 0])
        consumed_at = {t: i for i, (k, _, t) in enumerate(log) if k == "consumed"}
        for i, (k, pi, t) in enumerate(log):
            if k == "start" and t >= depth:
                assert consumed_at[t - depth] < i, (depth, pi, t)


def test_tick_pipeline_failure_wakes_both_sides(pkg):
    shard = __import__("importlib").import_module("lap-time-optimization_amd.sharding")
    import threading
    # the consumer fails: the parts stop waiting
    pipe = shard.TickPipeline(2, depth=1)
    ring = [np.zeros((4, 2))]
    th = _fake_parts(pipe, [(0, 2), (2, 4)], 50, ring, 0)
    def boom(t):
        if t == 3:
            raise ValueError("gather failed")
    with pytest.raises(ValueError):
        pipe.consume(50, boom)
    for q in th:
        q.join(timeout=20)
    assert not any(q.is_alive() for q in th) and isinstance(pipe.failure, ValueError)
    # a part fails: the consumer returns False instead of waiting for ever
    pipe = shard.TickPipeline(2, depth=2)
    def part1():
        try:
            pipe.before_tick(1, 0)
            raise RuntimeError("solver failed")
        except BaseException as e:
            pipe.fail(e)
    q = threading.Thread(target=part1); q.start()
    pipe.after_tick(0, 0)
    assert pipe.consume(5, lambda t: None) is False
    q.join()
    assert isinstance(pipe.failure, RuntimeError)


PIPE_WORKER = textwrap.dedent('''
    import os, sys, importlib, threading, time
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["LTOMPC_ROOT"])
    shard = importlib.import_module("lap-time-optimization_amd.sharding")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    B, K, P = 21, 12, 3                               # ragged batch, 12 ticks, 3 free-running parts per rank
    lo, hi = shard.shard_range(B, rank, world)
    n = hi - lo
    ring = [torch.full((n, 2), float("nan"), dtype=torch.float64) for _ in range(2)]
    pipe = shard.TickPipeline(P, depth=2)
    cuts = [round(i * n / P) for i in range(P + 1)]
    def part(pi):
        rng = np.random.default_rng(100 * rank + pi)
        try:
            for t in range(K):
                pipe.before_tick(pi, t)
                time.sleep(float(rng.uniform(0, 0.01)) * (1 + rank))    # rank 1 is the slower one
                rows = torch.arange(lo + cuts[pi], lo + cuts[pi + 1], dtype=torch.float64)
                ring[t % 2][cuts[pi]:cuts[pi + 1], 0] = 1000.0 * t + rows
                ring[t % 2][cuts[pi]:cuts[pi + 1], 1] = rows - 1000.0 * t
                pipe.after_tick(pi, t)
        except BaseException as e:
            pipe.fail(e)
    th = [threading.Thread(target=part, args=(pi,)) for pi in range(P)]
    for q in th: q.start()
    got = []
    ok = pipe.consume(K, lambda t: got.append(shard.gather_rows(ring[t % 2], B, rank, world).clone()))
    for q in th: q.join()
    assert ok and pipe.failure is None
    np.save(os.environ["LTOMPC_OUT"] + f".{rank}.npy", torch.stack(got).numpy())
    dist.destroy_process_group()
''')


def test_two_ranks_gather_every_tick_beside_free_running_parts(tmp_path):
    """The shape bench.py runs with several GPUs and several handles per GPU: on each of two gloo ranks three parts tick at their
    own pace while the rank's main thread gathers tick t's rows of the whole batch as soon as its own parts have finished tick t
    (sharding.TickPipeline + gather_rows).  Every rank must end up with every tick's full array."""
    script = tmp_path / "pipe_worker.py"
    script.write_text(PIPE_WORKER)
    out = tmp_path / "ring"
    env = dict(os.environ, LTOMPC_ROOT=ROOT, LTOMPC_OUT=str(out), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29549", str(script)], env=env, timeout=300)
    rows = np.arange(21, dtype=np.float64)
    for rank in (0, 1):
        a = np.load(str(out) + f".{rank}.npy")
        assert a.shape == (12, 21, 2)
        for t in range(12):
            assert np.array_equal(a[t, :, 0], 1000.0 * t + rows) and np.array_equal(a[t, :, 1], rows - 1000.0 * t)
