"""BatchedMPC: thin Python owner of one ltompc handle (B independent MPC instances on one MI355X)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import NX, NU, Options, Params, check, dptr, iptr, lib
from .tables import TrackTables

KERNEL_CLASSES = ("eval", "riccati", "expand", "linesearch", "pick", "update", "riccati1", "step1")  # include/ltompc.h


class BatchedMPC:
    """B receding-horizon NLPs of the reference's controller (src/mpc/controller.py:9-103), solved on the GPU.

    make_step(x0 (B,8)) -> u0 (B,2): the batched form of `controller.mpc.make_step(x0)` (src/mpc.py:142).
    """

    def __init__(self, tables: TrackTables, n_horizon: int = 10, batch: int = 1, params: Params | None = None,
                 options: Options | None = None, device: int = 0):
        self.tables = tables
        self.N, self.B = int(n_horizon), int(batch)
        self.params = params or _lib.default_params()
        self.options = options or _lib.default_options()
        self._tab = tables.packed()
        self._h = C.c_void_p()
        check(lib().ltompc_create(C.byref(self.params), C.byref(self.options), dptr(self._tab), self._tab.shape[1],
                                  self.N, self.B, int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().ltompc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference surface, batched -------------------------------------------------------------
    def set_initial_guess(self, x0):
        x0 = self._x(x0)
        check(lib().ltompc_set_initial_guess(self._h, dptr(x0)))

    def make_step(self, x0):
        x0 = self._x(x0)
        u0 = np.empty((self.B, NU))
        self.status = np.empty(self.B, dtype=np.int32)
        self.iters = np.empty(self.B, dtype=np.int32)
        check(lib().ltompc_make_step(self._h, dptr(x0), dptr(u0), iptr(self.status), iptr(self.iters)))
        return u0

    # ---- device-pointer variants (bench, closed loop on the GPU) --------------------------------
    def set_stream(self, hip_stream: int | None):
        check(lib().ltompc_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def set_initial_guess_dev(self, x0_ptr: int):
        check(lib().ltompc_set_initial_guess_dev(self._h, C.c_void_p(x0_ptr)))

    def make_step_dev(self, x0_ptr: int, u0_ptr: int):
        check(lib().ltompc_make_step_dev(self._h, C.c_void_p(x0_ptr), C.c_void_p(u0_ptr)))

    def rollout_dev(self, x_ptr: int, n_ticks: int, n_sub: int = 400, u_log_ptr: int = 0, status_log_ptr: int = 0, iters_log_ptr: int = 0):
        """Closed-loop rollout with free-running instances (ltompc_rollout_dev): n_ticks of make_step + plant step per instance."""
        check(lib().ltompc_rollout_dev(self._h, C.c_void_p(x_ptr), int(n_ticks), int(n_sub), C.c_void_p(u_log_ptr or None),
                                       C.c_void_p(status_log_ptr or None), C.c_void_p(iters_log_ptr or None)))
        it, ln = C.c_longlong(), C.c_longlong()
        check(lib().ltompc_rollout_info(self._h, C.byref(it), C.byref(ln)))
        return dict(iterations=it.value, launches=ln.value)

    def plant_step_dev(self, x_ptr: int, u_ptr: int, xn_ptr: int, n_sub: int = 400):
        check(lib().ltompc_plant_step_dev(self._h, C.c_void_p(x_ptr), C.c_void_p(u_ptr), int(n_sub), C.c_void_p(xn_ptr)))

    # ---- results --------------------------------------------------------------------------------
    def prediction(self):
        X, U = np.empty((self.B, self.N + 1, NX)), np.empty((self.B, self.N, NU))
        check(lib().ltompc_get_prediction(self._h, dptr(X), dptr(U)))
        return X, U

    def iterate(self):
        X, U = np.empty((self.B, self.N + 1, NX)), np.empty((self.B, self.N, NU))
        Cc, L1, L2 = (np.empty((self.B, self.N, NX)) for _ in range(3))
        check(lib().ltompc_get_iterate(self._h, dptr(X), dptr(Cc), dptr(U), dptr(L1), dptr(L2)))
        ni = C.c_int()
        check(lib().ltompc_get_ineq(self._h, None, None, C.byref(ni)))
        Tt, Nu = np.empty((self.B, self.N, ni.value)), np.empty((self.B, self.N, ni.value))
        check(lib().ltompc_get_ineq(self._h, dptr(Tt), dptr(Nu), C.byref(ni)))
        return dict(X=X, C=Cc, U=U, L1=L1, L2=L2, T=Tt, NU=Nu)

    def stats(self):
        st, it = np.empty(self.B, dtype=np.int32), np.empty(self.B, dtype=np.int32)
        kkt, obj, mu = np.empty(self.B), np.empty(self.B), np.empty(self.B)
        check(lib().ltompc_get_stats(self._h, iptr(st), iptr(it), dptr(kkt), dptr(obj), dptr(mu)))
        nr, nf = np.empty(self.B, dtype=np.int32), np.empty(self.B, dtype=np.int32)
        check(lib().ltompc_get_counters(self._h, iptr(nr), iptr(nf)))
        nre, viol = np.empty(self.B, dtype=np.int32), np.empty(self.B)
        check(lib().ltompc_get_restoration(self._h, iptr(nre), dptr(viol)))
        nsh, nfb, sst = (np.empty(self.B, dtype=np.int32) for _ in range(3))
        g0, pen = np.empty(self.B), np.empty(self.B)
        check(lib().ltompc_get_recovery(self._h, iptr(nsh), iptr(nfb), dptr(g0), iptr(sst), dptr(pen)))
        return dict(status=st, iters=it, kkt=kkt, obj=obj, mu=mu, n_reg=nr, n_lsfail=nf, n_resto=nre, viol=viol,
                    n_shift=nsh, n_fallback=nfb, g0=g0, status_solver=sst, penalty=pen)

    def plant_step(self, x, u, n_sub: int = 400):
        x, u = self._x(x), np.ascontiguousarray(np.asarray(u, float).reshape(self.B, NU))
        xn = np.empty_like(x)
        check(lib().ltompc_plant_step(self._h, dptr(x), dptr(u), int(n_sub), dptr(xn)))
        return xn

    def slip_forces(self, x):
        x = np.ascontiguousarray(np.asarray(x, float).reshape(-1, NX))
        a, F = np.empty((x.shape[0], 2)), np.empty((x.shape[0], 2))
        check(lib().ltompc_slip_forces(self._h, dptr(x), x.shape[0], dptr(a), dptr(F)))
        return a, F

    def synchronize(self):
        check(lib().ltompc_synchronize(self._h))

    def set_profiling(self, on, only: str | None = None):
        """on: False / True (every launch).  only='eval' | 'riccati' | ...: bracket the launches of that kernel class only."""
        mode = int(bool(on))
        if on and only is not None:
            mode = 2 + KERNEL_CLASSES.index(only)
        check(lib().ltompc_set_profiling(self._h, mode))

    def set_poll_every(self, n: int):
        check(lib().ltompc_set_poll_every(self._h, int(n)))

    def set_narrow_width(self, width: int):
        """Widest launch (unfinished instances) that uses the one-instance-per-workgroup kernels (default 512); scheduling only."""
        check(lib().ltompc_set_narrow_width(self._h, int(width)))

    def timing(self):
        ms, ln = np.zeros(8), np.zeros(8, dtype=np.int32)
        launches, its = C.c_int(), C.c_int()
        check(lib().ltompc_get_timing(self._h, dptr(ms), iptr(ln), C.byref(launches), C.byref(its)))
        names = KERNEL_CLASSES
        return dict(ms={n: float(m) for n, m in zip(names, ms)}, launches_by_kernel={n: int(v) for n, v in zip(names, ln)},
                    launches=launches.value, ip_iterations=its.value)

    def launch_log(self, with_iterations: bool = False):
        """(kind, width, ms[, iteration]) arrays of every kernel launch of the profiled make_steps (set_profiling(True))."""
        L = lib()
        n = L.ltompc_get_launch_log(self._h, None, None, None, 0)
        kind, width, ms = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32), np.empty(n)
        L.ltompc_get_launch_log(self._h, iptr(kind), iptr(width), dptr(ms), n)
        if not with_iterations:
            return kind, width, ms
        it = np.empty(n, dtype=np.int32)
        L.ltompc_get_launch_log_iterations(self._h, iptr(it), n)
        return kind, width, ms, it

    def active_history(self):
        """Unfinished instances after every interior-point iteration of the last make_step."""
        n = lib().ltompc_get_active_history(self._h, None, 0)
        a = np.zeros(max(n, 1), dtype=np.int32)
        lib().ltompc_get_active_history(self._h, iptr(a), n)
        return a[:n]

    def status_counts(self):
        """(histogram of the statuses of the last solve [8], sum of the iteration counts), reduced on the device."""
        c, s = np.zeros(8, dtype=np.int32), C.c_longlong()
        check(lib().ltompc_get_status_counts(self._h, iptr(c), C.byref(s)))
        return c, int(s.value)

    def solver_status_counts(self):
        """Histogram [8] of the solver's own statuses (before the node-0 rule), as reduced by the last status_counts() call."""
        c = np.zeros(8, dtype=np.int32)
        check(lib().ltompc_get_solver_status_counts(self._h, iptr(c)))
        return c

    def history(self):
        buf = np.zeros((4096, 3), dtype=np.int32)
        n = lib().ltompc_get_history(self._h, iptr(buf), 4096)
        return buf[:max(0, min(n, 4096))]

    def debug_fetch(self, which: int):
        L = lib(); L.ltompc_debug_fetch.restype = C.c_longlong
        n = L.ltompc_debug_fetch(self._h, int(which), None, C.c_longlong(0))
        buf = np.empty(n // (4 if which == 13 else 8), dtype=np.int32 if which == 13 else np.float64)
        L.ltompc_debug_fetch(self._h, int(which), buf.ctypes.data_as(C.c_void_p), C.c_longlong(n))
        return buf

    def test_model(self, x, lam, eps: float = 0.0):
        x = np.ascontiguousarray(np.asarray(x, float).reshape(-1, NX))
        lam = np.ascontiguousarray(np.asarray(lam, float).reshape(-1, NX))
        n = x.shape[0]
        out = dict(f=np.empty((n, 8)), J=np.empty((n, 8, 8)), H=np.empty((n, 8, 8)), cval=np.empty((n, 2)),
                   cgrad=np.empty((n, 2, 8)), cH=np.empty((n, 2, 8, 8)), gval=np.empty((n, 3)),
                   ggrad=np.empty((n, 3, 8)), gH=np.empty((n, 3, 8, 8)))
        check(lib().ltompc_test_model(self._h, n, C.c_double(eps), dptr(x), dptr(lam),
                                      *(dptr(out[k]) for k in ("f", "J", "H", "cval", "cgrad", "cH", "gval", "ggrad", "gH"))))
        return out

    def test_ellipse(self, x):
        x = np.ascontiguousarray(np.asarray(x, float).reshape(-1, NX))
        n = x.shape[0]
        v, g, H = np.empty((n, 2)), np.empty((n, 2, 8)), np.empty((n, 2, 8, 8))
        check(lib().ltompc_test_ellipse(self._h, n, dptr(x), dptr(v), dptr(g), dptr(H)))
        return v, g, H

    def _x(self, x0):
        x0 = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(-1, NX))
        if x0.shape[0] != self.B:
            raise ValueError(f"expected {self.B} states of dimension {NX}, got array of shape {x0.shape}")
        return x0


class SplitMPC:
    """The batch as `n_parts` handles of batch / n_parts instances, each on its own HIP stream and driven by its own host thread
    (ctypes releases the GIL during a call).  Instances are independent NLPs, so the parts need not tick together: while one
    part is in the narrow tail of its tick - the 1 % of its instances that need 2 - 10 times the iterations of the rest, a chain
    of one-wavefront launches on an otherwise idle chip - the other part's full-width launches fill the GPU.  Results are the
    single handle's, bit for bit (an instance's result does not depend on the batch it is solved in); 8192 instances at N = 40:
    134 k solves/s with two parts and 142 k with four against 124 k (five or more lose: the HIP runtime serves a process's streams
    with four hardware queues, and a narrow launch then waits behind another part's full-width launches in the same queue).

    The device-pointer interface of BatchedMPC for contiguous row blocks: part p owns rows [lo_p, hi_p) of every (B, .) array."""

    def __init__(self, tables: TrackTables, n_horizon: int = 10, batch: int = 1, n_parts: int = 4, params: Params | None = None,
                 options: Options | None = None, device: int = 0, narrow_width: int | None = None):
        from concurrent.futures import ThreadPoolExecutor
        self.N, self.B, self.n_parts = int(n_horizon), int(batch), int(n_parts)
        base, rem = divmod(self.B, self.n_parts)
        self.bounds = []
        lo = 0
        for p in range(self.n_parts):
            hi = lo + base + (1 if p < rem else 0)
            self.bounds.append((lo, hi)); lo = hi
        self.parts = [BatchedMPC(tables, n_horizon, hi - lo, params=params, options=options, device=device) for lo, hi in self.bounds]
        self.options, self.params = self.parts[0].options, self.parts[0].params
        # three or more parts beside each other: their one-instance workgroups queue for the same CUs, so the parts switch to the
        # one-instance kernels later (ltompc_set_narrow_width; scheduling only, same bits): 128 instead of 512, +2 % with four parts
        if narrow_width is None and self.n_parts >= 3:
            narrow_width = 128
        if narrow_width is not None:
            for p in self.parts:
                p.set_narrow_width(narrow_width)
        self._pool = ThreadPoolExecutor(max_workers=self.n_parts)

    def close(self):
        for p in self.parts:
            p.close()
        self._pool.shutdown(wait=True)

    def _each(self, fn):
        """fn(part, lo, hi) on every part, each in its own host thread; returns the results in part order."""
        futs = [self._pool.submit(fn, p, lo, hi) for p, (lo, hi) in zip(self.parts, self.bounds)]
        return [f.result() for f in futs]

    def set_initial_guess_dev(self, x0_ptr: int):
        self._each(lambda p, lo, hi: (p.set_initial_guess_dev(x0_ptr + 8 * NX * lo), p.synchronize()))

    def make_step_dev(self, x0_ptr: int, u0_ptr: int):
        self._each(lambda p, lo, hi: p.make_step_dev(x0_ptr + 8 * NX * lo, u0_ptr + 8 * NU * lo))

    def run_ticks(self, x_ptr: int, u_ptr, xn_ptr: int, n_ticks: int, n_sub: int = 400, after_tick=None, before_tick=None):
        """n_ticks of the closed loop [make_step; plant step] for every part at its own pace: x (B, 8) and xn (B, 8) are swapped
        after every tick (the states end in x if n_ticks is even, else in xn), u (B, 2) holds the last controls.  u_ptr may be a
        sequence of pointers: tick t then writes its controls to u_ptr[t % len(u_ptr)] (a ring: somebody else - a gather over the
        ranks, a logger - reads the controls of tick t while the parts are one tick further).
        before_tick(part_index, tick) / after_tick(part_index, tick) run in the part's thread around each of its ticks (before_tick
        may block: that is how a consumer of the ring holds a part back).  Returns when all parts are done."""
        ring = [int(u_ptr)] if isinstance(u_ptr, int) else [int(q) for q in u_ptr]
        def body(p, lo, hi):
            a, b = x_ptr + 8 * NX * lo, xn_ptr + 8 * NX * lo
            pi = self.parts.index(p)
            for t in range(n_ticks):
                if before_tick is not None:
                    before_tick(pi, t)
                u = ring[t % len(ring)] + 8 * NU * lo
                p.make_step_dev(a, u)
                p.plant_step_dev(a, u, b, n_sub)
                a, b = b, a
                if after_tick is not None:
                    after_tick(pi, t)
            p.synchronize()
        self._each(body)

    def rollout_dev(self, x_ptr: int, n_ticks: int, n_sub: int = 400, u_log_ptr: int = 0, status_log_ptr: int = 0, iters_log_ptr: int = 0):
        """BatchedMPC.rollout_dev on every part at once (the logs are (B, n_ticks, .) arrays: part p fills its rows).  Returns the
        longest chain of passes over the parts and the launches of all of them."""
        def body(p, lo, hi):
            return p.rollout_dev(x_ptr + 8 * NX * lo, n_ticks, n_sub, u_log_ptr + 8 * NU * n_ticks * lo if u_log_ptr else 0,
                                 status_log_ptr + 4 * n_ticks * lo if status_log_ptr else 0, iters_log_ptr + 4 * n_ticks * lo if iters_log_ptr else 0)
        r = self._each(body)
        return dict(iterations=max(q["iterations"] for q in r), launches=sum(q["launches"] for q in r), per_part=r)

    def synchronize(self):
        for p in self.parts:
            p.synchronize()

    def set_poll_every(self, n: int):
        for p in self.parts:
            p.set_poll_every(n)

    def set_profiling(self, on, only: str | None = None):
        for p in self.parts:
            p.set_profiling(on, only)

    def status_counts(self):
        r = [p.status_counts() for p in self.parts]
        return sum(c for c, _ in r), sum(s for _, s in r)

    def solver_status_counts(self):
        return sum(p.solver_status_counts() for p in self.parts)

    def stats(self):
        r = [p.stats() for p in self.parts]
        return {k: np.concatenate([q[k] for q in r]) for k in r[0]}

    def iterate(self):
        r = [p.iterate() for p in self.parts]
        return {k: np.concatenate([q[k] for q in r]) for k in r[0]}
