"""Oracle pinning (2): the NLP solve.  The reference's own solver stack (do_mpc 4.6.5 / CasADi 3.6.6 / IPOPT) is not
installable offline and the reference ships no make_step known-answer (its recorded `u` predates the current
objective), so make_step parity is pinned by
  (a) the independent SLSQP solves recorded in SURVEY.md §8(c) ("PROBE anchors"),
  (b) an independent torch-autograd evaluation of the KKT conditions of the NLP at the returned point."""
import numpy as np
import pytest
import torch

import nlp_reference as R

X0 = np.array([[0, 0, 0, 5, 0, 0, 0, 0.1]], dtype=float)  # src/mpc.py:107-110
# N -> (u0, J)
ANCHORS = {10: ((-0.02410604, 1.0), 708.8701288), 20: ((-0.06380024, 1.0), 1303.5282197),
           40: ((-0.06404819, 1.0), 2147.0976789)}


@pytest.mark.parametrize("N", [10, 20, 40])
def test_probe_anchors(oracle, N):
    r = oracle.solve(X0, N)
    (u_s, u_t), J = ANCHORS[N]
    assert r["status"][0] == 0
    assert r["u0"][0, 0] == pytest.approx(u_s, abs=1e-6) and r["u0"][0, 1] == pytest.approx(u_t, abs=1e-6)
    assert r["obj"][0] == pytest.approx(J, rel=1e-8)


def test_probe_terminal_states(oracle):
    r = oracle.solve(X0, 20)
    xN = r["X"][0, -1]
    # (the terminal throttle is an almost flat direction of the objective: looser there)
    assert xN[:7] == pytest.approx([11.30909, 0.10319, 0.02526, 6.59481, -0.00562, -0.00449, -0.00199], abs=2e-5)
    assert xN[7] == pytest.approx(1.0, abs=2e-4)
    dT = r["U"][0, :, 1]
    assert np.all(dT[:9] > 1 - 1e-6) and np.all(np.abs(dT[9:-1]) < 1e-5) and abs(dT[-1]) < 2e-3  # throttle ramps at +1 then saturates at T = 1
    r = oracle.solve(X0, 40)
    assert r["X"][0, -1, :7] == pytest.approx([26.49635, -0.43764, -0.02401, 8.59309, 0.01538, 0.01428, 0.00426], abs=3e-5)
    assert r["X"][0, -1, 7] == pytest.approx(0.99999, abs=2e-4)


@pytest.mark.parametrize("N,B", [(10, 6), (20, 6), (40, 4)])
def test_kkt_conditions_by_autograd(oracle, pkg, tables, N, B):
    """Algorithm-independent check: the returned primal-dual point satisfies the KKT conditions of the NLP as
    evaluated by a separate torch implementation (stationarity 1e-6, feasibility 1e-7, complementarity 1e-7)."""
    x0 = np.vstack([X0, pkg.sample_x0(tables, B - 1, seed=7)])
    r = oracle.solve(x0, N, nthreads=4)
    eps = oracle.o.smooth_eps_min  # smoothing length at the end of the solve
    for b in range(B):
        if r["status"][b] != 0:
            continue
        k = R.kkt_residuals(r, x0[b], np.zeros(2), tables, eps, b)
        assert k["stationarity"] < 1e-6 and k["equality"] < 1e-7, (b, k)
        assert k["ineq_violation"] < 1e-7 and k["complementarity"] < 1e-7 and k["min_multiplier"] >= 0.0, (b, k)
        assert k["objective"] == pytest.approx(r["obj"][b], rel=1e-10)
    assert (r["status"] == 0).sum() >= B - 1


def test_exact_tables_give_the_same_solution(orc, tables):
    """smooth_eps_min = smooth_scale = 0 is the reference's exact piece-wise-linear NLP; where it converges its
    solution agrees with the default (1e-4 m knot rounding) to ~1e-6."""
    o = orc.default_options()
    o.smooth_eps_min, o.smooth_scale = 0.0, 0.0
    exact = orc.Oracle(tables.packed(), options=o).solve(X0, 20)
    dflt = orc.Oracle(tables.packed()).solve(X0, 20)
    assert exact["status"][0] == 0 and dflt["status"][0] == 0
    assert np.abs(exact["u0"] - dflt["u0"]).max() < 1e-6
    assert np.abs(exact["X"] - dflt["X"]).max() < 1e-5


def test_right_constraint_split_is_equivalent(tables):
    """max(gR+, gR-) == the reference's -n + (L/2) sin(sign(mu) mu) + (W/2) cos(mu) - N_R on |mu| <= pi/2."""
    rng = np.random.default_rng(1)
    x = torch.tensor(np.column_stack([rng.uniform(0, 850, 500), rng.uniform(-3, 3, 500), rng.uniform(-np.pi / 2, np.pi / 2, 500)]
                                     + [np.zeros(500)] * 5))
    g = R.cons(x, tables, 0.0)
    ref = R.reference_right_constraint(x, tables)
    assert torch.max(torch.abs(torch.maximum(g[:, 1], g[:, 2]) - ref)) < 1e-14


def test_warm_start_and_uprev(oracle, pkg, tables):
    x0 = pkg.sample_x0(tables, 8, seed=11)
    r0 = oracle.solve(x0, 20, nthreads=4)
    x1 = oracle.plant_step(x0, r0["u0"])
    r1 = oracle.solve(x1, 20, uprev=r0["u0"], warm=r0, nthreads=4)
    cold = oracle.solve(x1, 20, uprev=r0["u0"], nthreads=4)
    ok = (r1["status"] == 0) & (cold["status"] == 0)
    assert ok.sum() >= 6
    # same KKT point from either start for most instances (the NLP is non-convex: a different start may reach
    # another local minimum; each one is checked against the KKT conditions in test_kkt_conditions_by_autograd)
    agree = np.abs(r1["u0"] - cold["u0"]).max(axis=1)[ok] < 1e-6
    assert agree.mean() >= 0.75
    assert r1["iters"][ok].mean() < cold["iters"][ok].mean()   # the un-shifted previous solution is a better start
    # rterm penalises u_0 - u_prev: a different u_prev changes the solution
    r2 = oracle.solve(x1, 20, uprev=r0["u0"] + 0.3, warm=r0, nthreads=4)
    assert np.abs(r2["u0"] - r1["u0"])[ok].max() > 1e-4
    for b in np.where(r1["status"] == 0)[0][:3]:
        k = R.kkt_residuals(r1, x1[b], r0["u0"][b], tables, oracle.o.smooth_eps_min, b)
        assert k["stationarity"] < 1e-6 and k["equality"] < 1e-7 and k["complementarity"] < 1e-7, k


def test_solver_failure_is_a_status_not_an_error(oracle):
    """Like the reference (IPOPT failure is silent): an infeasible start returns the last iterate + status."""
    x_bad = np.array([[100.0, 9.0, 0.0, 10.0, 0, 0, 0, 0]])  # 9 m off the race line: outside the track
    r = oracle.solve(x_bad, 10)
    assert r["status"][0] == 5 and np.all(np.isfinite(r["u0"]))   # INFEASIBLE: the restoration phase cannot remove the violation
    assert r["viol"][0] > 1.0 and r["iters"][0] < oracle.o.max_iter


def test_edge_horizons(oracle):
    r = oracle.solve(X0, 2)
    assert r["status"][0] == 0 and r["X"].shape == (1, 3, 8)
    r = oracle.solve(np.repeat(X0, 3, axis=0), 5, nthreads=3)
    assert np.abs(r["u0"] - r["u0"][0]).max() == 0.0  # identical instances -> identical bits, thread-count independent


def _midtrack_x0(tables, s):
    nl, nr = np.interp(s, tables.s_arc, tables.n_left), np.interp(s, tables.s_arc, tables.n_right)
    vref, kap = np.interp(s, tables.s_arc, tables.v_ref), np.interp(s, tables.s_kappa, tables.kappa)
    vx = 0.6 * vref
    return np.array([[s, 0.5 * (nl - nr), 0.0, vx, 0.0, kap * vx, np.arctan(3.0 * kap), 0.1]])


def test_soft_track_constraints_are_an_exact_penalty(orc, tables):
    """options.soft_rho (do_mpc: soft_constraint=True, penalty_term_cons): where the hard-constrained NLP is feasible
    and the penalty exceeds its multipliers, the softened NLP has the same solution."""
    hard = orc.Oracle(tables.packed())
    o = orc.default_options(); o.soft_rho = 100.0
    soft = orc.Oracle(tables.packed(), options=o)
    x0 = np.vstack([X0, _midtrack_x0(tables, 150.0), _midtrack_x0(tables, 600.0)])
    for N in (10, 20):
        a, b = hard.solve(x0, N, nthreads=3), soft.solve(x0, N, nthreads=3)
        assert np.all(a["status"] == 0) and np.all(b["status"] == 0)
        assert a["NU"][:, :, -3:].max() < 100.0  # the premise: multipliers of the hard problem below the penalty
        assert np.abs(a["u0"] - b["u0"]).max() < 1e-6
        assert np.abs(a["X"] - b["X"]).max() < 1e-5
        assert b["obj"] == pytest.approx(a["obj"], rel=1e-8, abs=1e-6)  # (the penalty adds rho e ~ mu_final per constraint)


# Closed-loop states (reference x0, hard constraints, oracle in the loop; scratch/oracle_lap.py) at which the
# hard-constrained solve stops with status STALLED, for N = 20 and N = 40: (x, u_prev)
STALL_STATES = {
    20: ([226.623754, -0.545036120, -0.0112268024, 8.52329373, 0.122918012, 0.161888169, 0.0837443810, 0.371730909],
         [0.78837973, -0.00572868]),
    40: ([271.551631, -3.15996996e-03, -0.138116402, 9.84560464, 0.483360380, 0.717172238, 0.338696158, -0.336634617],
         [1.17640462, -0.99999992]),
}


@pytest.mark.parametrize("N", [20, 40])
def test_soft_track_constraints_solve_where_the_hard_solve_stalls(orc, tables, N):
    """The states at which the closed loop with the reference's hard track constraints stopped converging in round 1
    (filter line search without a restoration phase: options.resto_rho = 0).  The softened NLP converges from them, to a
    point that satisfies its KKT conditions as evaluated by the independent torch implementation: for N = 20 without any
    violation (a KKT point of the hard NLP, multipliers 4 << rho: the stall was the solver's), for N = 40 with 0.1 mm of
    overlap at nu = rho."""
    x, up = (np.array([v]) for v in STALL_STATES[N])
    o0 = orc.default_options(); o0.resto_rho = 0.0
    assert orc.Oracle(tables.packed(), options=o0).solve(x, N, up)["status"][0] == 4
    rho = 100.0
    o = orc.default_options(); o.soft_rho = rho
    soft = orc.Oracle(tables.packed(), options=o)
    r = soft.solve(x, N, up)
    assert r["status"][0] == 0
    k = R.kkt_residuals(r, x[0], up[0], tables, soft.o.smooth_eps_min, 0, rho=rho)
    assert k["stationarity"] < 1e-6 and k["equality"] < 1e-7 and k["ineq_violation"] < 1e-7, k
    assert k["complementarity"] < 1e-6 and k["min_multiplier"] >= 0.0 and k["max_track_multiplier"] <= rho, k
    assert k["soft_violation"] < 1e-3, k
    assert (k["soft_violation"] == 0.0) == (N == 20)
    assert k["objective"] == pytest.approx(r["obj"][0], rel=1e-8, abs=1e-6)


def test_restoration_phase_solves_the_feasible_stall_state(orc, oracle, tables):
    """the N = 20 stall state is a FEASIBLE NLP on which the line search fails (the iterate jams
    against the steering-rate bound).  With the restoration phase (elastic mode, options.resto_rho = 1000, the default)
    the hard-constrained solve converges: status SOLVED on the hard problem's own KKT error, a KKT point by the
    independent torch evaluation, equal to what the softened NLP finds."""
    x, up = (np.array([v]) for v in STALL_STATES[20])
    r = oracle.solve(x, 20, up)
    assert r["status"][0] == 0 and r["n_resto"][0] == 1 and r["viol"][0] == 0.0 and r["kkt"][0] <= 1e-8
    k = R.kkt_residuals(r, x[0], up[0], tables, oracle.o.smooth_eps_min, 0)
    assert k["stationarity"] < 1e-6 and k["equality"] < 1e-7 and k["ineq_violation"] < 1e-7, k
    assert k["complementarity"] < 1e-7 and k["min_multiplier"] >= 0.0, k
    o = orc.default_options(); o.soft_rho = 100.0
    soft = orc.Oracle(tables.packed(), options=o).solve(x, 20, up)
    assert np.abs(r["u0"] - soft["u0"]).max() < 1e-6 and np.abs(r["X"] - soft["X"]).max() < 1e-5
    # ... also from a cold start and u_prev = 0 (the case traced in DESIGN.md §3)
    r0 = oracle.solve(x, 20)
    assert r0["status"][0] == 0 and r0["n_resto"][0] == 1 and r0["iters"][0] < 40


def test_restoration_phase_reports_local_infeasibility(orc, oracle, tables):
    """The N = 40 stall state: the horizon problem is (marginally) infeasible.  The elastic problem at resto_rho converges
    with 0.1 mm of overlap left, the penalty goes up to resto_rho_max (one escalation) and the overlap stays: status
    INFEASIBLE (IPOPT: 'converged to a point of local infeasibility'), and the iterate returned satisfies the KKT conditions
    of the elastic problem at that penalty (torch; stationarity in the units of the penalty scale S = rho / 1000)."""
    x, up = (np.array([v]) for v in STALL_STATES[40])
    r = oracle.solve(x, 40, up)
    assert r["status"][0] == 5 and r["status_solver"][0] == 5 and r["n_resto"][0] == 2 and 1e-5 < r["viol"][0] < 1e-3
    o = orc.default_options(); o.resto_rho_factor = 1.0   # round 2's rule: INFEASIBLE "at penalty resto_rho"
    r3 = orc.Oracle(tables.packed(), options=o).solve(x, 40, up)
    assert r3["status"][0] == 5 and r3["n_resto"][0] == 1 and r3["viol"][0] == pytest.approx(r["viol"][0], rel=0.1)
    rho, S = oracle.o.resto_rho_max, oracle.o.resto_rho_max / 1000.0
    k = R.kkt_residuals(r, x[0], up[0], tables, oracle.o.smooth_eps_min, 0, rho=rho)
    assert k["stationarity"] / S < 1e-5 and k["equality"] < 1e-7 and k["ineq_violation"] < 1e-7, k
    assert k["max_track_multiplier"] <= rho and k["soft_violation"] == pytest.approx(r["viol"][0], rel=1e-3), k


def _objective_free(orc, tables):
    """The feasibility problem behind status INFEASIBLE: no objective, the track constraints softened with an L1 penalty - its
    minimum is the least violation the horizon can reach from x0 (IPOPT's restoration problem without the proximity term)."""
    p = orc.default_params()
    for n in ("q_n", "q_mu", "q_vy", "q_v", "q_B"):
        setattr(p, n, 0.0)
    p.r_du[0] = p.r_du[1] = 0.0
    o = orc.default_options(); o.soft_rho, o.node0_check = 1e3, 0
    return orc.Oracle(tables.packed(), params=p, options=o)


def _shifted(w):
    s = {k: w[k].copy() for k in ("X", "C", "U", "L1", "L2")}
    for k in ("X", "C", "U"):
        s[k][:, :-1] = w[k][:, 1:]
    return s


def _certify_infeasible(orc, tables, x, up, warm, r, N, rng):
    """Multi-start certificate for a tick the solver called INFEASIBLE: the objective-free feasibility problem, started from the
    tick's warm start, its shifted copy, a cold start, the solver's own result and random perturbations, never gets below the
    violation the solver reported (the starts that converge all find that same least violation)."""
    feas = _objective_free(orc, tables)
    starts = [warm, _shifted(warm), None, r]
    for _ in range(3):
        s = {k: warm[k].copy() for k in ("X", "C", "U", "L1", "L2")}
        s["X"][:, 1:, 1:] += rng.normal(0, [0.2, 0.05, 0.5, 0.1, 0.1, 0.05, 0.1], s["X"][:, 1:, 1:].shape)
        s["C"] = 0.5 * (s["X"][:, :-1] + s["X"][:, 1:])
        starts.append(s)
    viols = []
    for s in starts:
        f = feas.solve(x, N, up, s, prev_status=None if s is None else np.array([4]))
        assert f["viol"][0] > 1e-8, "a feasible point exists: INFEASIBLE was a false negative"
        if f["status"][0] == 0:
            viols.append(f["viol"][0])
    assert len(viols) >= 4 and np.allclose(viols, r["viol"][0], rtol=2e-2), (viols, r["viol"][0])


def _reference_loop(orc, oracle, tables, N, ticks, certify):
    x, warm, st, up = np.array([[0, 0, 0, 5, 0, 0, 0, 0.1]], float), None, None, np.zeros((1, 2))
    hist, hist_solver, node0, rng = {}, {}, 0, np.random.default_rng(1)
    for tick in range(ticks):
        r = oracle.solve(x, N, up, warm, prev_status=st)
        s, ss = int(r["status"][0]), int(r["status_solver"][0])
        hist[s] = hist.get(s, 0) + 1; hist_solver[ss] = hist_solver.get(ss, 0) + 1
        assert ss in (0, 1, 5), (tick, ss)
        if ss == 5:   # the solver's own verdict: only at the largest penalty, and certified
            assert s == 5 and r["viol"][0] > 1e-8 and r["penalty"][0] == oracle.o.resto_rho_max, (tick, r["viol"][0], r["penalty"][0])
            if certify:
                _certify_infeasible(orc, tables, x, up, warm, r, N, rng)
        elif s == 5:  # the node-0 rule: the measured state is outside the band, the solve itself converged
            node0 += 1
            assert r["g0"][0] > oracle.o.acceptable_tol and r["viol"][0] == r["g0"][0], (tick, r["g0"][0])
        else:
            assert r["g0"][0] <= oracle.o.acceptable_tol and (s == 0) == (ss == 0 and r["g0"][0] <= oracle.o.tol), (tick, s, ss, r["g0"][0])
        warm, st, up = r, r["status"], r["u0"]
        x = oracle.plant_step(x, r["u0"], n_sub=100)
        if x[0, 0] > tables.s_arc[-1] - 5.0:
            break
    return hist, hist_solver, node0, x


def test_reference_loop_runs_with_hard_constraints(orc, oracle, tables):
    """The reference's loop (src/mpc.py:104-153: N = 10, x0 = [0,0,0,5,0,0,0,0.1], 500 ticks, hard track constraints):
    the solver itself ends every tick SOLVED / ACCEPTABLE except two (348, 349: a car 1e-5 m too wide for a corner), which are
    INFEASIBLE at the largest penalty and certified by the objective-free multi-start feasibility solves; the node-0 rule
    (the measured state is outside the band by more than acceptable_tol: the reference's NLP has no feasible point) flags
    further ticks INFEASIBLE.  Round 1 stopped converging at s = 227 m; round 2 had seven INFEASIBLE ticks "at penalty 1000"."""
    hist, hist_solver, node0, x = _reference_loop(orc, oracle, tables, 10, 500, certify=True)
    assert hist_solver.get(0, 0) + hist_solver.get(1, 0) >= 497 and 1 <= hist_solver.get(5, 0) <= 3, hist_solver
    assert hist.get(5, 0) == hist_solver.get(5, 0) + node0 and node0 <= 30 and x[0, 0] > 480.0, (hist, node0, x[0, 0])


def test_config_c5_lap_statuses_are_certified(orc, oracle, tables):
    """BASELINE config 5 (closed loop, N = 60, one lap) on the oracle: every INFEASIBLE verdict of the solver is certified."""
    hist, hist_solver, node0, x = _reference_loop(orc, oracle, tables, 60, 800, certify=True)
    assert hist_solver.get(0, 0) + hist_solver.get(1, 0) >= 700 and hist_solver.get(5, 0) <= 10, hist_solver


def test_round2_false_infeasibles_are_solved(orc, tables):
    """VERDICT r2 item 1.  Round 2's rule (elastic problem at penalty 1000 from the jam point, no escalation, no shifted
    restart) ended ticks 152 and 227 of the reference's loop INFEASIBLE although a feasible point exists (the
    elastic problem drifts into a local minimum of the violation; tick 227 is round 2's "tick 228").  With the defaults - the
    shifted restart first - both are SOLVED on the hard constraints, the restoration phase is not even entered."""
    o = orc.default_options(); o.resto_rho_factor, o.resto_shift_retry, o.node0_check, o.infeasible_sticky, o.dual_inf_max, o.max_mu_stay = 1.0, 0, 0, 0, 0.0, 0
    old, new = orc.Oracle(tables.packed(), options=o), orc.Oracle(tables.packed())
    x, warm, st, up = np.array([[0, 0, 0, 5, 0, 0, 0, 0.1]], float), None, None, np.zeros((1, 2))
    seen = {}
    for tick in range(228):
        r = old.solve(x, 10, up, warm, prev_status=st)
        if r["status"][0] == 5:
            seen[tick] = (r, new.solve(x, 10, up, warm, prev_status=st))
        warm, st, up = r, r["status"], r["u0"]
        x = old.plant_step(x, r["u0"], n_sub=100)
    assert set(seen) == {152, 153, 227}, sorted(seen)
    for tick in (152, 227):
        r_old, r_new = seen[tick]
        assert r_old["viol"][0] > 1e-3
        assert r_new["status"][0] == 0 and r_new["n_shift"][0] == 1 and r_new["n_resto"][0] == 0 and r_new["viol"][0] == 0.0, (tick, r_new["status"], r_new["n_resto"])
        assert r_new["obj"][0] < r_old["obj"][0]
    # tick 153 starts from the state tick 152's least-violation control led to: outside the band already (node 0)
    r_old, r_new = seen[153]
    assert r_new["g0"][0] > 1e-3 and r_new["status"][0] == 5


def test_node0_rule(orc, oracle, tables):
    """options.node0_check (do_mpc checks the track constraints at node 0 too, controller.py:69-70): the solve is the same, the
    status says what the reference's NLP is."""
    x = _midtrack_x0(tables, 100.0).copy()
    g = oracle.cons_derivs(x[0], eps=oracle.o.smooth_eps_min)[0]
    for shift, want in ((-g[0] + 5e-7, 1), (-g[0] + 1e-3, 5), (-g[0] - 1e-3, 0)):   # n moved so that gL(x0) = 5e-7 / 1e-3 / -1e-3
        xs = x.copy(); xs[0, 1] += shift
        r = oracle.solve(xs, 10)
        assert r["status_solver"][0] == 0 and r["status"][0] == want, (shift, r["status"], r["g0"])
        assert r["g0"][0] == pytest.approx(g[0] + shift, abs=1e-9)
        o = orc.default_options(); o.node0_check = 0
        r0 = orc.Oracle(tables.packed(), options=o).solve(xs, 10)
        assert r0["status"][0] == 0 and np.array_equal(r0["u0"], r["u0"])


def test_second_order_correction_in_the_oracle(orc, pkg, tables):
    """options.max_soc (IPOPT's second-order correction; oracle only): corrected steps are taken (n_soc > 0), the solves
    end at the same KKT points, and - the reason the kernels do not have it - the iteration counts do not improve."""
    x0 = pkg.sample_x0(tables, 96, seed=4)
    o = orc.default_options(); o.max_soc = 4
    a = orc.Oracle(tables.packed()).solve(x0, 40, nthreads=8)
    b = orc.Oracle(tables.packed(), options=o).solve(x0, 40, nthreads=8)
    assert a["n_soc"].sum() == 0 and b["n_soc"].sum() >= 10
    ok = (a["status"] == 0) & (b["status"] == 0)
    assert ok.mean() > 0.95
    same = np.abs(a["u0"] - b["u0"]).max(axis=1)[ok] < 1e-6
    assert same.mean() > 0.9
    assert b["iters"][ok].sum() > 0.9 * a["iters"][ok].sum()


def test_periodic_tables_option(orc, tables):
    """options.periodic_tables (closed track): tables are evaluated at s modulo their span, so a
    state one lap further on gives the same plant step and the same control; off (the reference: linear extrapolation
    beyond the last knot) the two differ.  On the first lap the option changes nothing."""
    L = tables.s_arc[-1] - tables.s_arc[0]
    x = np.vstack([_midtrack_x0(tables, s) for s in (30.0, 400.0, 845.0)])
    xl = x.copy(); xl[:, 0] += L
    o = orc.default_options(); o.periodic_tables = 1
    per, ref = orc.Oracle(tables.packed(), options=o), orc.Oracle(tables.packed())
    u = np.tile([[0.2, 0.5]], (3, 1))
    a, b = per.plant_step(x, u), per.plant_step(xl, u)
    assert np.abs(b[:, 0] - L - a[:, 0]).max() < 1e-9 and np.abs(b[:, 1:] - a[:, 1:]).max() < 1e-10
    assert np.array_equal(per.plant_step(x[:2], u[:2]), ref.plant_step(x[:2], u[:2]))  # first lap, away from the seam: identical
    assert np.abs(ref.plant_step(xl, u)[:, 1:] - a[:, 1:]).max() > 1e-4        # extrapolated tables are a different track
    N = 20
    ra, rb = per.solve(x, N, nthreads=3), per.solve(xl, N, nthreads=3)
    assert np.all(ra["status"] == 0) and np.all(rb["status"] == 0)
    assert np.abs(ra["u0"] - rb["u0"]).max() < 1e-7 and np.abs(ra["X"][:, :, 1:] - rb["X"][:, :, 1:]).max() < 1e-6
    assert np.abs(rb["X"][:, :, 0] - L - ra["X"][:, :, 0]).max() < 1e-6
    # the horizon of the third state crosses the seam (845 m + 20 x 0.1 s x ~9 m/s > 857.9 m)
    assert ra["X"][2, -1, 0] > tables.s_arc[-1]
    r0 = ref.solve(x[:2], N, nthreads=2)
    assert np.abs(r0["u0"] - ra["u0"][:2]).max() < 1e-12
