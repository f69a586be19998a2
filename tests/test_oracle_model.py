"""Oracle pinning (1): model functions against the reference's only recorded artefact
(simulation_recorded_results.json) and derivative self-consistency."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import GOLDEN

REC = json.load(open(os.path.join(GOLDEN, "simulation_recorded_results.json")))
X = np.array(REC["x"])[:, :, 0]
U = np.array(REC["u"])[:, :, 0]


def test_recorded_slip_angles_and_forces(oracle):
    """26 rows of alpha / Fy recomputed from x: pins model.py:101-114 incl. D_f = D_r = 1 and the minus sign."""
    a, F = oracle.slip_forces(X)
    assert np.abs(a[1:] - np.array(REC["alpha"])[1:]).max() <= 1e-15
    assert np.abs(F[1:] - np.array(REC["Fy"])[1:]).max() <= 1e-10
    assert F[1] == pytest.approx([-185.23375516094129, -79.17057174452631], abs=1e-10)


def test_recorded_plant_transitions(oracle):
    """x[i] --u[i+1], 0.1 s--> x[i+1] (25 transitions): pins the rhs (model.py:152-183), the kappa table and the
    plant integrator.  Tolerances as found in SURVEY.md §4 (the s/n/mu residue is the scipy-version LUT difference)."""
    xn = oracle.plant_step(X[:-1], U[1:], n_sub=400)
    err = np.abs(xn - X[1:]).max(axis=0)
    assert err[3] < 1e-9 and err[4] < 1e-9 and err[5] < 5e-9, err   # vx, vy, r
    assert err[0] < 2e-6 and err[1] < 5e-6 and err[2] < 1e-5, err   # s, n, mu
    assert err[6] < 1e-12 and err[7] < 1e-12                        # linear states


def test_plant_substep_convergence(oracle):
    a = oracle.plant_step(X[:5], U[1:6], n_sub=400)
    b = oracle.plant_step(X[:5], U[1:6], n_sub=1600)
    assert np.abs(a - b).max() < 1e-9


state = st.tuples(st.floats(5, 700), st.floats(-0.5, 0.5), st.floats(-0.3, 0.3), st.floats(3, 25), st.floats(-1, 1),
                  st.floats(-0.5, 0.5), st.floats(-0.4, 0.4), st.floats(-1, 1))


@settings(max_examples=60, deadline=None)
@given(state, st.sampled_from([0.0, 1e-3]))
def test_jacobian_matches_finite_differences(oracle, x, eps):
    x = np.array(x)
    lam = np.linspace(-1, 1, 8)
    f, fx, H = oracle.rhs_derivs(x, lam, eps)
    h = 1e-6
    for j in range(1, 8):  # s handled separately: the table is only piece-wise smooth
        e = np.zeros(8); e[j] = h
        fd = (oracle.rhs(x + e, np.zeros(2)) - oracle.rhs(x - e, np.zeros(2))) / (2 * h)
        assert np.abs(fd[:6] - fx[:6, j]).max() < 2e-5 * (1 + np.abs(fx[:, j]).max())
        Hp = oracle.rhs_derivs(x + e, lam, eps)[1]
        Hm = oracle.rhs_derivs(x - e, lam, eps)[1]
        col = lam[:6] @ (Hp[:6] - Hm[:6]) / (2 * h)
        assert np.abs(col[1:] - H[j, 1:]).max() < 2e-4 * (1 + np.abs(H).max())
    assert np.abs(H - H.T).max() == 0.0


def test_sin_abs_mu_convention(oracle):
    """d/dmu sin(sign(mu) mu) = sign(mu) cos(mu) with sign(0) = 0; second derivative -sin|mu| (App. A item 5)."""
    for mu, sg in ((0.2, 1.0), (-0.2, -1.0), (0.0, 0.0)):
        x = np.array([50.0, 0.1, mu, 10, 0, 0, 0, 0])
        v, g, H = oracle.cons_derivs(x)
        assert g[0, 2] == pytest.approx(-1.5 * sg * np.cos(mu) - 1.15 * np.sin(mu), abs=1e-14)
        assert H[0, 2, 2] == pytest.approx(1.5 * sg * sg * np.sin(abs(mu)) - 1.15 * np.cos(mu), abs=1e-14)
        # right constraint split: max(gR+, gR-) is the reference's expression with sin|mu|
        assert max(v[1], v[2]) == pytest.approx(-0.1 + 1.5 * np.sin(abs(mu)) + 1.15 * np.cos(mu) - (-(v[1] + 0.1 - 1.5 * np.sin(mu) - 1.15 * np.cos(mu))), abs=1e-13)


def test_torque_vectoring_term(orc, tables):
    """params.ptv (model.py:162-164, `Mtv = ptv * (rt - r)`, commented out in the reference = ptv 0): the r equation gains
    ptv (tan(delta) vx / (l_f + l_r) - r) / I_z, nothing else changes; jets against finite differences with ptv = 0.5
    (the value in MX5.json)."""
    p = orc.default_params(); p.ptv = 0.5
    a, b = orc.Oracle(tables.packed()), orc.Oracle(tables.packed(), params=p)
    rng = np.random.default_rng(3)
    for _ in range(20):
        x = np.array([rng.uniform(5, 700), rng.uniform(-.5, .5), rng.uniform(-.3, .3), rng.uniform(3, 25), rng.uniform(-1, 1),
                      rng.uniform(-.5, .5), rng.uniform(-.4, .4), rng.uniform(-1, 1)])
        fa, fb = a.rhs(x, np.zeros(2)), b.rhs(x, np.zeros(2))
        assert np.array_equal(np.delete(fa, 5), np.delete(fb, 5))
        assert fb[5] - fa[5] == pytest.approx(0.5 * (np.tan(x[6]) * x[3] / 3.0 - x[5]) / 1000.0, abs=1e-13)
        lam = np.linspace(-1, 1, 8)
        f, fx, H = b.rhs_derivs(x, lam)
        h = 1e-6
        for j in range(1, 8):
            e = np.zeros(8); e[j] = h
            fd = (b.rhs(x + e, np.zeros(2)) - b.rhs(x - e, np.zeros(2))) / (2 * h)
            assert np.abs(fd[:6] - fx[:6, j]).max() < 2e-5 * (1 + np.abs(fx[:, j]).max())
            col = lam[:6] @ (b.rhs_derivs(x + e, lam)[1][:6] - b.rhs_derivs(x - e, lam)[1][:6]) / (2 * h)
            assert np.abs(col[1:] - H[j, 1:]).max() < 2e-4 * (1 + np.abs(H).max())
    # it changes the solutions
    import importlib
    x0 = importlib.import_module("lap-time-optimization_amd").sample_x0(tables, 8, seed=19)
    ra, rb = a.solve(x0, 20, nthreads=4), b.solve(x0, 20, nthreads=4)
    ok = (ra["status"] == 0) & (rb["status"] == 0)
    assert ok.sum() >= 6 and np.abs(ra["X"] - rb["X"])[ok].max() > 1e-4   # (another predicted trajectory)
