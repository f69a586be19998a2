"""Independent Python/torch restatement of the reference NLP (functions only) used to CHECK solutions.

Third implementation besides the C oracle (forward-mode jets) and the HIP kernels (hand-derived derivatives):
plain torch float64 expressions + reverse-mode autograd.  Follows
  src/mpc/model.py:101-117,124-128,152-183 (dynamics, tyres), :70-84 (constraints),
  src/mpc/controller.py:36-55,57-103 (objective, bounds), do_mpc Radau-IIA(2) collocation.
It evaluates KKT residuals at a given primal-dual point; it contains no solver.
"""
from __future__ import annotations

import math

import numpy as np
import torch

torch.set_default_dtype(torch.float64)

P = dict(m=1000.0, Iz=1000.0, lf=1.5, lr=1.5, W=2.3, Bf=10.0, Cf=1.3, Df=1.0, Br=12.0, Cr=1.2, Dr=1.0, Cm=1000.0,
         Cr0=0.01, Cr2=0.0003, g=9.81, q_n=0.5, q_mu=3.0, q_B=1e-2, r=(1e-2, 1e-2))
# (state index, sign, value) in the solver's order: per state lower then upper (controller.py:79-95)
XB = [(0, -1, 0.0), (2, -1, -math.pi / 2), (2, 1, math.pi / 2), (3, -1, 0.0), (6, -1, -math.pi / 4), (6, 1, math.pi / 4),
      (7, -1, -1.0), (7, 1, 1.0)]
UB = [(0, -1, -math.pi / 2), (0, 1, math.pi / 2), (1, -1, -1.0), (1, 1, 1.0)]


def lut(grid: np.ndarray, y: np.ndarray, s: torch.Tensor, eps: float) -> torch.Tensor:
    """Piece-wise-linear table with linear extrapolation (CasADi `interpolant('linear')`), optionally with the
    compact pseudo-Huber rounding of the knots used by the solver."""
    g, yy = torch.as_tensor(grid), torch.as_tensor(y)
    n = g.numel()
    i = torch.clamp(torch.searchsorted(g, s.detach().contiguous(), right=True) - 1, 0, n - 2)
    sl = (yy[i + 1] - yy[i]) / (g[i + 1] - g[i])
    val = yy[i] + sl * (s - g[i])
    if eps > 0:
        slopes = (yy[1:] - yy[:-1]) / (g[1:] - g[:-1])
        d = g[i + 1] - g[i]
        for side in (0, 1):  # knot at the left / right end of the interval
            kn = i + side
            ok = (kn > 0) & (kn < n - 1)
            knc = torch.clamp(kn, 1, n - 2)
            Wk = 0.5 * torch.minimum(g[knc] - g[knc - 1], g[knc + 1] - g[knc])
            z = s - g[knc]
            inside = ok & ((z >= 0) & (z < Wk) if side == 0 else (z < 0) & (-z < Wk))
            J = slopes[knc] - slopes[knc - 1]
            RW = torch.sqrt(Wk * Wk + eps * eps)
            a = (1.0 - Wk / RW) / (2.0 * Wk)
            b = Wk - RW - a * Wk * Wk
            sg = 1.0 if side == 0 else -1.0
            corr = 0.5 * J * (torch.sqrt(z * z + eps * eps) + a * z * z + b - sg * z)
            val = val + torch.where(inside, corr, torch.zeros_like(corr))
        del d
    return val


def rhs(x: torch.Tensor, u: torch.Tensor, tab, eps: float) -> torch.Tensor:
    s, n, mu, vx, vy, r, de, th = x.unbind(-1)
    kap = lut(tab.s_kappa, tab.kappa, s, eps)
    sdot = (vx * torch.cos(mu) - vy * torch.sin(mu)) / (1 - n * kap)
    af = torch.atan2(vy + P["lf"] * r, vx) - de
    ar = torch.atan2(vy - P["lr"] * r, vx)
    L = P["lf"] + P["lr"]
    Fnf, Fnr = P["lr"] * P["m"] * P["g"] / L, P["lf"] * P["m"] * P["g"] / L
    Fyf = -Fnf * P["Df"] * torch.sin(P["Cf"] * torch.atan(P["Bf"] * af))
    Fyr = -Fnr * P["Dr"] * torch.sin(P["Cr"] * torch.atan(P["Br"] * ar))
    Fx = P["Cm"] * th - P["Cr0"] - P["Cr2"] * vx * vx
    return torch.stack([
        sdot, vx * torch.sin(mu) + vy * torch.cos(mu), r - kap * sdot,
        (Fx - Fyf * torch.sin(de) + P["m"] * vy * r) / P["m"],
        (Fyr + Fyf * torch.cos(de) - P["m"] * vx * r) / P["m"],
        (Fyf * P["lf"] * torch.cos(de) - Fyr * P["lr"]) / P["Iz"],
        u[..., 0], u[..., 1]], dim=-1)


def mterm(x):
    return P["q_n"] * x[..., 1] ** 2 + P["q_mu"] * x[..., 2] ** 2 + x[..., 4] ** 2


def lterm(x, tab, eps):
    vref = lut(tab.s_arc, tab.v_ref, x[..., 0], eps)
    bdyn = torch.atan(x[..., 4] / x[..., 3])
    bkin = torch.atan(x[..., 6] * P["lr"] / (P["lf"] + P["lr"]))
    return mterm(x) + (x[..., 3] - 0.6 * vref) ** 2 + P["q_B"] * (bdyn - bkin) ** 2


def cons(x, tab, eps):
    """gL, gR+, gR- (the reference's right constraint with sin|mu| == max(gR+, gR-) on |mu| <= pi/2)."""
    s, n, mu = x[..., 0], x[..., 1], x[..., 2]
    NL, NR = lut(tab.s_arc, tab.n_left, s, eps), lut(tab.s_arc, tab.n_right, s, eps)
    hl, hw = 0.5 * (P["lf"] + P["lr"]), 0.5 * P["W"]
    sabs = torch.sin(torch.sign(mu.detach()) * mu)  # sign' = 0 (CasADi), SURVEY App. A item 5
    gl = n - hl * sabs + hw * torch.cos(mu) - NL
    grp = -n + hl * torch.sin(mu) + hw * torch.cos(mu) - NR
    grm = -n - hl * torch.sin(mu) + hw * torch.cos(mu) - NR
    return torch.stack([gl, grp, grm], dim=-1)


def reference_right_constraint(x, tab):
    """The constraint exactly as written in model.py:78 (for the equivalence test of the split)."""
    s, n, mu = x[..., 0], x[..., 1], x[..., 2]
    NR = lut(tab.s_arc, tab.n_right, s, 0.0)
    return -n + 0.5 * (P["lf"] + P["lr"]) * torch.sin(torch.sign(mu) * mu) + 0.5 * P["W"] * torch.cos(mu) - NR


def objective(X, U, uprev, tab, eps):
    N = U.shape[0]
    J = lterm(X[:N], tab, eps).sum() + mterm(X[N])
    Uprev = torch.cat([torch.as_tensor(uprev).reshape(1, 2), U[:-1]], dim=0)
    dU = U - Uprev
    return J + (torch.as_tensor(P["r"]) * dU * dU).sum()


def collocation(X, C, U, tab, eps, h=0.1):
    xk, xp = X[:-1], X[1:]
    G1 = h * rhs(C, U, tab, eps) + 2 * xk - 1.5 * C - 0.5 * xp
    G2 = h * rhs(xp, U, tab, eps) - 2 * xk + 4.5 * C - 2.5 * xp
    return G1, G2


def ellipse(x, ell):
    """Friction-ellipse constraints (model.py:86-99), normalised by the radius: ell = (penalty, rho, D_f, D_r)."""
    vx, vy, r, de, th = x[..., 3], x[..., 4], x[..., 5], x[..., 6], x[..., 7]
    af = torch.atan2(vy + P["lf"] * r, vx) - de
    ar = torch.atan2(vy - P["lr"] * r, vx)
    L = P["lf"] + P["lr"]
    Fnf, Fnr = P["lr"] * P["m"] * P["g"] / L, P["lf"] * P["m"] * P["g"] / L
    Fyf = -Fnf * P["Df"] * torch.sin(P["Cf"] * torch.atan(P["Bf"] * af))
    Fyr = -Fnr * P["Dr"] * torch.sin(P["Cr"] * torch.atan(P["Br"] * ar))
    lng = ell[1] * 0.5 * P["Cm"] * th
    return torch.stack([(lng * lng + Fyf * Fyf) / ell[2] ** 2 - 1.0, (lng * lng + Fyr * Fyr) / ell[3] ** 2 - 1.0], dim=-1)


def inequalities(X, C, U, tab, eps, ell=None):
    """(N, 23 [+ 2]): u bounds, c bounds, x+ bounds, gL/gR+/gR- [, ellipse front / rear] at x+; h <= 0.  Last row's nl entries are
    not constraints."""
    cols = []
    for idx, sg, val in UB:
        cols.append(sg * (U[:, idx] - val))
    for idx, sg, val in XB:
        cols.append(sg * (C[:, idx] - val))
    for idx, sg, val in XB:
        cols.append(sg * (X[1:, idx] - val))
    g = cons(X[1:], tab, eps)
    if ell is not None:
        g = torch.cat([g, ellipse(X[1:], ell)], dim=-1)
    return torch.cat([torch.stack(cols, dim=-1), g], dim=-1)


def kkt_residuals(sol: dict, x0, uprev, tab, eps: float, b: int = 0, h: float = 0.1, rho: float = 0.0, ell=None):
    """Residuals of the KKT conditions of the NLP at solution `sol` (dict with X, C, U, L1, L2, NU of instance b):
    returns dict(stationarity, equality, ineq_violation, complementarity, min_multiplier, objective).

    ell = (penalty, rho_long, D_f, D_r): the two friction-ellipse constraints are present (always soft, penalty ell[0]); returned
    in addition: ell_violation, ell_complementarity, max_ell_multiplier; the objective includes their penalty.
    rho > 0: the track constraints are softened with an exact L1 penalty, min J + rho sum(e), g - e <= 0, e >= 0.  With
    the elastic variables at their optimum e = max(g, 0) the conditions are: stationarity in (x, c, u) as before,
    0 <= nu <= rho, nu * max(-g, 0) = 0 and (rho - nu) * max(g, 0) = 0; `ineq_violation` then covers the bounds only,
    the track violation is returned as `soft_violation` and the objective includes the penalty."""
    X = torch.tensor(sol["X"][b]).clone()
    X[0] = torch.as_tensor(x0)
    Xf = X[1:].clone().requires_grad_(True)
    C = torch.tensor(sol["C"][b]).clone().requires_grad_(True)
    U = torch.tensor(sol["U"][b]).clone().requires_grad_(True)
    L1, L2, NU = torch.tensor(sol["L1"][b]), torch.tensor(sol["L2"][b]), torch.tensor(sol["NU"][b]).clone()
    N = U.shape[0]
    NU[N - 1, -3:] = 0.0  # no nl constraint at node N
    Xall = torch.cat([X[:1], Xf], dim=0)
    J = objective(Xall, U, uprev, tab, eps)
    G1, G2 = collocation(Xall, C, U, tab, eps, h)
    nnl = 3 if ell is None else 5
    NU[N - 1, -nnl:] = 0.0
    H = inequalities(Xall, C, U, tab, eps, ell)
    Lag = J + (L1 * G1).sum() + (L2 * G2).sum() + (NU * H).sum()
    gX, gC, gU = torch.autograd.grad(Lag, [Xf, C, U])
    Hd = H.detach().clone()
    Hd[N - 1, -nnl:] = -1.0
    extra = {}
    if ell is not None:   # split the (always soft) ellipse columns off: the rest is checked as before
        He, NUe = Hd[:, -2:], NU[:, -2:]
        Hd, NU = Hd[:, :-2], NU[:, :-2]
        compe = torch.maximum((NUe * torch.clamp(-He, min=0.0)).abs().max(), ((ell[0] - NUe) * torch.clamp(He, min=0.0)).abs().max())
        extra = dict(ell_violation=float(torch.clamp(He, min=0.0).max()), ell_complementarity=float(compe), max_ell_multiplier=float(NUe.max()),
                     min_ell_multiplier=float(NUe.min()))
        J = J + ell[0] * torch.clamp(He, min=0.0).sum()
    if rho > 0:
        Ht, NUt = Hd[:, -3:], NU[:, -3:]
        comp = torch.maximum((NUt * torch.clamp(-Ht, min=0.0)).abs().max(), ((rho - NUt) * torch.clamp(Ht, min=0.0)).abs().max())
        comp = torch.maximum(comp, (NU[:, :-3] * Hd[:, :-3]).abs().max())
        return dict(stationarity=float(max(gX.abs().max(), gC.abs().max(), gU.abs().max())),
                    equality=float(max(G1.detach().abs().max(), G2.detach().abs().max())),
                    ineq_violation=float(Hd[:, :-3].max()), soft_violation=float(torch.clamp(Ht, min=0.0).max()),
                    complementarity=float(comp), min_multiplier=float(NU.min()), max_track_multiplier=float(NUt.max()),
                    objective=float(J.detach() + rho * torch.clamp(Ht, min=0.0).sum()), **extra)
    return dict(stationarity=float(max(gX.abs().max(), gC.abs().max(), gU.abs().max())),
                equality=float(max(G1.detach().abs().max(), G2.detach().abs().max())),
                ineq_violation=float(Hd.max()), complementarity=float((NU * Hd).abs().max()),
                min_multiplier=float(NU.min()), objective=float(J.detach()), **extra)
