"""TEST INFRASTRUCTURE - an independent exact solver for the QP of the iterative-linear agent.

Only tests/ and tests/golden/make_ltv_independent.py import this; the product never does.

The reference hands the QP of `_linear_mpc_control` (agents/pure_mpc_linear.py:205-257) to cvxpy -> ECOS, which do not
exist offline.  The kernel (csrc/mpc_ltv.hpp) and oracle/ltv_oracle.py both run the same Mehrotra interior-point
iteration, so their agreement says nothing about the QP.  This file is the third party: the dual active-set method of
Goldfarb and Idnani (Math. Programming 27, 1983) for strictly convex QPs,

    minimise 1/2 x'Hx + g'x   subject to   A x >= b,

which shares neither iterates nor linear algebra nor stopping rule with an interior-point method: it starts at the
unconstrained minimiser, adds the most violated constraint, walks along the dual-feasible path (dropping constraints
whose multipliers reach zero) and ends, after finitely many steps, at the exact minimiser with the exact active set -
accurate to the rounding of a few dense solves, not to a barrier parameter.  `build_from_loops` assembles (H, g, A, b)
from the plain-loop transcription of the cvxpy statements (`ltv_oracle.objective_loops` / `constraint_loops`) by
evaluating them at unit vectors - the QP is quadratic / affine, so this is exact - and so shares no matrix code with
`ltv_oracle.build_qp` either.
"""
from __future__ import annotations

import numpy as np


class Infeasible(Exception):
    pass


def solve(H, g, A, b, tol=1e-11, max_steps=2000):
    """Goldfarb-Idnani.  H [n,n] symmetric positive definite, A [m,n], b [m].  Returns x, multipliers u [m] (>= 0, zero
    off the active set), active set (sorted list)."""
    H = np.asarray(H, dtype=np.float64)
    n = H.shape[0]
    L = np.linalg.cholesky(H)
    Hinv = lambda v: np.linalg.solve(L.T, np.linalg.solve(L, v))
    x = -Hinv(g)
    act: list[int] = []
    u = np.zeros(0)
    scale = 1.0 + np.abs(b)
    for _ in range(max_steps):
        s = A @ x - b
        viol = s / scale
        viol[act] = 0.0
        p = int(np.argmin(viol))
        if viol[p] >= -tol:
            mult = np.zeros(A.shape[0])
            mult[act] = u
            return x, mult, sorted(act)
        npv = A[p]
        up = 0.0
        while True:
            if act:
                N = A[act].T                               # [n, q]
                HiN = Hinv(N)
                G = N.T @ HiN
                r = np.linalg.solve(G, HiN.T @ npv)        # change of the active multipliers per unit of u_p
                z = Hinv(npv) - HiN @ r                    # primal direction
            else:
                r = np.zeros(0)
                z = Hinv(npv)
            zn = float(z @ npv)
            # largest dual step keeping the active multipliers non-negative
            t1, drop = np.inf, -1
            for j in range(len(act)):
                if r[j] > 1e-14 and u[j] / r[j] < t1:
                    t1, drop = u[j] / r[j], j
            sp = float(npv @ x - b[p])
            t2 = -sp / zn if zn > 1e-13 * (1.0 + float(npv @ npv)) else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                raise Infeasible("constraints are inconsistent")
            if np.isfinite(t2):
                x = x + t * z
            u = u - t * r
            up += t
            if t == t2:                                    # full step: constraint p becomes active
                act.append(p)
                u = np.append(u, up)
                break
            del act[drop]                                  # partial step: drop the blocking constraint, try again
            u = np.delete(u, drop)
    raise RuntimeError("active-set iteration limit")


def build_from_loops(L, x0, xref, xbar, dt, T):
    """(H, g, A, b, f0) of one instance from `L.objective_loops` / `L.constraint_loops` (L = ltv_oracle) evaluated at 0,
    the unit vectors and their pairwise sums.  Rows of A whose normal is zero (the speed bounds of node 0, constants)
    are checked for feasibility and left out."""
    n = 2 * T
    f = lambda v: L.objective_loops(v.reshape(T, 2), x0, xref, xbar, dt)
    c = lambda v: L.constraint_loops(v.reshape(T, 2), x0, xbar, dt)
    zero = np.zeros(n)
    f0, c0 = f(zero), c(zero)
    E = np.eye(n)
    fe = np.array([f(E[i]) for i in range(n)])
    fm = np.array([f(-E[i]) for i in range(n)])
    g = 0.5 * (fe - fm)
    H = np.zeros((n, n))
    for i in range(n):
        H[i, i] = fe[i] + fm[i] - 2.0 * f0
        for j in range(i + 1, n):
            H[i, j] = H[j, i] = f(E[i] + E[j]) - fe[i] - fe[j] + f0
    A = np.stack([c(E[i]) - c0 for i in range(n)], axis=1)     # c(v) = c0 + A v >= 0
    const = np.abs(A).max(axis=1) == 0.0
    if (c0[const] < 0.0).any():
        raise Infeasible("a constant constraint row is violated")
    return H, g, A[~const], -c0[~const], f0


def certify(H, g, C, c0, u, eps_c=1e-5):
    """KKT certificate of a claimed minimiser u of  1/2 u'Hu + g'u  s.t.  c0 + C u >= 0, without the solver's
    multipliers: the rows with c <= eps_c (1 + |c0|) count as active and non-negative multipliers on them are fitted
    to the gradient by NNLS.  Returns (stationarity residual |grad - C_A' z|_inf / max(1, |grad|_inf), worst constraint
    violation, number of active rows).  The QP is strictly convex: a feasible point with zero residual IS the
    minimiser, and a residual r bounds the distance to it by r |grad| / lambda_min(H)."""
    from scipy.optimize import nnls
    c = c0 + C @ u
    grad = H @ u + g
    act = c <= eps_c * (1.0 + np.abs(c0))
    if act.any():
        z, _ = nnls(C[act].T, grad, maxiter=50 * C.shape[1])
        res = grad - C[act].T @ z
    else:
        res = grad
    return float(np.abs(res).max() / max(1.0, np.abs(grad).max())), float(max(0.0, -c.min())), int(act.sum())
