"""TEST INFRASTRUCTURE - independent-solver cross-check of the CPU oracle.

Solves the same NLP (oracle/nlp_spec.py) with scipy's SLSQP (a different algorithm family:
active-set SQP with dense BFGS) from the oracle's cold-start rollout and reports the agreement
of the returned action u0 and of the full control sequence.  Used by tests/ and by
tests/golden/make_golden.py; never by the product path.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import minimize

import nlp_spec as S


def solve_slsqp(p: S.Problem, U_init=None, maxiter=400, ftol=1e-14):
    N = p.N

    def rollout(U):
        X = np.zeros((N + 1, 4))
        X[0] = p.state
        for k in range(N):
            X[k + 1] = X[k] + S.f_dyn(X[k], U[k]) * p.dt
        return X

    def fun(zu):
        U = zu.reshape(N, 2)
        X = rollout(U)
        gX, gU = S.cost_grad(p, X, U)
        # adjoint sweep for the reduced gradient
        lam = np.zeros(4)
        g = np.zeros_like(U)
        for k in range(N - 1, -1, -1):
            A, B = S.f_jac(X[k], U[k])
            g[k] = gU[k] + p.dt * B.T @ lam
            lam = gX[k] + lam + p.dt * A.T @ lam
        return S.cost(p, X, U), g.ravel()

    def state_ineq(zu):     # theta, v bounds on X[1..N]
        X = rollout(zu.reshape(N, 2))[1:]
        return np.concatenate([X[:, 2] - S.X_LO[2], S.X_HI[2] - X[:, 2], X[:, 3] - S.X_LO[3], S.X_HI[3] - X[:, 3]])

    U0 = np.zeros((N, 2)) if U_init is None else np.array(U_init, dtype=float)
    bnds = [(S.U_LO[i % 2], S.U_HI[i % 2]) for i in range(2 * N)]
    res = minimize(fun, U0.ravel(), jac=True, method="SLSQP", bounds=bnds,
                   constraints=[dict(type="ineq", fun=state_ineq)],
                   options=dict(maxiter=maxiter, ftol=ftol))
    U = res.x.reshape(N, 2)
    return dict(U=U, X=rollout(U), success=res.success, nit=res.nit, fun=res.fun)
