"""The CPU oracle of the iterative-linear agent (oracle/ltv_oracle.py) against what pins it: the reference's own numpy
helpers (golden vectors made by importing agents/pure_mpc_linear.py, tests/golden/make_golden.py), an independent
evaluation of the QP (plain loops over the cvxpy statements) driving scipy SLSQP, and KKT certificates."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ltv_states


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "ltv_reference_numpy.npz"))


def test_constants_match_reference(gold, ltv_oracle):
    L = ltv_oracle
    want = gold["constants"]
    got = [L.MAX_STEER, L.MAX_DSTEER, L.MAX_ACCEL, L.MAX_DECEL, L.MAX_SPEED, L.R_DIAG[0], L.R_DIAG[1], L.RD_DIAG[0],
           L.RD_DIAG[1], L.Q_V, L.Q_YAW, *L.QF_DIAG]
    assert np.array_equal(np.array(got), want)


def test_helpers_match_reference(gold, ltv_oracle, ref_table):
    L = ltv_oracle
    pos = gold["nearest_in"]
    assert np.array_equal(L.nearest_index(pos[:, 0], pos[:, 1], ref_table), gold["nearest_out"])
    A, Bm = L.linear_model(gold["linmodel_in"][:, 0], gold["linmodel_in"][:, 1], 0.1)
    assert np.array_equal(A, gold["linmodel_A"]) and np.array_equal(Bm, gold["linmodel_B"])
    xbar = L.nominal_rollout(gold["nominal_x0"], gold["nominal_oa"], gold["nominal_od"], 0.1)
    np.testing.assert_allclose(xbar, gold["nominal_xbar"], rtol=0, atol=1e-13)


def test_condensed_qp_is_the_cvxpy_problem(ltv_oracle, ref_table):
    """build_qp's matrices against the loop evaluation of the objective and the constraints at random controls."""
    L = ltv_oracle
    T = 12
    st = ltv_states(6, seed=3)
    rng = np.random.default_rng(0)
    nom = rng.uniform(-0.3, 0.3, (6, T, 2))
    tgt = L.nearest_index(st[:, 0], st[:, 1], ref_table)
    xref = L.reference_window(ref_table, tgt, T)
    xbar = L.nominal_rollout(st, nom[:, :, 0], nom[:, :, 1], 0.1)
    qp = L.build_qp(st, xref, xbar, 0.1)
    for b in range(6):
        f0 = L.objective_loops(np.zeros((T, 2)), st[b], xref[b], xbar[b], 0.1)
        for _ in range(3):
            u = rng.uniform(-1, 1, (T, 2))
            f = L.objective_loops(u, st[b], xref[b], xbar[b], 0.1)
            uq = u.ravel()
            assert abs(0.5 * uq @ qp["H"][b] @ uq + qp["g"][b] @ uq + f0 - f) <= 1e-9 * max(1.0, abs(f))
            c_loop = np.sort(L.constraint_loops(u, st[b], xbar[b], 0.1))
            c_mat = np.sort(np.concatenate([qp["c0"][b] + qp["C"][b] @ uq, [st[b, 2], L.MAX_SPEED - st[b, 2]]]))
            np.testing.assert_allclose(c_mat, c_loop, rtol=0, atol=1e-12)


def test_solution_matches_slsqp_and_kkt(ltv_oracle, ref_table):
    from scipy.optimize import minimize
    L = ltv_oracle
    T = 10
    st = ltv_states(12, seed=11)
    st = st[(st[:, 2] > 0.5) & (st[:, 2] < 10.5)][:5]
    out = L.solve_batch(ref_table, st, np.zeros((len(st), T, 2)))
    assert (out["status"] == 0).all() and out["iters"].max() < 30
    for b in range(len(st)):
        x0, xref, xbar = st[b], out["xref"][b], out["xbar"][b]
        f = lambda u: L.objective_loops(u.reshape(T, 2), x0, xref, xbar, 0.1)
        cons = {"type": "ineq", "fun": lambda u: L.constraint_loops(u.reshape(T, 2), x0, xbar, 0.1)}
        r = minimize(f, np.full(2 * T, 0.05), constraints=[cons], method="SLSQP", options=dict(ftol=1e-15, maxiter=800))
        fo = f(out["U"][b].ravel())
        assert fo <= r.fun + 1e-6 * max(1.0, abs(r.fun))                 # the oracle's point is at least as good ...
        assert L.constraint_loops(out["U"][b], x0, xbar, 0.1).min() >= -1e-8    # ... and feasible
        if r.success or abs(fo - r.fun) <= 1e-7 * abs(fo):
            assert np.abs(r.x[:2] - out["u0"][b]).max() <= 2e-4
    # KKT certificate from the condensed data: stationarity with non-negative multipliers, complementarity
    qp = L.build_qp(st, out["xref"], out["xbar"], 0.1)
    u = out["U"].reshape(len(st), -1)
    c = qp["c0"] + np.einsum("bmn,bn->bm", qp["C"], u)
    grad = np.einsum("bkl,bl->bk", qp["H"], u) + qp["g"]
    assert c.min() >= -1e-8 and out["z"].min() >= 0.0
    assert np.abs(grad - np.einsum("bmn,bm->bn", qp["C"], out["z"])).max() <= 1e-5
    assert np.abs(c * out["z"]).max() <= 1e-6


def test_failure_and_warm_semantics(ltv_oracle, ref_table):
    L = ltv_oracle
    T = 20
    st = ltv_states(8, seed=2)
    st[0, 2] = L.MAX_SPEED + 0.5       # x[2, 0] == v0 violates v <= MAX_SPEED: the QP is infeasible
    st[1, 2] = 0.0                     # on the bound: feasible
    nom = np.random.default_rng(1).uniform(-0.2, 0.2, (8, T, 2))
    out = L.solve_batch(ref_table, st, nom)
    assert out["status"][0] == L.STATUS_INFEASIBLE and np.array_equal(out["u0"][0], [0.0, 0.0])
    assert np.array_equal(out["U"][0], nom[0])            # profile kept (pure_mpc_linear.py:193-196)
    assert out["status"][1] == 0 and out["u0"][1, 0] >= -1e-7
    ok = out["status"] == 0
    assert np.array_equal(out["u0"][ok], out["U"][ok][:, 0])
    # second call linearises about the first solution: different QP, still solved
    out2 = L.solve_batch(ref_table, st, out["U"])
    assert (out2["status"][1:] == 0).all()
    assert np.abs(out2["u0"][1:] - out["u0"][1:]).max() > 1e-6
