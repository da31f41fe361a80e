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


# ---------------------------------------------------------------------------------------------------------------
# the independent exact QP solver (oracle/qp_active_set.py) and the fixtures made with it
# ---------------------------------------------------------------------------------------------------------------
def _brute_force_qp(H, g, A, b):
    """all active sets of a tiny QP: the feasible KKT point with non-negative multipliers"""
    import itertools
    n, m = H.shape[0], A.shape[0]
    best = None
    for q in range(0, n + 1):
        for act in itertools.combinations(range(m), q):
            N = A[list(act)]
            K = np.block([[H, -N.T], [N, np.zeros((q, q))]])
            try:
                sol = np.linalg.solve(K, np.concatenate([-g, b[list(act)]]))
            except np.linalg.LinAlgError:
                continue
            x, lam = sol[:n], sol[n:]
            if (A @ x - b).min() >= -1e-9 and (q == 0 or lam.min() >= -1e-9):
                f = 0.5 * x @ H @ x + g @ x
                if best is None or f < best[1]:
                    best = (x, f)
    return best[0]


def test_active_set_solver_known_answers():
    import qp_active_set as Q
    # box: the minimiser of |x - t|^2 over [-1, 1]^3 is the clipped target
    t = np.array([2.0, -0.3, -5.0])
    A = np.concatenate([np.eye(3), -np.eye(3)])
    x, mult, act = Q.solve(2.0 * np.eye(3), -2.0 * t, A, -np.ones(6))
    assert np.allclose(x, np.clip(t, -1, 1), atol=1e-14) and sorted(act) == [2, 3]
    assert np.allclose(mult, [0, 0, 8.0, 2.0, 0, 0], atol=1e-13)            # 2 (x - t) = A' mult
    # random tiny QPs against enumeration of every active set, including dependent constraints
    rng = np.random.default_rng(4)
    for trial in range(40):
        n, m = 3, 6
        R = rng.normal(size=(n, n))
        H = R @ R.T + 0.1 * np.eye(n)
        g = rng.normal(size=n) * 3
        A = rng.normal(size=(m, n))
        if trial % 4 == 0:
            A[5] = A[4] * 2.0                 # a redundant pair
        b = A @ rng.normal(size=n) * 0.2 - rng.uniform(0.0, 1.0, m)   # x = that point is strictly feasible
        x, mult, act = Q.solve(H, g, A, b)
        assert np.abs(x - _brute_force_qp(H, g, A, b)).max() <= 1e-9
        assert np.abs(H @ x + g - A.T @ mult).max() <= 1e-9 and mult.min() >= 0.0
    with pytest.raises(Q.Infeasible):
        Q.solve(np.eye(2), np.zeros(2), np.array([[1.0, 0.0], [-1.0, 0.0]]), np.array([1.0, 1.0]))   # x >= 1 and x <= -1


def test_loop_built_qp_equals_condensed_qp(ltv_oracle, ref_table):
    """`build_from_loops` (polarisation of the loop transcription) against `build_qp` (condensed matrices)."""
    import qp_active_set as Q
    L, T = ltv_oracle, 8
    st = ltv_states(8, seed=9)
    st = st[(st[:, 2] > 0.0) & (st[:, 2] < L.MAX_SPEED)][:2]
    nom = np.random.default_rng(2).uniform(-0.3, 0.3, (2, T, 2))
    tgt = L.nearest_index(st[:, 0], st[:, 1], ref_table)
    xref = L.reference_window(ref_table, tgt, T)
    xbar = L.nominal_rollout(st, nom[:, :, 0], nom[:, :, 1], 0.1)
    qp = L.build_qp(st, xref, xbar, 0.1)
    for b in range(2):
        H, g, A, bb, _ = Q.build_from_loops(L, st[b], xref[b], xbar[b], 0.1, T)
        scale = np.abs(qp["H"][b]).max()
        assert np.abs(H - qp["H"][b]).max() <= 1e-9 * scale and np.abs(g - qp["g"][b]).max() <= 1e-9 * np.abs(g).max()
        # same constraint set, row order aside
        mine = np.concatenate([A, -bb[:, None]], axis=1)
        theirs = np.concatenate([qp["C"][b], qp["c0"][b][:, None]], axis=1)
        key = lambda M: M[np.lexsort(np.round(M, 9).T[::-1])]
        np.testing.assert_allclose(key(mine), key(theirs), rtol=0, atol=1e-11)


def test_independent_fixtures(ltv_oracle, ref_table):
    """tests/golden/ltv_independent_solutions.npz (make_ltv_independent.py: Goldfarb-Idnani on the loop-built QP): a few
    entries are regenerated, and the interior-point oracle lands on every one of them."""
    import qp_active_set as Q
    L = ltv_oracle
    fx = np.load(os.path.join(GOLDEN, "ltv_independent_solutions.npz"))
    for T in (20, 12):
        st, nom, U = fx[f"state_T{T}"], fx[f"nominal_T{T}"], fx[f"U_T{T}"]
        assert len(st) >= (128 if T == 20 else 32)
        for b in (0, 1, 2, len(st) - 1):
            tgt = L.nearest_index(st[b:b + 1, 0], st[b:b + 1, 1], ref_table)
            xref = L.reference_window(ref_table, tgt, T)[0]
            xbar = L.nominal_rollout(st[b:b + 1], nom[b:b + 1, :, 0], nom[b:b + 1, :, 1], 0.1)[0]
            H, g, A, bb, _ = Q.build_from_loops(L, st[b], xref, xbar, 0.1, T)
            x, _, act = Q.solve(H, g, A, bb)
            assert np.abs(x.reshape(T, 2) - U[b]).max() <= 1e-10 and len(act) == fx[f"n_active_T{T}"][b]
        out = L.solve_batch(ref_table, st, nom)
        assert (out["status"] == 0).all() and np.array_equal(out["target_index"], fx[f"target_index_T{T}"])
        err = np.abs(out["U"] - U).reshape(len(st), -1).max(axis=1)
        # measured: max 1.5e-5 (a weakly active row: the iterate with mu <= 1e-10 sits ~sqrt(mu) from the exact vertex),
        # 90 % of the instances below 2e-8; first controls 5e-7
        assert err.max() <= 5e-5 and np.percentile(err, 90) <= 1e-6
        assert np.abs(out["u0"] - U[:, 0]).max() <= 1e-5
        # certificates of the oracle's points, multipliers fitted independently (NNLS)
        qp = L.build_qp(st, out["xref"], out["xbar"], 0.1)
        cert = np.array([Q.certify(qp["H"][b], qp["g"][b], qp["C"][b], qp["c0"][b], out["U"][b].ravel())
                         for b in range(len(st))])
        assert cert[:, 0].max() <= 1e-6 and cert[:, 1].max() <= 1e-9


def test_passes_are_sequential_calls(ltv_oracle, ref_table):
    """passes = n is the loop of agents/pure_mpc_linear.py:189 run n times: the same as n calls feeding the profile back,
    iteration counts summed; a failing pass keeps the last stored profile and returns (0, 0)."""
    L = ltv_oracle
    st = ltv_states(24, seed=31)
    nom = np.zeros((24, 20, 2))
    three = L.solve_batch(ref_table, st, nom, passes=3)
    u, its = nom, np.zeros(24, dtype=np.int64)
    for _ in range(3):
        o = L.solve_batch(ref_table, st, u)
        u, its = o["U"], its + o["iters"]
    assert np.array_equal(three["status"], o["status"]) and np.array_equal(three["iters"], its)
    assert np.array_equal(three["U"], o["U"]) and np.array_equal(three["u0"], o["u0"])
    # a pass that cannot finish (iteration cap 3 after a solved first pass is emulated by max_iter on the whole call)
    capped = L.solve_batch(ref_table, st, nom, passes=2, max_iter=3)
    assert (capped["status"][capped["status"] != L.STATUS_INFEASIBLE] == L.STATUS_MAX_ITER).all()
    assert np.array_equal(capped["U"], nom) and not capped["u0"].any()
