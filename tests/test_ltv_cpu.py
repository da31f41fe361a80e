"""The LTV-QP kernel solver (mpc_ltv.hpp: one wave per instance), compiled for the host with its lanes emulated
(tests/cpu_ltv_harness.cpp), against the CPU oracle (oracle/ltv_oracle.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ltv_states, rel_u0_err

TOL = 1e-4      # BASELINE north_star: controls within 1e-4 relative of the reference path


@pytest.mark.parametrize("N", [5, 16, 20, 33, 64])
def test_matches_oracle(cpu_ltv, ltv_oracle, ref_table, N):
    st = ltv_states(48 if N <= 20 else 12, seed=100 + N)
    nom = np.zeros((len(st), N, 2))
    got = cpu_ltv(ref_table, st, nom, N=N)
    want = ltv_oracle.solve_batch(ref_table, st, nom)
    assert np.array_equal(got["status"], want["status"])
    assert np.array_equal(got["target_index"], want["target_index"])
    ok = want["status"] == 0
    assert ok.mean() > 0.8
    assert rel_u0_err(got["u0"], want["u0"])[ok].max() <= TOL
    assert np.abs(got["U"] - want["U"])[ok].max() <= 1e-3
    assert np.abs(got["X"] - want["X"])[ok].max() <= 1e-3
    assert np.abs(got["iters"] - want["iters"])[ok].max() <= 3      # same iteration, rounding decides the last step
    # where no QP was solved: action (0, 0), stored profile untouched
    assert np.array_equal(got["u0"][~ok], np.zeros_like(got["u0"][~ok]))
    assert np.array_equal(got["U"][~ok], nom[~ok])


def test_second_call_linearises_about_the_first(cpu_ltv, ltv_oracle, ref_table):
    st = ltv_states(64, seed=7)
    first = cpu_ltv(ref_table, st, np.zeros((64, 20, 2)))
    got = cpu_ltv(ref_table, st, first["U"])
    want = ltv_oracle.solve_batch(ref_table, st, first["U"])
    ok = (want["status"] == 0) & (got["status"] == 0)
    assert np.array_equal(got["status"], want["status"]) and ok.mean() > 0.8
    assert rel_u0_err(got["u0"], want["u0"])[ok].max() <= TOL
    assert np.abs(got["u0"] - first["u0"])[ok].max() > 1e-3       # the model did change


def test_edge_cases(cpu_ltv, ltv_oracle, ref_table):
    L = ltv_oracle
    st = ltv_states(6, seed=4)
    st[0, 2] = 0.0                       # standing still: on the lower speed bound
    st[1, 2] = np.float32(L.MAX_SPEED)   # float32 rounding of 40/3.6 is just below it: feasible
    st[2, 2] = 11.5                      # above MAX_SPEED: infeasible
    st[3, :2] = [200.0, 200.0]           # far from the path: huge terminal cost
    st[4, 3] = 3.1                       # heading opposite to the path's -pi/2..-pi
    nom = np.random.default_rng(5).uniform(-0.5, 0.5, (6, 20, 2))
    nom[5, :, 0] = 9.0                   # a stored profile far outside the bounds only moves the operating point
    got = cpu_ltv(ref_table, st, nom)
    want = L.solve_batch(ref_table, st, nom)
    assert np.array_equal(got["status"], want["status"])
    assert got["status"][2] == L.STATUS_INFEASIBLE and got["iters"][2] == 0
    ok = want["status"] == 0
    assert ok.sum() == 5
    assert rel_u0_err(got["u0"], want["u0"])[ok].max() <= TOL
    assert got["u0"][0, 0] >= -1e-6      # cannot decelerate below zero speed
    # bounds hold on what is returned
    U = got["U"][ok]
    assert U[:, :, 0].min() >= L.MAX_DECEL - 1e-7 and U[:, :, 0].max() <= L.MAX_ACCEL + 1e-7
    assert np.abs(U[:, :, 1]).max() <= L.MAX_STEER + 1e-7
    assert np.abs(np.diff(U[:, :, 1], axis=1)).max() <= L.MAX_DSTEER * 0.1 + 1e-7
    assert got["X"][ok][:, :, 2].min() >= -1e-7 and got["X"][ok][:, :, 2].max() <= L.MAX_SPEED + 1e-7


def test_hard_closed_loop_instances(cpu_ltv, ltv_oracle, ref_table):
    """Regression set: (state, stored profile) pairs met in closed loop far off the path (tools/gpu_ltv_closed_loop.py
    with LTV_DUMP), whose multipliers are ~1e5: the measured dual residual is rounding noise long before mu reaches its
    tolerance there, which an absolute dual tolerance never survived."""
    d = np.load(os.path.join(GOLDEN, "ltv_closed_loop_hard.npz"))
    got = cpu_ltv(ref_table, d["state"], d["U"])
    want = ltv_oracle.solve_batch(ref_table, d["state"], d["U"])
    assert (want["status"] == 0).all() and (got["status"] == 0).all()
    assert rel_u0_err(got["u0"], want["u0"]).max() <= TOL
    assert got["iters"].max() <= 30


def test_golden_solutions(cpu_ltv, ltv_oracle, ref_table):
    """Committed oracle solutions (tests/golden/ltv_oracle_solutions.npz): the oracle still reproduces them, the kernel
    solver matches them, and they are as good as an independent SLSQP run where that was recorded."""
    g = np.load(os.path.join(GOLDEN, "ltv_oracle_solutions.npz"))
    now = ltv_oracle.solve_batch(ref_table, g["state"], g["U0"])
    assert np.array_equal(now["status"], g["status_first"]) and np.array_equal(now["target_index"], g["target_index"])
    ok = g["status_first"] == 0
    assert np.abs(now["u0"] - g["u0_first"])[ok].max() <= 1e-7
    got = cpu_ltv(ref_table, g["state"], g["U0"])
    assert np.array_equal(got["status"], g["status_first"])
    assert rel_u0_err(got["u0"], g["u0_first"])[ok].max() <= TOL and np.abs(got["U"] - g["U_first"])[ok].max() <= 1e-3
    got2 = cpu_ltv(ref_table, g["state"], g["U_first"])
    ok2 = g["status_second"] == 0
    assert np.array_equal(got2["status"], g["status_second"])
    assert rel_u0_err(got2["u0"], g["u0_second"])[ok2].max() <= TOL
    rec = np.isfinite(g["slsqp_objective"])
    assert rec.sum() >= 4
    assert (g["oracle_objective"][rec] <= g["slsqp_objective"][rec] + 1e-6 * np.abs(g["slsqp_objective"][rec])).all()


def test_independent_fixtures(cpu_ltv, ref_table):
    """The kernel solver against exact solutions of the QP from an independent method (Goldfarb-Idnani active set on the
    loop transcription of the cvxpy statements, tests/golden/make_ltv_independent.py) - no interior point, no oracle."""
    fx = np.load(os.path.join(GOLDEN, "ltv_independent_solutions.npz"))
    for T in (20, 12):
        st, nom, U = fx[f"state_T{T}"], fx[f"nominal_T{T}"], fx[f"U_T{T}"]
        got = cpu_ltv(ref_table, st, nom, N=T)
        assert (got["status"] == 0).all() and np.array_equal(got["target_index"], fx[f"target_index_T{T}"])
        err = np.abs(got["U"] - U).reshape(len(st), -1).max(axis=1)
        assert err.max() <= 5e-5 and np.percentile(err, 90) <= 1e-6     # see tests/test_ltv_oracle.py for the measured values
        assert rel_u0_err(got["u0"], U[:, 0]).max() <= 1e-5


def test_linearisation_passes(cpu_ltv, ltv_oracle, ref_table):
    """LtvParams.passes = the trip count of the loop at agents/pure_mpc_linear.py:189."""
    st = ltv_states(32, seed=17)
    nom = np.zeros((32, 20, 2))
    got = cpu_ltv(ref_table, st, nom, passes=3)
    want = ltv_oracle.solve_batch(ref_table, st, nom, passes=3)
    ok = want["status"] == 0
    assert np.array_equal(got["status"], want["status"]) and ok.mean() > 0.8
    assert rel_u0_err(got["u0"], want["u0"])[ok].max() <= TOL and np.abs(got["iters"] - want["iters"])[ok].max() <= 6
    # the same as three calls feeding the profile back, bit for bit
    u, its = nom, np.zeros(32, dtype=np.int64)
    for _ in range(3):
        o = cpu_ltv(ref_table, st, u)
        u, its = o["U"], its + o["iters"]
    assert np.array_equal(got["U"], o["U"]) and np.array_equal(got["u0"], o["u0"]) and np.array_equal(got["iters"], its)
    # first pass solved, second one capped: action (0, 0), the first pass's profile stays stored
    one = cpu_ltv(ref_table, st, nom)
    cap = int(one["iters"][ok].max())
    two = cpu_ltv(ref_table, st, nom, passes=2, max_iter=cap)
    second = cpu_ltv(ref_table, st, one["U"], max_iter=cap)
    failed = ok & (second["status"] != 0)
    assert failed.any()
    assert np.array_equal(two["U"][failed], one["U"][failed]) and not two["u0"][failed].any()
    assert (two["status"][failed] == 1).all()


def same_stage_cases(ltv_oracle, ref_table, B=24):
    """A reference speed above MAX_SPEED and ego speeds 0.2 k below it: the optimum accelerates at the input bound for k
    stages and lands exactly on the speed bound - acceleration bound and speed bound active in the same stage, and the
    speed bound of the LAST node degenerate (multiplier exactly 0: the terminal cost has no speed term, Qf[2] = 0).
    Returns the table, states and the exact minimisers (Goldfarb-Idnani on the condensed QP)."""
    import qp_active_set as Q
    L = ltv_oracle
    ref = np.array(ref_table, copy=True)
    ref[:, 2] = 15.0
    st = ltv_states(B, seed=91)
    st[:, 2] = np.clip(L.MAX_SPEED - 0.2 * np.random.default_rng(3).integers(0, 16, B), 0.0, L.MAX_SPEED)
    nom = np.zeros((B, 20, 2))
    tgt = L.nearest_index(st[:, 0], st[:, 1], ref)
    qp = L.build_qp(st, L.reference_window(ref, tgt, 20), L.nominal_rollout(st, nom[:, :, 0], nom[:, :, 1], 0.1), 0.1)
    exact = np.stack([Q.solve(qp["H"][b], qp["g"][b], qp["C"][b], -qp["c0"][b])[0].reshape(20, 2) for b in range(B)])
    return ref, st, nom, exact


def check_same_stage(got, exact, L):
    assert (got["status"] == 0).all() and got["iters"].max() <= 30
    a, v = got["U"][:, :, 0], got["X"][:, 1:, 2]
    both = (np.abs(a - L.MAX_ACCEL) < 1e-6) & (v > L.MAX_SPEED - 1e-6)
    assert both.any(axis=1).mean() > 0.8                          # the case is present
    err = np.abs(got["U"] - exact)
    # everything but the last acceleration is exact to rounding; that one sits sqrt(s z / H_aa) below its degenerate
    # bound (measured 2.6e-4 ... 5.9e-4, the same in the dense-LU oracle): the central path, not the factorisation
    assert err[:, :-1].max() <= 1e-9 and err[:, -1, 1].max() <= 1e-9
    assert err[:, -1, 0].max() <= 1e-3
    assert np.abs(got["u0"] - exact[:, 0]).max() <= 1e-9


def test_bounds_active_in_the_same_stage(cpu_ltv, ltv_oracle, ref_table):
    ref, st, nom, exact = same_stage_cases(ltv_oracle, ref_table)
    check_same_stage(cpu_ltv(ref, st, nom), exact, ltv_oracle)
    check_same_stage(ltv_oracle.solve_batch(ref, st, nom), exact, ltv_oracle)


def test_both_builds_take_the_same_path(cpu_ltv, ref_table):
    """mpc_ltv_kernel ships in two builds (mpc_engine.hip: launch_ltv): the one for shallow batches keeps the primal
    residuals in registers and forms the gain rows of a Riccati stage once per wave, the one for three waves per SIMD
    recomputes the residuals and lets every lane select and scale its own entries (mpc_ltv.hpp: relax_bits).  Same
    arithmetic in the same order: identical results."""
    st = ltv_states(64, seed=321)
    nom = np.zeros((len(st), 20, 2))
    a = cpu_ltv(ref_table, st, nom, passes=2)
    b = cpu_ltv(ref_table, st, nom, passes=2, relaxed=True)
    assert (a["status"] == 0).mean() > 0.8
    for k in ("status", "iters", "target_index", "u0", "U", "X"):
        assert np.array_equal(a[k], b[k], equal_nan=k == "X"), k      # X is not written where no QP was posed (status 3)
