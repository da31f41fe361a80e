"""The iterative-linear MPC path on the MI355X (mpc_ltv_solve_batch / mpc_ltv_predict_batch through the C ABI) against
the CPU oracle (oracle/ltv_oracle.py) on the same seeded inputs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ltv_states, rel_u0_err

pytestmark = pytest.mark.gpu
TOL = 1e-4      # BASELINE north_star: controls within 1e-4 relative of the reference path


@pytest.fixture(scope="module")
def eng():
    from mpc_rl_for_avs_amd import engine
    e = engine.MPCEngine(horizon=20, max_iter=50)
    yield e
    e.close()


def test_solve_matches_oracle(eng, ltv_oracle, ref_table):
    st = ltv_states(512, seed=21)
    nom = np.zeros((512, 20, 2))
    got = eng.ltv_solve_batch(st, nom, want_traj=True)
    want = ltv_oracle.solve_batch(ref_table, st, nom)
    assert np.array_equal(got["status"], want["status"])
    assert np.array_equal(got["target_index"], want["target_index"])
    ok = want["status"] == 0
    assert ok.mean() > 0.85
    assert rel_u0_err(got["u0"], want["u0"])[ok].max() <= TOL
    assert np.abs(got["U"] - want["U"])[ok].max() <= 1e-3
    assert np.abs(got["X"] - want["X"])[ok].max() <= 1e-3
    d_it = np.abs(got["iters"] - want["iters"])[ok]          # same iteration; rounding decides the last steps
    assert d_it.mean() < 0.7 and d_it.max() <= 8
    assert np.array_equal(got["u0"][~ok], np.zeros_like(got["u0"][~ok]))
    assert np.array_equal(got["U"][~ok], nom[~ok])
    # a second round linearised about the first solutions, then a third
    for _ in range(2):
        nom = got["U"]
        got = eng.ltv_solve_batch(st, nom)
        want = ltv_oracle.solve_batch(ref_table, st, nom)
        ok = (want["status"] == 0) & (got["status"] == 0)
        assert np.array_equal(got["status"], want["status"])
        assert rel_u0_err(got["u0"], want["u0"])[ok].max() <= TOL


@pytest.mark.parametrize("N", [5, 16, 33, 64])
def test_other_horizons(ltv_oracle, ref_table, N):
    from mpc_rl_for_avs_amd import engine
    e = engine.MPCEngine(horizon=N, max_iter=50)
    st = ltv_states(96 if N <= 20 else 24, seed=300 + N)
    nom = np.zeros((len(st), N, 2))
    got = e.ltv_solve_batch(st, nom)
    want = ltv_oracle.solve_batch(ref_table, st, nom)
    ok = want["status"] == 0
    assert np.array_equal(got["status"], want["status"]) and ok.mean() > 0.8
    assert rel_u0_err(got["u0"], want["u0"])[ok].max() <= TOL
    e.close()


def test_batch_of_one_and_permutation(eng):
    st = ltv_states(64, seed=5)
    nom = np.random.default_rng(2).uniform(-0.2, 0.2, (64, 20, 2))
    full = eng.ltv_solve_batch(st, nom)
    one = eng.ltv_solve_batch(st[7:8], nom[7:8])
    assert np.array_equal(one["u0"][0], full["u0"][7]) and one["iters"][0] == full["iters"][7]
    perm = np.random.default_rng(3).permutation(64)
    shuf = eng.ltv_solve_batch(st[perm], nom[perm])
    assert np.array_equal(shuf["u0"], full["u0"][perm]) and np.array_equal(shuf["U"], full["U"][perm])


def test_two_builds_of_the_ltv_kernel_agree(eng):
    """mpc_ltv_kernel ships in two builds (mpc_engine.hip: launch_ltv): up to two waves per SIMD deep (B <= 2048 on 256
    CUs) the 215-register one, deeper the 165-register one that runs three waves per SIMD.  The same instances through
    both: statuses and iteration counts equal, profiles equal to 1e-9 (same statements; the compiler may contract a
    multiply-add differently in the two)."""
    st = ltv_states(4096, seed=11)
    nom = np.random.default_rng(4).uniform(-0.2, 0.2, (4096, 20, 2))
    deep = eng.ltv_solve_batch(st, nom)                       # 4096 instances: the three-wave build
    shallow = eng.ltv_solve_batch(st[:1024], nom[:1024])      # 1024: the latency build
    assert np.array_equal(deep["status"][:1024], shallow["status"])
    ok = shallow["status"] == 0
    assert ok.mean() > 0.85
    assert np.abs(deep["iters"][:1024] - shallow["iters"])[ok].max() <= 1
    assert np.abs(deep["U"][:1024] - shallow["U"])[ok].max() <= 1e-9
    assert np.array_equal(deep["u0"][:1024][~ok], shallow["u0"][~ok])


def test_torch_zero_copy(eng):
    import torch
    st = ltv_states(128, seed=9)
    host = eng.ltv_solve_batch(st, np.zeros((128, 20, 2)))
    dev = torch.device("cuda:0")
    t_state = torch.as_tensor(st, device=dev)
    t_U = torch.zeros((128, 20, 2), dtype=torch.float64, device=dev)
    out = eng.ltv_solve_batch_torch(t_state, t_U, sync=True)
    assert np.array_equal(out["u0"].cpu().numpy(), host["u0"])
    assert np.array_equal(t_U.cpu().numpy(), host["U"])
    assert np.array_equal(out["status"].cpu().numpy(), host["status"])
    with pytest.raises(ValueError):
        eng.ltv_solve_batch_torch(t_state.float(), t_U)


def _obs_from_state(st, rows=10):
    """observation rows [presence, x, y, vx, vy, heading, sin, cos] with the ego of state (x, y, v, yaw) in row 0"""
    B = st.shape[0]
    obs = np.zeros((B, rows, 8), dtype=np.float32)
    obs[:, 0, 0] = 1
    obs[:, 0, 1] = st[:, 0]
    obs[:, 0, 2] = st[:, 1]
    obs[:, 0, 3] = st[:, 2] * np.cos(st[:, 3])
    obs[:, 0, 4] = st[:, 2] * np.sin(st[:, 3])
    obs[:, 0, 5] = st[:, 3]
    obs[:, 0, 6] = np.sin(st[:, 3])
    obs[:, 0, 7] = np.cos(st[:, 3])
    obs[:, 1, 0] = 1
    obs[:, 1, 1:3] = (-20.0, 2.0)
    obs[:, 1, 3] = 8.0
    return obs


def test_predict_batch_is_the_agent_looped(ltv_oracle, ref_table):
    """mpc_ltv_predict_batch (device parse + per-environment stored profile) against B single-environment agents
    stepping the same observation sequences, and against the oracle fed the agents' parsed states."""
    from mpc_rl_for_avs_amd.pure_mpc_linear import IterativeLinearMPC_Agent

    class Env:
        config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}
    cfg = dict(horizon=20, render=False)
    B = 12
    batch = IterativeLinearMPC_Agent(Env, cfg)
    singles = [IterativeLinearMPC_Agent(Env, cfg, engine=batch._engine) for _ in range(B)]
    nom = np.zeros((B, 20, 2))
    for step in range(3):
        st = ltv_states(B, seed=40 + step)
        st[:, 2] = np.minimum(st[:, 2], 10.5)
        obs = _obs_from_state(st)
        obs[3, 0, 5] += np.float32(2 * np.pi) if step == 1 else 0      # heading beyond pi gets wrapped by the parse
        act = batch.predict_batch(obs)
        looped = np.stack([singles[b].predict(obs[b]) for b in range(B)])
        assert np.array_equal(act, looped)
        parsed = np.array([[a.ego_vehicle.position[0], a.ego_vehicle.position[1], a.ego_vehicle.speed, a.ego_vehicle.heading]
                           for a in singles], dtype=np.float64)
        want = ltv_oracle.solve_batch(ref_table, parsed, nom)
        assert (want["status"] == 0).all()
        assert rel_u0_err(act, want["u0"]).max() <= TOL
        nom = np.stack([np.stack([a.oa, a.od], axis=1) for a in singles])
    assert singles[0].target_ind == int(want["target_index"][0])
    # episode boundary for environments 2 and 5: their stored profiles are forgotten, the others keep theirs
    batch.reset_env_state([2, 5])
    obs = _obs_from_state(st)
    act = batch.predict_batch(obs)
    nom2 = nom.copy()
    nom2[[2, 5]] = 0.0
    want = ltv_oracle.solve_batch(ref_table, parsed, nom2)
    assert rel_u0_err(act, want["u0"]).max() <= TOL
    assert np.abs(act[2] - looped[2]).max() > 1e-9 or np.abs(nom[2]).max() < 1e-9


def test_failure_semantics(eng, ltv_oracle):
    st = ltv_states(4, seed=8)
    st[1, 2] = 11.5                                  # above MAX_SPEED: the QP has no feasible point
    nom = np.random.default_rng(4).uniform(-0.2, 0.2, (4, 20, 2))
    out = eng.ltv_solve_batch(st, nom)
    assert out["status"][1] == ltv_oracle.STATUS_INFEASIBLE and out["iters"][1] == 0
    assert np.array_equal(out["u0"][1], [0.0, 0.0]) and np.array_equal(out["U"][1], nom[1])
    assert (out["status"][[0, 2, 3]] == 0).all()
    with pytest.raises(ValueError):
        eng.ltv_solve_batch(st, nom[:, :10])


def test_hard_closed_loop_instances(eng, ltv_oracle, ref_table):
    d = np.load(os.path.join(GOLDEN, "ltv_closed_loop_hard.npz"))
    got = eng.ltv_solve_batch(d["state"], d["U"])
    want = ltv_oracle.solve_batch(ref_table, d["state"], d["U"])
    assert (want["status"] == 0).all() and (got["status"] == 0).all()
    assert rel_u0_err(got["u0"], want["u0"]).max() <= TOL


def test_gpu_solution_against_an_independent_solver(eng, ltv_oracle, ref_table):
    """The engine's controls against scipy SLSQP driven by the loop transcription of the cvxpy problem (no code shared
    with the oracle's matrices or its interior-point method): feasible, and no worse in objective."""
    from scipy.optimize import minimize
    L = ltv_oracle
    from mpc_rl_for_avs_amd import engine
    T = 10
    e = engine.MPCEngine(horizon=T, max_iter=50)
    st = ltv_states(16, seed=12)
    st = np.ascontiguousarray(st[(st[:, 2] > 0.5) & (st[:, 2] < 10.5)][:4])
    nom = np.zeros((len(st), T, 2))
    got = e.ltv_solve_batch(st, nom)
    assert (got["status"] == 0).all()
    tgt = L.nearest_index(st[:, 0], st[:, 1], ref_table)
    xref = L.reference_window(ref_table, tgt, T)
    xbar = L.nominal_rollout(st, nom[:, :, 0], nom[:, :, 1], 0.1)
    for b in range(len(st)):
        f = lambda u: L.objective_loops(u.reshape(T, 2), st[b], xref[b], xbar[b], 0.1)
        cons = {"type": "ineq", "fun": lambda u: L.constraint_loops(u.reshape(T, 2), st[b], xbar[b], 0.1)}
        r = minimize(f, np.full(2 * T, 0.05), constraints=[cons], method="SLSQP", options=dict(ftol=1e-15, maxiter=800))
        fg = f(got["U"][b].ravel())
        assert L.constraint_loops(got["U"][b], st[b], xbar[b], 0.1).min() >= -1e-8
        assert fg <= r.fun + 1e-6 * max(1.0, abs(r.fun))
        if r.success or abs(fg - r.fun) <= 1e-7 * abs(fg):
            assert np.abs(r.x[:2] - got["u0"][b]).max() <= 2e-4
    e.close()


def test_golden_solutions(eng):
    g = np.load(os.path.join(GOLDEN, "ltv_oracle_solutions.npz"))
    got = eng.ltv_solve_batch(g["state"], g["U0"])
    ok = g["status_first"] == 0
    assert np.array_equal(got["status"], g["status_first"]) and np.array_equal(got["target_index"], g["target_index"])
    assert rel_u0_err(got["u0"], g["u0_first"])[ok].max() <= TOL and np.abs(got["U"] - g["U_first"])[ok].max() <= 1e-3
    got2 = eng.ltv_solve_batch(g["state"], g["U_first"])
    ok2 = g["status_second"] == 0
    assert np.array_equal(got2["status"], g["status_second"])
    assert rel_u0_err(got2["u0"], g["u0_second"])[ok2].max() <= TOL


def test_independent_fixtures(eng, ref_table):
    """The engine against exact solutions of the QP from an independent method (Goldfarb-Idnani active set on the loop
    transcription of the cvxpy statements of agents/pure_mpc_linear.py:205-257, tests/golden/make_ltv_independent.py):
    168 instances at T = 20 (first calls, random stored profiles, second calls), 48 at T = 12."""
    from mpc_rl_for_avs_amd import engine
    fx = np.load(os.path.join(GOLDEN, "ltv_independent_solutions.npz"))
    for T in (20, 12):
        e = eng if T == 20 else engine.MPCEngine(horizon=T, max_iter=50)
        st, nom, U = fx[f"state_T{T}"], fx[f"nominal_T{T}"], fx[f"U_T{T}"]
        got = e.ltv_solve_batch(st, nom)
        assert (got["status"] == 0).all() and np.array_equal(got["target_index"], fx[f"target_index_T{T}"])
        err = np.abs(got["U"] - U).reshape(len(st), -1).max(axis=1)
        assert err.max() <= 5e-5 and np.percentile(err, 90) <= 1e-6
        assert rel_u0_err(got["u0"], U[:, 0]).max() <= 1e-5
        if T != 20:
            e.close()


def test_full_batch_certified(eng, ltv_oracle, ref_table):
    """Every solved instance of a 4096 batch (first call, then the call linearised about its result) carries a KKT
    certificate whose multipliers are fitted independently of the solver (NNLS on the rows within 1e-5 of their bound):
    feasible to 1e-9, stationarity <= 1e-6 relative - for a strictly convex QP that pins the minimiser."""
    import qp_active_set as Q
    L = ltv_oracle
    B = 4096
    st = ltv_states(B, seed=77)
    nom = np.zeros((B, 20, 2))
    for call in range(2):
        got = eng.ltv_solve_batch(st, nom)
        ok = got["status"] == 0
        assert ((got["status"] == 0) | (got["status"] == L.STATUS_INFEASIBLE)).all() and ok.mean() > 0.85
        tgt = L.nearest_index(st[:, 0], st[:, 1], ref_table)
        assert np.array_equal(tgt, got["target_index"])
        worst_s = worst_v = 0.0
        for lo in range(0, B, 512):
            sl = slice(lo, lo + 512)
            xref = L.reference_window(ref_table, tgt[sl], 20)
            xbar = L.nominal_rollout(st[sl], nom[sl, :, 0], nom[sl, :, 1], 0.1)
            x0 = st[sl].copy()
            x0[:, 2] = np.clip(x0[:, 2], 0.0, L.MAX_SPEED)
            qp = L.build_qp(x0, xref, xbar, 0.1)
            for i in np.nonzero(ok[sl])[0]:
                s, v, _ = Q.certify(qp["H"][i], qp["g"][i], qp["C"][i], qp["c0"][i], got["U"][lo + i].ravel())
                worst_s, worst_v = max(worst_s, s), max(worst_v, v)
        print(f"LTV call {call + 1}: {int(ok.sum())} of {B} certified, stationarity max {worst_s:.2e}, violation max {worst_v:.2e}")
        assert worst_s <= 1e-6 and worst_v <= 1e-9
        nom = got["U"]


def test_linearisation_passes(ltv_oracle, ref_table):
    """mpc_config.ltv_passes = the trip count of the loop at agents/pure_mpc_linear.py:189."""
    from mpc_rl_for_avs_amd import engine
    from mpc_rl_for_avs_amd.pure_mpc_linear import IterativeLinearMPC_Agent
    st = ltv_states(256, seed=17)
    nom = np.zeros((256, 20, 2))
    e1 = engine.MPCEngine(horizon=20, max_iter=50)
    e3 = engine.MPCEngine(horizon=20, max_iter=50, ltv_passes=3)
    got = e3.ltv_solve_batch(st, nom)
    want = ltv_oracle.solve_batch(ref_table, st, nom, passes=3)
    ok = want["status"] == 0
    assert np.array_equal(got["status"], want["status"]) and ok.mean() > 0.8
    assert rel_u0_err(got["u0"], want["u0"])[ok].max() <= TOL
    u, its = nom, np.zeros(256, dtype=np.int64)
    for _ in range(3):
        o = e1.ltv_solve_batch(st, u)
        u, its = o["U"], its + o["iters"]
    assert np.array_equal(got["U"], o["U"]) and np.array_equal(got["u0"], o["u0"]) and np.array_equal(got["iters"], its)
    # a capped second pass: action (0, 0), the first pass's profile stays stored
    one = e1.ltv_solve_batch(st, nom)
    cap = int(one["iters"][ok].max())
    ec1 = engine.MPCEngine(horizon=20, max_iter=cap)
    ec2 = engine.MPCEngine(horizon=20, max_iter=cap, ltv_passes=2)
    two = ec2.ltv_solve_batch(st, nom)
    second = ec1.ltv_solve_batch(st, one["U"])
    failed = ok & (second["status"] != 0)
    assert failed.any()
    assert np.array_equal(two["U"][failed], one["U"][failed]) and not two["u0"][failed].any()
    with pytest.raises(engine.EngineError):
        engine.MPCEngine(horizon=20, ltv_passes=0)

    class Env:
        config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}
    agent = IterativeLinearMPC_Agent(Env, dict(horizon=20, render=False, linearization_passes=3))
    s1 = st[ok][:1].copy()
    s1[:, 2] = min(s1[0, 2], 10.5)
    act = agent.predict(_obs_from_state(s1)[0])
    parsed = np.array([[agent.ego_vehicle.position[0], agent.ego_vehicle.position[1], agent.ego_vehicle.speed,
                        agent.ego_vehicle.heading]], dtype=np.float64)
    want1 = ltv_oracle.solve_batch(ref_table, parsed, np.zeros((1, 20, 2)), passes=3)
    assert rel_u0_err(act[None], want1["u0"]).max() <= TOL and np.abs(np.stack([agent.oa, agent.od], 1) - want1["U"][0]).max() <= 1e-3
    for e in (e1, e3, ec1, ec2):
        e.close()


def test_bounds_active_in_the_same_stage(ltv_oracle, ref_table):
    """Acceleration bound and speed bound active in the same stage (and a degenerate speed bound at the last node)
    against the exact minimiser; see tests/test_ltv_cpu.py."""
    from mpc_rl_for_avs_amd import engine
    from test_ltv_cpu import same_stage_cases, check_same_stage
    ref, st, nom, exact = same_stage_cases(ltv_oracle, ref_table, B=64)
    e = engine.MPCEngine(horizon=20, max_iter=50, ref_table=ref)
    check_same_stage(e.ltv_solve_batch(st, nom, want_traj=True), exact, ltv_oracle)
    e.close()
