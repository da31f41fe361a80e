"""The kernel's solver (mpc_wave.hpp: one wave per instance), compiled for the host, against the oracle (CPU only).  The
64 lanes are emulated by loops over the phases, the matrix core, lane permutations and reductions by host models
(tests/cpu_wave_harness.cpp).  Same arithmetic as the HIP kernel minus the lean device math (frcp/frsqrt use 1/x,
1/sqrt)."""
import numpy as np
import pytest

from conftest import rel_u0_err


@pytest.fixture()
def cpu_core(cpu_wave):
    return cpu_wave


@pytest.mark.parametrize("V,cc", [(4, False), (8, True)])
def test_core_matches_oracle(cpu_core, oracle, ref_table, V, cc):
    from mpc_rl_for_avs_amd import synth
    inp = synth.solver_inputs(192, V, seed=3)
    want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                              vref=inp["vref"], others=inp["others"], collision_cost=cc, max_iter=100, xy_bounds=False)
    got = cpu_core(ref_table, inp, collision_cost=cc)
    both = (got["status"] == 0) & (want["status"] == 0)
    assert both.mean() > 0.85
    assert (got["status"] == want["status"]).mean() > 0.97
    err = rel_u0_err(got["u0"], want["u0"])[both]
    assert (err <= 1e-4).mean() >= 0.99         # north_star tolerance; a handful of ill-posed instances may flip
    assert np.percentile(err, 95) < 1e-9
    assert (got["iters"] == want["iters"])[both].mean() > 0.95


def test_core_edge_cases(cpu_core, oracle, ref_table):
    inp = dict(state=np.array([[2.0, 45.0, -np.pi / 2, 0.0], [2.0, 45.0, -np.pi / 2, 10.0]]),
               ego_index=np.array([4, 4], np.int32), weights=np.ones((2, 3)), is_collide=np.zeros(2, np.uint8),
               vref=None, others=None)
    for N in (5, 16, 20, 32, 33, 64):
        got = cpu_core(ref_table, inp, N=N)
        want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], N=N,
                                  max_iter=100, xy_bounds=False)
        assert np.array_equal(got["status"], want["status"])
        assert rel_u0_err(got["u0"], want["u0"]).max() < 1e-8
        if N <= 32:                                      # still on the straight: on the reference at reference speed
            assert np.abs(got["U"][1]).max() < 1e-7      # the optimal controls are zero
    # collision cost requested but no vehicles
    inp["others"] = np.zeros((2, 0, 4))
    got = cpu_core(ref_table, inp, collision_cost=True)
    assert np.all(got["status"] == 0)
    # infeasible start is flagged
    inp2 = dict(inp, state=np.array([[2.0, 45.0, -np.pi / 2, 31.0]]), ego_index=np.array([4], np.int32),
                weights=np.ones((1, 3)), is_collide=np.zeros(1, np.uint8), others=None)
    assert cpu_core(ref_table, inp2)["status"][0] == 3


def test_wave_core_warm_start_matches_oracle(cpu_wave, oracle, ref_table):
    """Opt-in warm start (initial controls instead of the reference's cold start): same iterates as the oracle given the
    same initial controls, including the clamp into the bounds and the fall-back to the cold start."""
    from mpc_rl_for_avs_amd import synth
    inp = synth.solver_inputs(96, 8, seed=6)
    kw = dict(vref=inp["vref"], others=inp["others"], collision_cost=True, max_iter=100, xy_bounds=False)
    cold = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], **kw)
    rng = np.random.default_rng(0)
    shifted = np.concatenate([cold["U"][:, 1:], cold["U"][:, -1:]], axis=1)
    shifted[::7] = rng.uniform(-8.0, 8.0, shifted[::7].shape)          # some wild ones: clamped, maybe infeasible
    want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                              u_init=shifted, **kw)
    got = cpu_wave(ref_table, inp, collision_cost=True, u_init=shifted)
    both = (got["status"] == 0) & (want["status"] == 0)
    assert both.mean() > 0.8 and (got["status"] == want["status"]).mean() > 0.97
    assert (rel_u0_err(got["u0"], want["u0"])[both] <= 1e-4).mean() >= 0.99
    assert (got["iters"] == want["iters"])[both].mean() > 0.95
    assert want["iters"][both].mean() < cold["iters"][both].mean()


def heading_on_the_bound_inputs(ref_table):
    """An ego on the exit straight (reference heading -pi, rows 62.. of the table) whose observed heading is +-pi to
    float32 rounding: 5.6e-8 outside the NLP's relaxed bound [-pi, pi] (1 + 1e-8).  IPOPT accepts that at node 0 as a
    constraint violation below its tolerance and pushes its start inside; zero controls keep every node of a
    single-shooting start on the bound."""
    rows = np.array([64, 66, 70, 74, 78, 80, 66, 70])
    th = np.float64(np.float32(-np.pi)) * np.ones(len(rows))
    th[-2:] = np.float64(np.float32(np.pi))                 # the same heading, wrapped to the other side
    v = np.array([10.0, 9.9, 8.0, 5.0, 2.0, 0.5, 10.0, 6.0])
    state = np.stack([ref_table[rows, 0], ref_table[rows, 1] + 0.05, th, v], axis=1)
    return dict(state=state, ego_index=rows.astype(np.int32), weights=np.ones((len(rows), 3)),
                is_collide=np.zeros(len(rows), np.uint8), vref=None, others=None)


def test_heading_on_the_bound(cpu_core, oracle, ref_table):
    import kkt_batch as kb
    import nlp_batch as nb
    inp = heading_on_the_bound_inputs(ref_table)
    assert (np.abs(inp["state"][:, 2]) > np.pi * (1 + 1e-8)).all()
    want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], max_iter=100,
                              xy_bounds=False)
    got = cpu_core(ref_table, inp)
    assert np.array_equal(got["status"], want["status"]) and (want["status"][:6] == 0).all()      # was 3 for all of them
    assert rel_u0_err(got["u0"], want["u0"])[want["status"] == 0].max() < 1e-7
    # every later node is strictly inside, and the points are KKT points of the reference NLP (theta_0 itself is data)
    ok = want["status"] == 0
    assert (np.abs(want["X"][ok][:, 1:, 2]) < np.pi * (1 + 1e-8)).all()
    p = nb.Batch.build(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"])
    cert = kb.certify(p.take(np.nonzero(ok)[0]), want["X"][ok], want["U"][ok])
    assert cert["stationarity"].max() <= 1e-8 and cert["feasibility"].max() <= 1e-10


def test_stall_window_guard(cpu_wave, oracle, ref_table):
    """mpc_config.stall_window (off by default): a solve whose KKT error has not halved within W iterations ends with
    status 4.  Host build of the kernel source == oracle, instances that converge quickly are untouched.  (Until round 4 the
    slowest instances of this batch ran to the reference's max_iter 1000 and W = 64 was what bounded a batch; with round 5's
    globalisation - DESIGN.md section 2 (vii) - (x) - the slowest needs 150 iterations, so the guard is exercised with W = 32.)"""
    from mpc_rl_for_avs_amd import synth
    from conftest import converged
    inp = synth.solver_inputs(4096, 8, seed=0)
    # the slowest instances of the batch at max_iter 1000 (tools/tail_study.py) and a random handful
    sel = np.array([3424, 1037, 3543, 965, 1611, 2461, 1655, 3000] + list(range(0, 200, 8)))
    sub = {k: (v[sel] if v is not None else None) for k, v in inp.items() if k in ("state", "ego_index", "weights", "is_collide", "vref", "others")}
    kw = dict(collision_cost=True, max_iter=1000)
    free = oracle.solve_batch(ref_table, sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=sub["vref"],
                              others=sub["others"], xy_bounds=False, **kw)
    guard = oracle.solve_batch(ref_table, sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=sub["vref"],
                               others=sub["others"], xy_bounds=False, stall_window=32, **kw)
    got = cpu_wave(ref_table, sub, stall_window=32, **kw)
    assert 100 < free["iters"].max() <= 200 and guard["iters"].max() <= 64
    quick = free["iters"] <= 30
    assert np.array_equal(guard["status"][quick], free["status"][quick]) and np.array_equal(guard["iters"][quick], free["iters"][quick])
    assert (guard["status"][~quick] == 4).sum() >= 3
    assert (got["status"] == guard["status"]).mean() > 0.9 and np.array_equal(got["status"][quick], guard["status"][quick])
    assert (got["iters"] == guard["iters"]).mean() > 0.85
    both = converged(got["status"])
    assert np.abs(got["u0"] - guard["u0"])[both].max() < 1e-6


def test_row_cooperative_rollout_equals_the_oracle(cpu_wave, oracle, ref_table):
    """Round 5: the line search integrates trial t in the 16-lane row t with the lanes of a row as the components of the stage
    (mpc_wave.hpp: rollouts) and the linearised Newton step as a fifth trial in lanes 8..13 - one code path for every build
    (the split / fused variants of rounds 3 - 4 are gone).  Host model of the wave (row broadcasts modelled lane by lane) against
    the oracle: same statuses, same iteration counts but for instances that are chaotic in the last bit, same actions."""
    from mpc_rl_for_avs_amd import synth
    from conftest import converged, rel_u0_err
    for V, cc, seed in ((8, True, 5), (4, False, 6)):
        inp = synth.solver_inputs(192, V, seed=seed)
        a = cpu_wave(ref_table, inp, collision_cost=cc)
        want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                                  others=inp["others"], collision_cost=cc, max_iter=100, xy_bounds=False)
        assert np.array_equal(a["status"], want["status"])
        assert (a["iters"] == want["iters"]).mean() >= 0.98
        both = converged(a["status"]) & converged(want["status"])
        assert both.mean() > 0.95 and rel_u0_err(a["u0"], want["u0"])[both].max() < 1e-6


@pytest.mark.parametrize("N", [2, 16, 30, 40])
def test_collision_potential_over_lane_groups_for_every_horizon(cpu_wave, oracle, ref_table, N):
    """Round 5: the preparation phase deals the vehicles of the collision potential to up to three lane groups (lane =
    g (N - 1) + k - 1; three groups while 3 (N - 1) <= 64, two up to N = 33, one beyond) and adds the partial sums from words of
    the stage that are dead at that point - with every LDS access of the host build checked against lds_doubles().  Against
    the oracle, which sums vehicle by vehicle: same statuses and actions, same iteration counts but for chaotic instances."""
    from mpc_rl_for_avs_amd import synth
    from conftest import converged, rel_u0_err
    inp = synth.solver_inputs(160, 8, seed=3, N=N)
    a = cpu_wave(ref_table, inp, collision_cost=True, N=N)
    want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                              others=inp["others"], collision_cost=True, max_iter=100, xy_bounds=False, N=N)
    assert (a["status"] == want["status"]).mean() >= 0.99
    assert (a["iters"] == want["iters"]).mean() >= 0.96
    both = converged(a["status"]) & converged(want["status"])
    assert both.mean() > 0.9 and (rel_u0_err(a["u0"], want["u0"])[both] < 1e-6).mean() >= 0.99


def test_acceptable_level_termination(cpu_wave, oracle, ref_table):
    """IPOPT's acceptable-level termination with IPOPT's defaults (acceptable_tol 1e-6, acceptable_iter
    15) - a solve at tol 1e-8 whose scaled error has been below 1e-6 for 15 consecutive iterations ends with status 6 (7 with a
    wall multiplier) instead of dithering to the cap.  It fires on 0 - 1 instance of a config-3 batch; the kernel source on the
    host agrees with the oracle on them, the answers are the 1e-6 solve's to 1e-6, and the rule is off when tol >= 1e-6."""
    from mpc_rl_for_avs_amd import synth
    from conftest import rel_u0_err
    found = 0
    for seed in (5, 7):
        inp = synth.solver_inputs(4096, 8, seed=seed)
        kw = dict(vref=inp["vref"], others=inp["others"], collision_cost=True, xy_bounds=False)
        o8 = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], max_iter=100, tol=1e-8, **kw)
        idx = np.nonzero(o8["status"] >= 6)[0]
        assert idx.size <= 3 and (o8["iters"][idx] >= 16).all()
        if idx.size == 0:
            continue
        found += idx.size
        sub = {k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
        a = cpu_wave(ref_table, sub, collision_cost=True, max_iter=100)
        # (the same exit; WHEN the fifteen iterations are complete differs - these instances sit on their rounding floor, where the
        # error dithers differently in two implementations: 33 against 52 iterations on seed 5's)
        assert np.array_equal(a["status"], o8["status"][idx]) and (a["iters"] >= 16).all()
        assert rel_u0_err(a["u0"], o8["u0"][idx]).max() < 1e-5
        o6 = oracle.solve_batch(ref_table, sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=sub["vref"],
                                others=sub["others"], collision_cost=True, xy_bounds=False, max_iter=1000, tol=1e-6)
        assert (o6["status"] < 6).all() and (o6["iters"] <= o8["iters"][idx]).all()
        ok = (o6["status"] == 0) | (o6["status"] == 5)
        assert rel_u0_err(o8["u0"][idx], o6["u0"])[ok].max() < 1e-5
    assert found >= 1
