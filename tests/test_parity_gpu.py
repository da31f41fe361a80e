"""Parity tests proper: the HIP engine, called through the C ABI, against the CPU oracle on the same seeded
inputs, against the committed golden fixtures, and - at BASELINE's full sizes - through size-independent
properties.  Tolerance: north_star's 1e-4 relative on the returned action u0 (expected ~1e-9); a small fraction
of synthetic instances is ill-posed / multi-modal (ego beyond the end of the path table, vehicles inside each
other with the collision cost on) and may legitimately end in another local minimum, so the bar is stated on
the fraction of instances both solvers converge on."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, converged, rel_u0_err, unexplained_disagreements

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star: "matching reference controls to 1e-4 rel"


@pytest.fixture(scope="module")
def eng():
    from mpc_rl_for_avs_amd import engine
    e = engine.MPCEngine(horizon=20, max_iter=100)
    yield e
    e.close()


def _oracle(oracle, ref, inp, cc, N=20, **kw):
    return oracle.solve_batch(ref, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                              vref=inp.get("vref"), others=inp.get("others"), collision_cost=cc, max_iter=100,
                              xy_bounds=False, N=N, **kw)


def _gpu(eng, inp, cc):
    return eng.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp.get("vref"),
                           others=inp.get("others"), collision_cost=cc)


@pytest.mark.parametrize("B,V,cc,seed", [(1024, 4, False, 0), (1024, 8, True, 0), (333, 1, False, 4), (257, 9, True, 7)])
def test_engine_matches_oracle(eng, oracle, ref_table, B, V, cc, seed):
    from mpc_rl_for_avs_amd import synth
    inp = synth.solver_inputs(B, V, seed=seed)
    want = _oracle(oracle, ref_table, inp, cc)
    got = _gpu(eng, inp, cc)
    both = converged(got["status"]) & converged(want["status"])
    # measured (rounds 3 and 4, every build of the kernel): 99.2 - 100 % converge within 100 iterations, statuses equal
    # on >= 99.8 %, iteration counts equal on >= 99.6 %.  The action gate is EXACT: no instance beyond 1e-4 unless it is
    # proven to be two certified minimisers of an instance on which the oracle itself is last-bit chaotic
    # (conftest.unexplained_disagreements; round 4 measured 0 such instances in 18 432 on three seeds,
    # profiles/r04_bitcompare.txt)
    assert both.mean() >= 0.99
    assert (got["status"] == want["status"]).mean() >= 0.995
    err = rel_u0_err(got["u0"], want["u0"])[both]
    assert unexplained_disagreements(oracle, ref_table, inp, cc, got, want, TOL, max_iter=100) == []
    assert np.percentile(err, 99) < 1e-8
    assert (got["iters"] == want["iters"])[both].mean() > 0.99
    # full trajectories of the agreeing instances
    ok = both.copy()
    ok[both] = err <= TOL
    assert np.abs(got["X"] - want["X"])[ok].max() < 1e-3
    # dynamics are satisfied exactly by construction (single shooting)
    X, U = got["X"][ok], got["U"][ok]
    beta = np.arctan(0.5 * np.tan(U[:, :, 1]))
    nxt = X[:, :-1] + 0.1 * np.stack([X[:, :-1, 3] * np.cos(X[:, :-1, 2] + beta), X[:, :-1, 3] * np.sin(X[:, :-1, 2] + beta),
                                      X[:, :-1, 3] / 2.5 * np.sin(beta), U[:, :, 0]], axis=-1)
    assert np.abs(nxt - X[:, 1:]).max() < 1e-11
    assert np.all(np.abs(U[:, :, 0]) <= 5 + 1e-7) and np.all(np.abs(U[:, :, 1]) <= np.pi / 3 + 1e-7)
    assert np.all(X[:, :, 3] >= -1e-7) and np.all(np.abs(X[:, :, 2]) <= np.pi + 1e-7)


def test_long_horizons(oracle, ref_table):
    """Horizons beyond half a wave (one line-search trial per cost pass) up to the interface maximum."""
    from mpc_rl_for_avs_amd import engine, synth
    inp = synth.solver_inputs(200, 8, seed=13)
    for N in (30, 40, 64):
        e = engine.MPCEngine(horizon=N, max_iter=100)
        sub = dict(inp, vref=np.concatenate([inp["vref"], np.repeat(inp["vref"][:, -1:], N - 20, axis=1)], axis=1),
                   others=None)
        got = e.solve_batch(sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=sub["vref"])
        want = _oracle(oracle, ref_table, sub, False, N=N)
        both = (got["status"] == 0) & (want["status"] == 0)
        assert both.mean() > 0.75 and (rel_u0_err(got["u0"], want["u0"])[both] <= TOL).mean() > 0.99, N
        assert (got["status"] == want["status"]).mean() > 0.95, N
        e.close()
    with pytest.raises(engine.EngineError):
        engine.MPCEngine(horizon=65)


def test_golden_fixtures(eng):
    g = np.load(os.path.join(GOLDEN, "oracle_solutions.npz"))
    for name, cc in (("cfg2", False), ("cfg3", True)):
        inp = {k: g[f"{name}_{k}"] for k in ("state", "ego_index", "vref", "weights", "is_collide", "others")}
        got = _gpu(eng, inp, cc)
        assert np.array_equal(got["status"], g[f"{name}_status"]), name        # all 32, incl. the one status-5 instance
        ok = converged(g[f"{name}_status"])
        assert ok.sum() >= 31
        # the fixture is the oracle solving what the engine solves (xy_bounds=False: the never-active |x|, |y| <= 500 bounds
        # of agents/pure_mpc.py:272-274 carry no barrier terms, tests/test_oracle.py::test_xy_bounds_never_matter): every one
        # of the 32 instances per configuration to 1e-6
        err = rel_u0_err(got["u0"], g[f"{name}_u0"])
        assert err[ok].max() < 1e-6, (name, np.nonzero(ok & (err >= 1e-6))[0], err[ok].max())


def test_known_answers(eng):
    st = np.array([[2.0, 45.0, -np.pi / 2, 10.0]])
    base = dict(state=st, ego_index=np.array([4], np.int32), weights=np.ones((1, 3)), is_collide=np.zeros(1, np.uint8))
    out = _gpu(eng, base, False)                                  # on the reference at reference speed
    assert out["status"][0] == 0 and np.abs(out["U"]).max() < 1e-7
    out = _gpu(eng, dict(base, vref=np.full((1, 21), 0.7)), False)   # RL speed override 0.7 -> full braking
    assert out["status"][0] == 0 and abs(out["u0"][0, 0] + 5.0) < 1e-6 and abs(out["u0"][0, 1]) < 1e-6
    out = _gpu(eng, dict(base, state=np.array([[2.0, 45.0, -np.pi / 2, 31.0]])), False)
    assert out["status"][0] == 3                                  # state outside the bounds is flagged


def test_edge_shapes(eng, oracle, ref_table):
    from mpc_rl_for_avs_amd import engine, synth
    inp = synth.solver_inputs(130, 3, seed=12)
    full = _gpu(eng, inp, True)
    for B in (1, 63, 65, 130):                       # ragged batches: the same instance gives the same answer
        sub = {k: (v[:B] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
        got = _gpu(eng, sub, True)
        assert np.array_equal(got["u0"], full["u0"][:B]) and np.array_equal(got["iters"], full["iters"][:B])
    empty = {k: (v[:0] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    assert _gpu(eng, empty, True)["u0"].shape == (0, 2)
    # no vref -> speeds of the table; no others with the collision cost on
    a = _gpu(eng, dict(inp, vref=None, others=None), False)
    b = _oracle(oracle, ref_table, dict(inp, vref=None, others=None), False)
    both = (a["status"] == 0) & (b["status"] == 0)
    assert (rel_u0_err(a["u0"], b["u0"])[both] <= TOL).mean() > 0.99
    # other horizons (compile-time 16, runtime 9) and a maximal vehicle count
    for N in (16, 9):
        e = engine.MPCEngine(horizon=N, max_iter=100)
        sub = dict(inp, vref=inp["vref"][:, :N + 1])
        got = e.solve_batch(sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=sub["vref"])
        want = _oracle(oracle, ref_table, dict(sub, others=None), False, N=N)
        both = (got["status"] == 0) & (want["status"] == 0)
        assert got["U"].shape == (130, N, 2) and both.mean() > 0.9
        assert (rel_u0_err(got["u0"], want["u0"])[both] <= TOL).mean() > 0.99
        e.close()
    with pytest.raises(ValueError):
        eng.solve_batch(inp["state"][:, :3], inp["ego_index"], inp["weights"], inp["is_collide"])
    with pytest.raises(ValueError):
        eng.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"][:, :5])


def test_full_size_properties(eng):
    """BASELINE sizes (B = 4096, V = 8, collision cost on): permutation equivariance, batch-size independence,
    determinism, sane statuses, bounds."""
    from mpc_rl_for_avs_amd import synth
    inp = synth.solver_inputs(4096, 8, seed=0)
    a = _gpu(eng, inp, True)
    assert converged(a["status"]).mean() >= 0.99 and set(np.unique(a["status"])) <= {0, 1, 2, 4, 5}
    assert np.all(np.isfinite(a["u0"]))
    assert np.all(np.abs(a["u0"][:, 0]) <= 5 + 1e-7) and np.all(np.abs(a["u0"][:, 1]) <= np.pi / 3 + 1e-7)
    b = _gpu(eng, inp, True)
    assert np.array_equal(a["u0"], b["u0"])                                   # run-to-run determinism
    perm = np.random.default_rng(1).permutation(4096)
    p = {k: (v[perm] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    c = _gpu(eng, p, True)
    assert np.array_equal(c["u0"], a["u0"][perm]) and np.array_equal(c["iters"], a["iters"][perm])
    one = {k: (v[1234:1235] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    assert np.array_equal(_gpu(eng, one, True)["u0"][0], a["u0"][1234])       # B = 1 equals its row of B = 4096


def test_device_pointer_path_equals_host_path(eng):
    import torch
    from mpc_rl_for_avs_amd import synth
    inp = synth.solver_inputs(512, 8, seed=5)
    host = _gpu(eng, inp, True)
    dev = torch.device("cuda:0")
    t = lambda x, dt_: torch.as_tensor(np.ascontiguousarray(x), dtype=dt_, device=dev)
    out = eng.solve_batch_torch(t(inp["state"], torch.float64), t(inp["ego_index"], torch.int32),
                                t(inp["weights"], torch.float64), t(inp["is_collide"], torch.uint8),
                                vref=t(inp["vref"], torch.float64), others=t(inp["others"], torch.float64),
                                collision_cost=True)
    torch.cuda.synchronize()
    assert np.array_equal(out["u0"].cpu().numpy(), host["u0"])
    assert np.array_equal(out["status"].cpu().numpy(), host["status"])
    with pytest.raises(ValueError):
        eng.solve_batch_torch(t(inp["state"], torch.float32), t(inp["ego_index"], torch.int32),
                              t(inp["weights"], torch.float64), t(inp["is_collide"], torch.uint8))


def test_agent_api_end_to_end(oracle, ref_table):
    """PureMPC_Agent.predict / predict_batch on the GPU reproduce the oracle fed with the same problem data."""
    from mpc_rl_for_avs_amd import synth
    from mpc_rl_for_avs_amd.pure_mpc import PureMPC_Agent

    class Env:
        config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}
    cfg = dict(horizon=20, render=False, ttc_threshold=3, weight_speed=1, weight_control=1, weight_input_diff=1,
               speed_override=0)
    obs = synth.make_obs_batch(48, 4, seed=3)
    agent = PureMPC_Agent(Env(), cfg)
    act = agent.predict_batch(obs)
    assert act.shape == (48, 2)
    from host_preamble import HostPreambleAgent
    ref_agent = HostPreambleAgent(Env(), cfg, engine=agent._engine)
    egos, others = zip(*[ref_agent._vehicles_from_obs(o) for o in obs])
    while len(ref_agent._states) < 48:
        ref_agent._states.append(type(ref_agent._states[0])())
    for b in range(48):
        ref_agent._check_collision_env(ref_agent._states[b], egos[b], others[b])
    inp = ref_agent.build_solver_inputs(ref_agent._states, list(egos), list(others))
    want = _oracle(oracle, ref_table, inp, False)
    ok = (want["status"] == 0) & (agent.last_solve["status"] == 0)
    assert ok.mean() > 0.9 and (rel_u0_err(act, want["u0"])[ok] <= TOL).mean() > 0.97
    single = PureMPC_Agent(Env(), cfg)
    a0 = single.predict(obs[0])
    assert a0.shape == (2,) and np.array_equal(a0, act[0])
    single.reset_env_state()
    assert single.predict(obs[0], return_numpy=False).steer == single.last_solve["act"][0, 1]


def test_limits_of_the_interface(oracle):
    """MPC_MAX_OTHERS vehicles, the shortest horizons, a caller-supplied reference path of another length."""
    from mpc_rl_for_avs_amd import engine, synth
    from mpc_rl_for_avs_amd.reference_path import reference_states
    ref = reference_states()
    inp = synth.solver_inputs(96, 8, seed=17)
    # 16 vehicles: the 8 synthetic ones and 8 copies shifted sideways
    far = inp["others"].copy()
    far[:, :, 0] += 7.0
    far[:, :, 1] -= 9.0
    oth16 = np.concatenate([inp["others"], far], axis=1)
    e = engine.MPCEngine(horizon=20, max_iter=100)
    got = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"], others=oth16,
                        collision_cost=True)
    want = oracle.solve_batch(ref, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                              others=oth16, collision_cost=True, max_iter=100, xy_bounds=False)
    both = (got["status"] == 0) & (want["status"] == 0)
    assert both.mean() > 0.8 and (rel_u0_err(got["u0"], want["u0"])[both] <= TOL).mean() > 0.99
    with pytest.raises(engine.EngineError):
        e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                      others=np.concatenate([oth16, far[:, :1]], axis=1), collision_cost=True)      # 17 vehicles
    e.close()
    # the largest workspace the interface allows: horizon 64 with 16 vehicles in the collision cost (41 KB of LDS): 65 slots per
    # node + the 12 words of the PQ / PB table of the rollout, the table of constants (3 + 1, 12 trig, 10 log, 8 bounds, 6 solve
    # constants, 3 no-bound), 4 per vehicle; the BASELINE shape (horizon 20, 8 vehicles) is 13.5 KB, i.e. 12 instances per CU -
    # the same layout in every build since round 5 (rounds 2 - 4: 9.9 / 11.9 KB)
    e = engine.MPCEngine(horizon=64, max_iter=100)
    assert e.workspace_bytes(1, 16) == (77 * 65 + 43 + 64) * 8
    e20 = engine.MPCEngine(horizon=20, max_iter=100)
    assert e20.workspace_bytes(65536, 8) == e20.workspace_bytes(1, 8) == (77 * 21 + 43 + 32) * 8 <= 163840 // 12
    assert e20.workspace_bytes(4096, 0) == e20.workspace_bytes(1024, 0) == (65 * 21 + 43) * 8
    e20.close()
    sub = {k: (v[:24] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    vref64 = np.concatenate([sub["vref"], np.repeat(sub["vref"][:, -1:], 44, axis=1)], axis=1)
    got = e.solve_batch(sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=vref64, others=oth16[:24],
                        collision_cost=True)
    want = oracle.solve_batch(ref, sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=vref64,
                              others=oth16[:24], collision_cost=True, N=64, max_iter=100, xy_bounds=False)
    both = (got["status"] == 0) & (want["status"] == 0)
    assert both.sum() >= 8 and (rel_u0_err(got["u0"], want["u0"])[both] <= TOL).all()
    assert (got["status"] == want["status"]).mean() > 0.8
    e.close()
    for N in (1, 2, 3):
        e = engine.MPCEngine(horizon=N, max_iter=100)
        got = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"][:, :N + 1])
        want = oracle.solve_batch(ref, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                                  vref=inp["vref"][:, :N + 1], N=N, max_iter=100, xy_bounds=False)
        both = (got["status"] == 0) & (want["status"] == 0)
        assert both.mean() > 0.9 and (rel_u0_err(got["u0"], want["u0"])[both] <= TOL).mean() > 0.99, N
        assert got["U"].shape == (96, N, 2) and got["X"].shape == (96, N + 1, 4)
        e.close()
    # a straight 30-point path heading east at 6 m/s
    M = 30
    path = np.stack([np.arange(M) * 1.0, np.full(M, 3.0), np.full(M, 6.0), np.zeros(M)], axis=1)
    rng = np.random.default_rng(3)
    B = 64
    idx = rng.integers(0, M - 2, B).astype(np.int32)
    state = np.stack([path[idx, 0] + rng.uniform(-0.3, 0.3, B), 3.0 + rng.uniform(-0.4, 0.4, B),
                      rng.uniform(-0.1, 0.1, B), rng.uniform(0.0, 9.0, B)], axis=1)
    w = np.ones((B, 3))
    coll = np.zeros(B, np.uint8)
    e = engine.MPCEngine(horizon=20, max_iter=100, ref_table=path)
    got = e.solve_batch(state, idx, w, coll)
    want = oracle.solve_batch(path, state, idx, w, coll, max_iter=100, xy_bounds=False)
    both = (got["status"] == 0) & (want["status"] == 0)
    assert (got["status"] == want["status"]).mean() > 0.97            # egos near the end of the path do not converge
    assert both.mean() > 0.8 and (rel_u0_err(got["u0"], want["u0"])[both] <= TOL).mean() > 0.99
    e.close()


def _certified_full_batch(eng, oracle, ref_table, B, V, cc):
    """One BASELINE configuration at its full batch size: the engine against the oracle instance by instance, and -
    independently of any solver - a KKT certificate of the reference NLP for EVERY solution the engine calls converged
    (oracle/kkt_batch.py: multipliers re-fitted from the primal point alone)."""
    import kkt_batch as kb
    import nlp_batch as nb
    from mpc_rl_for_avs_amd import synth
    inp = synth.solver_inputs(B, V, seed=0)
    got = _gpu(eng, inp, cc)
    want = _oracle(oracle, ref_table, inp, cc)
    conv = converged(got["status"])
    assert conv.mean() >= 0.99, np.bincount(got["status"], minlength=6)
    assert got["iters"].max() <= 100 and np.percentile(got["iters"], 99) <= 60
    # an exact count, not a fraction (a dropped or doubled instance of the launch order must not hide in a tolerance; measured: 0):
    # at most three instances whose last-bit-chaotic end differs between device and oracle
    assert int((got["status"] != want["status"]).sum()) <= 3, np.nonzero(got["status"] != want["status"])[0]
    both = conv & converged(want["status"])
    err = rel_u0_err(got["u0"], want["u0"])[both]
    # measured: 0 of 4081 beyond 1e-4 (worst 2.1e-9); exact gate, see test_engine_matches_oracle
    assert unexplained_disagreements(oracle, ref_table, inp, cc, got, want, TOL, max_iter=100) == []
    assert np.percentile(err, 99) < 1e-8 and np.percentile(err, 99.9) < 1e-6
    assert (got["iters"] == want["iters"])[both].mean() > 0.99
    p = nb.Batch.build(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                       others=inp["others"], collision_cost=cc)
    sel = np.nonzero(conv)[0]
    # stationarity relative to max(1, |grad f|_inf) with non-negative multipliers complementary to 1e-8 in the units of
    # IPOPT's criterion (the objective scaled by sf = 100 / |grad f(start)|_inf, computed here from the NLP data alone),
    # dynamics to rounding, no bound violated
    sf = kb.objective_scale(p.take(sel))
    # (an instance that ended at IPOPT's acceptable level - status 6 / 7, a handful per batch - is certified at that level)
    tol_i = np.where(got["status"][sel] >= 6, 1e-6, 1e-8)
    assert (got["status"][sel] >= 6).sum() <= 8
    cert = kb.certify(p.take(sel), got["X"][sel], got["U"][sel], eps_c=tol_i / sf, sf=sf)
    # Exact gate (round 6): EVERY solution the engine calls converged certifies at its own tolerance, both relative to
    # max(1, |grad f|_inf) and in IPOPT's own units (residual of the scaled problem / s_d, Waechter & Biegler eq. (5), (6)).
    # Round 5 had allowed "<= 2 instances over, all within 10 x" here: that was the certifier's least-squares routine stopping
    # early on instance 2791 (BVLS at 1.4e-8 where the optimum is 2e-16), not the solver - kkt_batch now cross-checks BVLS with
    # two other exact methods.  Measured on the oracle's answers: worst 4e-11 (own units), 0.27 tol (IPOPT's units).
    assert (cert["stationarity"] <= tol_i).all(), (cert["stationarity"].max(), sel[cert["stationarity"].argmax()])
    assert (cert["stationarity_ipopt"] <= tol_i).all(), (cert["stationarity_ipopt"].max(), sel[cert["stationarity_ipopt"].argmax()])
    assert cert["feasibility"].max() <= 1e-10
    assert cert["bound_violation"].max() == 0.0
    # SURVEY section 8(c) pin (1) literally - complementarity 1e-8 in UNSCALED units: holds for all but a handful, and
    # every exception is an instance whose objective the scaling shrinks (measured: 3 of 4083, each with sf = 0.01)
    plain = kb.certify(p.take(sel), got["X"][sel], got["U"][sel], eps_c=tol_i)
    miss = plain["stationarity"] > tol_i
    assert miss.sum() <= 8 and (sf[miss] < 1.0).all(), (int(miss.sum()), sf[miss])
    # status 5 = a wall constraint carries a multiplier: nearly all of them hold a vehicle within 1e-6 of d^2 = 1 (the rest
    # have the multiplier large enough for the flag with a slack of mu / z just above that)
    if cc:
        assert (cert["n_wall"][got["status"][sel] == 5] >= 1).mean() > 0.8
    # and the certifier does tell: of the iterates the engine does NOT call converged (iteration cap) many fail it - some
    # are KKT points to the certificate's tolerances that miss the solver's stricter complementarity
    rest = np.nonzero(~conv)[0]
    if rest.size >= 4:
        bad = kb.certify(p.take(rest), got["X"][rest], got["U"][rest])
        assert (bad["stationarity"] > 1e-8).mean() > 0.4
    return dict(n=int(sel.size), stat=float(cert["stationarity"].max()))


def test_config2_full_batch_certified(eng, oracle, ref_table):
    """BASELINE config 2: B = 1024, horizon 20, 4 other vehicles, live objective."""
    r = _certified_full_batch(eng, oracle, ref_table, 1024, 4, False)
    assert r["n"] >= 1014


def test_config3_full_batch_certified(eng, oracle, ref_table):
    """BASELINE config 3 (the headline): B = 4096, horizon 20, 8 other vehicles, collision cost on."""
    r = _certified_full_batch(eng, oracle, ref_table, 4096, 8, True)
    assert r["n"] >= 4055


def test_independent_solver_fixtures(eng, ref_table):
    """tests/golden/independent_solutions.npz: the same instances solved by oracle/ipopt_restated.py, a dense
    full-space restatement of IPOPT's algorithm that shares no solver code with the engine or with mpc_oracle.c
    (generator: tests/golden/make_independent.py).  The NLP is non-convex: where the two solvers end in different
    minima BOTH points must be certified KKT points, and that may happen on a few per cent of the instances only."""
    import kkt_batch as kb
    import nlp_batch as nb
    from mpc_rl_for_avs_amd import synth
    g = np.load(os.path.join(GOLDEN, "independent_solutions.npz"))
    for name, V, cc in (("c2", 4, False), ("c3", 8, True)):
        n = g[f"{name}_u0"].shape[0]
        assert n >= 128
        inp = synth.solver_inputs(n, V, seed=0)
        got = _gpu(eng, inp, cc)
        both = (g[f"{name}_status"] == 0) & (got["status"] == 0)
        assert both.sum() >= 0.8 * n
        err = rel_u0_err(got["u0"], g[f"{name}_u0"])
        agree = both & (err <= TOL)
        assert agree.sum() >= 0.93 * both.sum(), (name, agree.sum(), both.sum())
        assert np.median(err[agree]) < 1e-9
        other = np.nonzero(both & ~agree)[0]
        if other.size:
            p = nb.Batch.build(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                               vref=inp["vref"], others=inp["others"], collision_cost=cc).take(other)
            mine = kb.certify(p, got["X"][other], got["U"][other])
            theirs = kb.certify(p, g[f"{name}_X"][other], g[f"{name}_U"][other])
            assert mine["stationarity"].max() <= 1e-8 and mine["feasibility"].max() <= 1e-10
            assert theirs["stationarity"].max() <= 1e-6       # the fixture's own points are KKT points too


def test_both_builds_of_the_solve_kernel_agree(oracle, ref_table):
    """The engine launches one of two builds of the same solver source by how deep the batch fills the SIMDs (the latency
    build - 218 registers, two resident waves per SIMD, everything hoisted - up to a batch of 4096 on 256 CUs; the
    128-register build beyond and with MPC_FLAG_THROUGHPUT).  Same instances through both: statuses equal, actions equal to
    1e-6 (the builds fuse loops differently and may contract a multiply-add differently; an instance whose iterates are
    chaotic in the last bit may then take another path to the same point)."""
    import torch
    from mpc_rl_for_avs_amd import engine, synth
    dev = torch.device("cuda", 0)
    inp = synth.solver_inputs(4096, 8, seed=3)
    e = engine.MPCEngine(horizon=20, max_iter=100)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)

    def run(n, throughput):
        args = dict(state=t(inp["state"][:n], torch.float64), ego_index=t(inp["ego_index"][:n], torch.int32),
                    weights=t(inp["weights"][:n], torch.float64), is_collide=t(inp["is_collide"][:n], torch.uint8),
                    vref=t(inp["vref"][:n], torch.float64), others=t(inp["others"][:n], torch.float64), collision_cost=True)
        o = e.solve_batch_torch(**args, sync=True, throughput=throughput)
        return {k: v.cpu().numpy() for k, v in o.items()}
    lat1 = run(1024, False)          # every wave alone on its SIMD
    lat4 = run(4096, False)          # the same build, four waves per SIMD of batch depth
    bulk = run(4096, True)           # 128-register build
    assert np.array_equal(lat1["status"], lat4["status"][:1024]) and np.array_equal(lat1["u0"], lat4["u0"][:1024])
    for a, b, n in ((lat4, bulk, 4096),):
        same = (a["status"][:n] == b["status"][:n])
        assert same.mean() >= 0.998
        ok = same & converged(a["status"][:n])
        err = rel_u0_err(a["u0"][:n], b["u0"][:n])[ok]
        assert (err <= 1e-6).mean() >= 0.999 and np.median(err) < 1e-12
    want = _oracle(oracle, ref_table, {k: v[:1024] for k, v in inp.items() if isinstance(v, np.ndarray)}, True)
    both = converged(lat1["status"]) & converged(want["status"])
    assert (rel_u0_err(lat1["u0"], want["u0"])[both] <= TOL).all()
    e.close()


def test_stall_window_on_the_gpu(oracle, ref_table):
    """mpc_config.stall_window = 64 with the reference's max_iter 1000: the batch no longer waits 40 ms for the instances
    that never converge, the instances that converge quickly are untouched, and the engine agrees with the oracle run under
    the same rule."""
    from mpc_rl_for_avs_amd import engine, synth
    inp = synth.solver_inputs(4096, 8, seed=0)
    e = engine.MPCEngine(horizon=20, max_iter=1000, stall_window=64)
    got = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                        others=inp["others"], collision_cost=True)
    want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                              others=inp["others"], collision_cost=True, max_iter=1000, xy_bounds=False, stall_window=64)
    assert got["iters"].max() < 300 and (got["status"] == 1).sum() == 0
    assert (got["status"] == want["status"]).mean() >= 0.998 and converged(got["status"]).mean() >= 0.996
    assert 3 <= (got["status"] == 4).sum() <= 30
    both = converged(got["status"]) & converged(want["status"])
    assert unexplained_disagreements(oracle, ref_table, inp, True, got, want, TOL, max_iter=1000, stall_window=64) == []
    e.close()


def test_closed_loop_fixtures_vs_independent_solver(oracle, ref_table):
    """tests/golden/closed_loop_ipopt.npz: problem data recorded from closed-loop runs (BASELINE config 1 and the
    config-4 rollout, collision cost off / on, RL speed override / none, and the v1 input domain: cost weights from
    [-1, 1]^3; generator tests/golden/make_closed_loop.py) solved by oracle/ipopt_restated.py - IPOPT's algorithm incl. its
    restoration phase - at the REFERENCE's settings (tol 1e-6, max_iter 1000, agents/pure_mpc.py:294-295).
    The engine at max_iter 1000 must reproduce, instance by instance, the classification the CPU analysis made with the
    same algorithm (profiles/r04_parity_vs_ipopt.txt; tests/test_oracle.py pins its counts): the same statuses, the same
    actions to 1e-6 where converged, hence the same set of instances within 1e-4 of the proxy; and every converged
    answer that differs from the proxy's must be a certified KKT point (another local minimiser of a non-convex NLP)."""
    import kkt_batch as kb
    import nlp_batch as nb
    from mpc_rl_for_avs_amd import engine
    from test_oracle import CLOSED_LOOP_COUNTS
    g = np.load(os.path.join(GOLDEN, "closed_loop_ipopt.npz"))
    e = engine.MPCEngine(horizon=20, max_iter=1000)
    tot_both = tot_agree = 0
    for name, (n_proxy, n_eng, n_both, n_agree) in CLOSED_LOOP_COUNTS.items():
        cc = name.endswith("cc")
        d = {k: g[f"{name}_{k}"] for k in ("state", "ego_index", "vref", "weights", "is_collide", "others")}
        got = e.solve_batch(d["state"], d["ego_index"], d["weights"], d["is_collide"], vref=d["vref"], others=d["others"],
                            collision_cost=cc)
        assert (got["status"] == 2).sum() == 0, name      # the slack floor (kMinSlack) removed the NaN sweeps
        # the device against the CPU run of the same algorithm: converged on the same instances, same actions there
        ok = converged(got["status"])
        assert np.array_equal(ok, converged(g[f"{name}_oracle_status"])), (name, np.bincount(got["status"], minlength=6))
        assert int(ok.sum()) == n_eng
        dev_err = rel_u0_err(got["u0"], g[f"{name}_oracle_u0"])
        split = np.zeros(0, dtype=np.int64)
        if name != "c4v1":
            assert dev_err[ok].max() < 1e-6, name
        else:
            # negative cost weights: bang-bang steering profiles with many local minimisers, and an iteration that is chaotic
            # in the rounding (the device contracts multiply-adds, the CPU build does not) - an instance may end in another
            # minimiser on the device than in the CPU run.  Round 6: no bare count any more - each such instance goes through
            # the exact gate of the engine-vs-oracle tests (conftest.unexplained_disagreements): BOTH answers must carry a KKT
            # certificate AND the oracle itself must jump by more than the tolerance when the ego state moves by 1 - 2 ulp.
            split = np.nonzero(ok & ~(dev_err < 1e-6))[0]
            if split.size:
                want = oracle.solve_batch(ref_table, d["state"], d["ego_index"], d["weights"], d["is_collide"], vref=d["vref"],
                                          others=d["others"], collision_cost=cc, max_iter=1000, xy_bounds=False)
                assert np.array_equal(want["status"], g[f"{name}_oracle_status"]), name     # the fixture's own generator call
                assert unexplained_disagreements(oracle, ref_table, d, cc, got, want, 1e-6, max_iter=1000) == [], (name, split)
        both = (g[f"{name}_status"] == 0) & ok
        err = rel_u0_err(got["u0"], g[f"{name}_u0"])
        agree = both & (err <= TOL)
        assert int(both.sum()) == n_both and abs(int(agree.sum()) - n_agree) <= split.size, (name, both.sum(), agree.sum())
        assert np.median(err[agree]) < 1e-6          # the proxy stops at tol 1e-6
        other = np.nonzero(both & ~agree)[0]
        if other.size:
            p = nb.Batch.build(ref_table, d["state"], d["ego_index"], d["weights"], d["is_collide"], vref=d["vref"],
                               others=d["others"], collision_cost=cc).take(other)
            mine = kb.certify(p, got["X"][other], got["U"][other])
            assert mine["stationarity"].max() <= 1e-8 and mine["feasibility"].max() <= 1e-10, name
        if name != "c4v1":
            tot_both += int(both.sum())
            tot_agree += int(agree.sum())
    assert tot_agree / tot_both >= 0.97              # scenarios with non-negative weights: 747 of 766
    e.close()
    # the two c1 instances the engine's algorithm does not finish at tol 1e-8 while the proxy converges at its 1e-6: at the
    # reference's own tolerance the engine converges on them and returns the proxy's action.  (The one c4v1 instance - input-
    # difference weight -0.92, a bang-bang steering zig-zag with many local minimisers - ends with status 4 at either tolerance;
    # round 4's was another instance of the same kind.)
    e6 = engine.MPCEngine(horizon=20, max_iter=1000, tol=1e-6)
    n_hard = 0
    for name in ("c1",):
        hard = np.nonzero((g[f"{name}_status"] == 0) & ~converged(g[f"{name}_oracle_status"]))[0]
        d = {k: g[f"{name}_{k}"][hard] for k in ("state", "ego_index", "vref", "weights", "is_collide", "others")}
        got = e6.solve_batch(d["state"], d["ego_index"], d["weights"], d["is_collide"], vref=d["vref"], others=d["others"])
        assert converged(got["status"]).all() and got["iters"].max() <= 40
        assert rel_u0_err(got["u0"], g[f"{name}_u0"][hard]).max() <= TOL      # both stop at tol 1e-6: 2.6e-5 at most
        n_hard += hard.size
    assert n_hard == 2
    e6.close()


def test_config4_rollout_256_envs_against_oracle(oracle, ref_table):
    """BASELINE config 4: 256 parallel intersection environments, MPC in the loop (v0: the policy's action is the
    reference speed).  Every step's MPC actions against the oracle fed with the problem data the device preamble
    derived from that step's observations."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout
    dev = torch.device("cuda", 0)
    B, T = 256, 10
    e = engine.MPCEngine(horizon=20, max_iter=100)
    env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=5, n_others=4)
    pol = rollout.ActorCritic(1).to(dev)
    col = rollout.BatchedCollector(env, pol, e, version="v0", algorithm="ppo", n_steps=T, seed=3)
    col._begin_rollout()
    n_all = n_conv = 0
    for t in range(T):
        col._rollout_step()
        torch.cuda.synchronize()
        inp = e.last_inputs(B, 10)
        act = col.last_mpc["act"].cpu().numpy()
        st = col.last_mpc["status"].cpu().numpy()
        want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], np.ones((B, 3)), inp["is_collide"],
                                  vref=inp["vref"], max_iter=100, xy_bounds=False)
        assert (st == want["status"]).mean() >= 0.99, t
        both = converged(st) & converged(want["status"])
        assert (rel_u0_err(act, want["u0"])[both] <= TOL).mean() >= 0.995, t
        n_all += B
        n_conv += int(converged(st).sum())
    assert n_conv >= 0.97 * n_all
    e.close()


def test_heading_on_the_bound_and_config1_closed_loop(eng, oracle, ref_table):
    """The exit straight of the intersection: the observed heading is -pi to float32 rounding, 5.6e-8 outside the relaxed
    bound of the NLP.  The reference's IPOPT solves there (violation below its tolerance, start pushed inside); the
    engine used to answer status 3 and the action (0, 0) for the rest of the episode (round 1 and the first half of
    round 2: 17 % of the steps of BASELINE config 1)."""
    import sys
    from test_core_cpu import heading_on_the_bound_inputs
    inp = heading_on_the_bound_inputs(ref_table)
    want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], max_iter=100,
                              xy_bounds=False)
    got = eng.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"])
    assert np.array_equal(got["status"], want["status"]) and (got["status"][:6] == 0).all()
    assert rel_u0_err(got["u0"], want["u0"])[want["status"] == 0].max() <= TOL
    # BASELINE config 1: one ego, one other vehicle, the reference's stand-alone loop on the synthetic intersection
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import run_pure_mpc
    outcome, log, _ = run_pure_mpc.run(steps=150, verbose=False)
    status = np.array([r[6] for r in log])
    assert outcome == "arrived" and (status != 3).all() and converged(status).mean() >= 0.95


@pytest.mark.parametrize("B", [1025, 2047, 4096, 4097, 6000, 8192, 8193, 20001])
def test_launch_order_is_a_permutation(oracle, ref_table, B):
    """mpc_order_kernel (mpc_engine.hip) hands the workgroups their instances through a stable three-tier partition whenever a
    batch puts more than one wave on a SIMD (up to 64 per SIMD; since round 6 also the bulk launches of the throughput build,
    B > 8192).  Property: EVERY instance is solved exactly once, whatever the batch size
    does to the chunked prefix sums (B not a multiple of 1024), with non-finite states (NaN / inf compare false in every tier
    test) among the inputs: status and iteration arrays are prefilled with a sentinel and none may survive; the instances'
    results equal those of the same instances solved in a batch too small to be reordered."""
    import torch
    from mpc_rl_for_avs_amd import engine, synth
    inp = synth.solver_inputs(B, 8, seed=3)
    state = inp["state"].copy()
    bad = np.arange(7, B, 97)
    state[bad[0::3], 3] = np.nan           # NaN speed
    state[bad[1::3], 0] = np.inf           # infinite position
    state[bad[2::3], 2] = -np.inf
    dev = torch.device("cuda", 0)
    t = lambda x, dt_: torch.as_tensor(np.ascontiguousarray(x), dtype=dt_, device=dev)
    args = dict(state=t(state, torch.float64), ego_index=t(inp["ego_index"], torch.int32), weights=t(inp["weights"], torch.float64),
                is_collide=t(inp["is_collide"], torch.uint8), vref=t(inp["vref"], torch.float64), others=t(inp["others"], torch.float64),
                collision_cost=True)
    e = engine.MPCEngine(horizon=20, max_iter=30)
    out = dict(u0=torch.full((B, 2), 1e30, dtype=torch.float64, device=dev),
               status=torch.full((B,), -77, dtype=torch.int32, device=dev), iters=torch.full((B,), -77, dtype=torch.int32, device=dev))
    e.solve_batch_torch(**args, out=out, sync=True)
    st, it, u0 = out["status"].cpu().numpy(), out["iters"].cpu().numpy(), out["u0"].cpu().numpy()
    assert (st != -77).all() and (it != -77).all() and (u0 != 1e30).all()
    ok = np.ones(B, bool)
    ok[bad] = False
    assert np.isin(st[bad], (1, 2, 3, 4)).all()                    # a non-finite state ends as "not solved", never hangs
    # the same instances in batches of 1000 (not reordered: at most one wave per SIMD)
    for lo in (0, B - 1000):
        sub = {k: (v[lo:lo + 1000] if torch.is_tensor(v) else v) for k, v in args.items()}
        sub = {k: (v.contiguous() if torch.is_tensor(v) else v) for k, v in sub.items()}
        o2 = e.solve_batch_torch(**sub, sync=True)
        assert np.array_equal(o2["status"].cpu().numpy(), st[lo:lo + 1000]) and np.array_equal(o2["iters"].cpu().numpy(), it[lo:lo + 1000])
        assert np.array_equal(o2["u0"].cpu().numpy()[ok[lo:lo + 1000]], u0[lo:lo + 1000][ok[lo:lo + 1000]])
    e.close()


def test_strict_discontinuity_flag(oracle, ref_table):
    """MPC_FLAG_STRICT_DISCONTINUITY: the same iterates, but an instance that ends on the d = 1 discontinuity of the collision
    cost (status 5, or 7 at IPOPT's acceptable level) is reported as not solved (status 8) with its last iterate - what the
    reference's IPOPT reports for such a point (agents/pure_mpc.py:303-305: success false, the last iterate is used)."""
    from mpc_rl_for_avs_amd import engine, synth
    inp = synth.solver_inputs(2048, 8, seed=0)
    e = engine.MPCEngine(horizon=20, max_iter=100)
    kw = dict(vref=inp["vref"], others=inp["others"], collision_cost=True)
    a = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], **kw)
    b = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], strict_discontinuity=True, **kw)
    kink = (a["status"] == 5) | (a["status"] == 7)
    assert 20 < kink.sum() < 200
    assert (b["status"][kink] == 8).all() and np.array_equal(b["status"][~kink], a["status"][~kink])
    assert np.array_equal(a["u0"], b["u0"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["U"], b["U"])
    assert not engine.converged(b["status"][kink]).any()
    # without the collision cost there is no discontinuity and the flag changes nothing
    c = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"], strict_discontinuity=True)
    assert not (c["status"] == 8).any()
    e.close()
