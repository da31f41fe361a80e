"""Observation-level path on the GPU: `mpc_predict_batch` (device preamble + solve, per-environment detector memory
inside the engine) against the host mirror of the reference's preamble (pinned by the reference's own numpy outputs,
tests/test_host.py / test_preamble_cpu.py) followed by the same engine's `mpc_solve_batch`, over multi-step episodes
with resets.  Problem data must agree exactly (indices, flags, float32-derived values), actions to the 1e-4 of the
north star (expected ~1e-9: the host path pads absent vehicles far away instead of dropping them)."""
import os

import numpy as np
import pytest

from conftest import rel_u0_err
from test_host import CFG, Env

pytestmark = pytest.mark.gpu


def _episode_obs(B, V, t, rng):
    from mpc_rl_for_avs_amd import synth
    obs = synth.make_obs_batch(B, V, seed=500 * V + t)
    if V >= 3:
        drop = rng.uniform(size=B) < 0.3
        obs[drop, 2:, 0] = 0
    return obs


@pytest.mark.parametrize("V,cc", [(4, False), (9, True)])
def test_predict_batch_matches_host_preamble_plus_solve(V, cc):
    from mpc_rl_for_avs_amd.pure_mpc import PureMPC_Agent
    from host_preamble import HostPreambleAgent
    B, T = 200, 13
    dev = PureMPC_Agent(Env(), dict(CFG), collision_cost=cc)
    host = HostPreambleAgent(Env(), dict(CFG), collision_cost=cc, engine=dev._engine)
    rng = np.random.default_rng(V)
    fired = 0
    for t in range(T):
        obs = _episode_obs(B, V, t, rng)
        w = rng.uniform(0, 1, (B, 3)) if t % 3 == 1 else None
        rs = rng.uniform(0, 35, (B, 1)) if t == 8 else None
        act = dev.predict_batch(obs, w, rs)
        st = dev.last_solve["status"].copy()
        got = dev._engine.last_inputs(B, 10)
        env = dev.batch_env_state(B)
        want_act = host.predict_batch_host(obs, w, rs)
        want = host.last_inputs
        assert np.array_equal(got["state"], want["state"])
        assert np.array_equal(got["ego_index"], want["ego_index"])
        assert np.array_equal(got["is_collide"], want["is_collide"])
        assert np.array_equal(got["vref"], want["vref"])
        assert np.array_equal(env["is_collide"], want["is_collide"])
        assert np.array_equal(env["collision_memory"], [s.collision_memory for s in host._states[:B]])
        for b in range(B):
            ci = [-1 if c is None else c for c in host._states[b].conflict_index]
            assert list(env["conflict_index"][b, :len(ci)]) == ci
            for j, pt in enumerate(host._states[b].conflict_points):
                if pt is not None:
                    assert np.allclose(env["conflict_points"][b, j], pt, rtol=0, atol=1e-9)
        both = (st == 0) & (host.last_solve["status"] == 0)
        assert both.mean() > 0.8
        assert (rel_u0_err(act, want_act)[both] <= 1e-4).mean() > 0.995
        assert np.percentile(rel_u0_err(act, want_act)[both], 90) < 1e-7
        fired += int(want["is_collide"].sum())
        if t == 5:
            ids = np.where(rng.uniform(size=B) < 0.4)[0]
            dev.reset_env_state(ids)              # the engine's records; the mirror keeps its own states
            for i in ids:
                host._states[i] = type(host._states[i])()
    assert fired > B


def test_predict_batch_torch_zero_copy_and_mask_reset():
    import torch
    from mpc_rl_for_avs_amd import engine, synth
    B = 128
    e = engine.MPCEngine(horizon=20, max_iter=100)
    obs = synth.make_obs_batch(B, 5, seed=3)
    w = np.ones((B, 3))
    first = e.predict_batch(obs, w)
    mem0 = e.env_state(B)["collision_memory"].copy()
    assert mem0.max() == 10
    # second step through torch tensors on a side stream
    dev = torch.device("cuda", 0)
    t_obs = torch.as_tensor(obs, device=dev)
    t_w = torch.as_tensor(w, device=dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        out = e.predict_batch_torch(t_obs, t_w)
        done = torch.zeros(B, dtype=torch.uint8, device=dev)
        done[::2] = 1
        e.reset_env_mask_torch(done)
    s.synchronize()
    mem1 = e.env_state(B)["collision_memory"]
    assert np.all(mem1[::2] == 0)
    assert np.array_equal(mem1[1::2], np.where(mem0[1::2] > 0, mem0[1::2] - 1, 0))
    # inside the memory window the detector replays its memory: same problem data except the stop profile's start
    both = (first["status"] == 0) & (out["status"].cpu().numpy() == 0)
    assert (rel_u0_err(out["act"].cpu().numpy(), first["act"])[both] <= 1e-9).mean() > 0.95
    with pytest.raises(ValueError):
        e.predict_batch_torch(t_obs.double(), t_w)
    e.close()


def test_predict_batch_argument_errors():
    from mpc_rl_for_avs_amd import engine
    e = engine.MPCEngine(horizon=20)
    with pytest.raises(ValueError):
        e.predict_batch(np.zeros((4, 10, 7), np.float32), np.ones((4, 3)))
    with pytest.raises(engine.EngineError):
        e.predict_batch(np.zeros((4, 18, 8), np.float32), np.ones((4, 3)))       # vehicles_count > 17
    assert e.predict_batch(np.zeros((0, 10, 8), np.float32), np.ones((0, 3)))["act"].shape == (0, 2)
    e.close()


def test_collector_on_device_equals_single_instance_path():
    """Config-4 style rollout (policy -> MPC -> env on the GPU): every step's MPC actions equal what looping the
    reference's single-environment call sequence over the environments gives for the same observations."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout
    from host_preamble import HostPreambleAgent as PureMPC_Agent
    dev = torch.device("cuda", 0)
    B, T = 48, 12
    eng = engine.MPCEngine(horizon=20, max_iter=100)
    env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=11, n_others=4)
    pol = rollout.ActorCritic(3).to(dev)
    col = rollout.BatchedCollector(env, pol, eng, version="v1", algorithm="ppo", n_steps=T, seed=2)
    stats = col.collect_rollouts()
    assert stats["steps"] == B * T
    buf = col.buffer
    obs = buf.obs.cpu().numpy()
    w = torch.clamp(buf.actions, -1, 1)[:, :, :3].double().cpu().numpy()
    got = buf.mpc_actions.cpu().numpy()
    agents = [PureMPC_Agent(Env(), dict(CFG), engine=eng) for _ in range(B)]      # one reference-style agent per env
    worst = 0.0
    n_ok = n_all = 0
    for t in range(T):
        for b in range(0, B, 4):                                                    # a quarter of the envs, all steps
            a = agents[b].predict(obs[t, b], weights_from_RL=w[t, b][None])
            # (the policy is untrained: its clipped weights are often negative, many of these problems are nonconvex
            # and end at the iteration cap - both paths run the same iterations on the same data either way)
            e = float(rel_u0_err(got[t, b][None], a[None])[0])
            worst = max(worst, e)
            n_ok += e <= 1e-4
            n_all += 1
    assert n_all == T * B // 4 and n_ok / n_all > 0.97, (n_ok, n_all, worst)
    assert torch.isfinite(buf.advantages).all() and torch.isfinite(buf.returns).all()
    eng.close()


def test_closed_loop_single_ego_follows_the_route():
    """BASELINE config 1 (run_pure_mpc.py: single ego, horizon 20, one other vehicle), closed loop over 150 steps:
    the ego tracks the reference path at the reference speed, yields when the detector fires, and arrives."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("run_pure_mpc", os.path.join(ROOT, "tools", "run_pure_mpc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    outcome, log, lat = mod.run(steps=150, n_others=0, seed=3, verbose=False)     # free road
    log = np.array([r[:5] for r in log])
    assert outcome == "arrived"
    from mpc_rl_for_avs_amd.reference_path import reference_states
    ref = reference_states()
    d = np.min(np.linalg.norm(ref[None, :, :2] - log[:, None, :2], axis=2), axis=1)
    assert d.max() < 0.8                                     # stays on the lane (half width 2 m)
    assert abs(np.median(log[5:, 2]) - 10.0) < 1.0           # cruises at the reference speed
    assert np.abs(log[:, 3]).max() <= 5.0 + 1e-6 and np.abs(log[:, 4]).max() <= np.pi / 3 + 1e-6
    outcome2, log2, _ = mod.run(steps=150, n_others=1, seed=1, verbose=False)     # with cross traffic
    assert outcome2 in ("arrived", "crashed", "timeout", "running")
    assert np.mean([r[6] == 0 for r in log2]) > 0.8


def test_learn_loop_with_mpc_in_the_loop_on_device():
    """collect -> PPO update -> collect with the real engine and everything on the GPU (config 4 in miniature)."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout
    dev = torch.device("cuda", 0)
    eng = engine.MPCEngine(horizon=20, max_iter=100)
    env = rollout.SyntheticIntersectionEnv(64, device=dev, seed=4, n_others=3)
    pol = rollout.ActorCritic(1).to(dev)
    col = rollout.BatchedCollector(env, pol, eng, version="v0", algorithm="ppo", n_steps=8)
    tr = rollout.OnPolicyTrainer(col, n_epochs=2, batch_size=128)
    log = tr.learn(total_timesteps=3 * 64 * 8)
    assert len(log) == 3 and col.num_timesteps == 3 * 64 * 8
    assert all(np.isfinite([r["loss"], r["value_loss"], r["mean_reward"]]).all() for r in log)
    assert (col.last_mpc["status"] == 0).float().mean() > 0.9
    eng.close()


def test_graph_captured_step_and_pipelined_groups():
    """The rollout step replayed as a hipGraph writes the same kind of data as the eager step - every buffer row is the
    MPC action of that row's observation - and two environment groups on two streams feed one PPO update."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout
    dev = torch.device("cuda", 0)
    pol = rollout.ActorCritic(1).to(dev)
    eng = engine.MPCEngine(horizon=20, max_iter=100)
    env = rollout.SyntheticIntersectionEnv(48, device=dev, seed=2, n_others=0)     # no traffic: no detector memory
    col = rollout.BatchedCollector(env, pol, eng, version="v0", algorithm="ppo", n_steps=6, use_graph=True)
    for _ in range(2):
        st = col.collect_rollouts()
    assert st["steps"] == 6 * 48 and col.num_timesteps == 2 * 6 * 48 and col.buffer.pos == 6
    b = col.buffer
    assert int(b.pos_dev.item()) == 6
    assert torch.isfinite(b.rewards).all() and torch.isfinite(b.advantages).all() and torch.isfinite(b.mpc_actions).all()
    assert float(b.obs[:, :, 0, 0].min()) == 1.0                       # every row holds an observation (ego present)
    assert float((b.obs[1:, :, 0, 1:3] - b.obs[:-1, :, 0, 1:3]).abs().max()) > 0.1     # ... and time advances
    # the actions stored next to the observations are what a fresh engine computes for them, i.e. the replayed graph
    # really ran policy clipping, preamble and solve on those rows
    chk = engine.MPCEngine(horizon=20, max_iter=100)
    w = torch.ones((48, 3), dtype=torch.float64, device=dev)
    for t in (0, 3, 5):
        rs = torch.clamp(b.actions[t, :, 0], -1.0, 1.0).to(torch.float64).contiguous()
        out = chk.predict_batch_torch(b.obs[t].contiguous(), w, rs, sync=True)
        assert torch.allclose(out["act"], b.mpc_actions[t], rtol=0, atol=1e-9)
    chk.close()
    # two groups, one update
    engs = [engine.MPCEngine(horizon=20, max_iter=100) for _ in range(2)]
    cols = [rollout.BatchedCollector(rollout.SyntheticIntersectionEnv(32, device=dev, seed=10 + g, n_others=3), pol,
                                     engs[g], version="v0", algorithm="ppo", n_steps=6, seed=g, use_graph=bool(g))
            for g in range(2)]
    pipe = rollout.PipelinedCollector(cols)
    tr = rollout.OnPolicyTrainer(pipe, n_epochs=1, batch_size=128)
    log = tr.learn(total_timesteps=2 * 64 * 6)
    assert len(log) == 2 and pipe.num_timesteps == 2 * 64 * 6 and log[0]["steps"] == 64 * 6
    assert pipe.flat_buffer()[0].shape == (64 * 6, 10, 8)
    assert all(np.isfinite([r["loss"], r["mean_reward"]]).all() for r in log)
    with pytest.raises(ValueError):
        rollout.PipelinedCollector([cols[0], rollout.BatchedCollector(cols[1].env, pol, engs[0], n_steps=6)])
    for e in engs + [eng]:
        e.close()


def test_warm_start_flag(oracle, ref_table):
    """MPC_FLAG_WARM_START (opt-in, the reference always cold-starts): given the same initial controls the engine and
    the oracle walk the same iterates; in closed loop (previous solution advanced one stage) the solves need fewer
    iterations and mostly end in the same minimiser as cold-started ones."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout, synth
    inp = synth.solver_inputs(256, 4, seed=8)
    e = engine.MPCEngine(horizon=20, max_iter=100)
    cold = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"])
    init = np.concatenate([cold["U"][:, 1:], cold["U"][:, -1:]], axis=1)
    init[::9] = 40.0                                                    # far outside the bounds: clamped
    got = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"], u_init=init)
    want = oracle.solve_batch(ref_table, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                              vref=inp["vref"], u_init=init, max_iter=100, xy_bounds=False)
    both = (got["status"] == 0) & (want["status"] == 0)
    assert both.mean() > 0.9 and (got["status"] == want["status"]).mean() > 0.98
    assert (rel_u0_err(got["u0"], want["u0"])[both] <= 1e-4).mean() > 0.995
    assert (got["iters"] == want["iters"])[both].mean() > 0.95
    e.close()
    # closed loop: two identical environments and policies, one engine warm-started
    dev = torch.device("cuda", 0)
    runs = {}
    for warm in (False, True):
        eng = engine.MPCEngine(horizon=20, max_iter=100)
        env = rollout.SyntheticIntersectionEnv(128, device=dev, seed=21, n_others=3)
        torch.manual_seed(0)
        pol = rollout.ActorCritic(1).to(dev)
        # eager steps: the spy below sits on the Python call, which a replayed hipGraph does not pass through
        col = rollout.BatchedCollector(env, pol, eng, version="v0", algorithm="ppo", n_steps=12, seed=5, warm_start=warm,
                                       use_graph=False)
        iters, conv = [], []
        inner = eng.predict_batch_torch

        def spy(*a, **k):
            out = inner(*a, **k)
            iters.append(out["iters"].float().mean().item())
            conv.append((out["status"] == 0).float().mean().item())
            return out

        eng.predict_batch_torch = spy
        col.collect_rollouts()
        runs[warm] = (np.array(iters), np.array(conv), col.buffer.mpc_actions[0].cpu().numpy())
        eng.close()
    (ic, cc_, a0c), (iw, cw, a0w) = runs[False], runs[True]
    assert np.array_equal(a0c, a0w)                                     # first step: nothing to start from yet
    assert iw[1:].mean() < ic[1:].mean()                                # fewer iterations afterwards (a few %: the
    #                                                                     barrier path is walked again)
    assert cw.mean() > cc_.mean() - 0.03


def test_reference_call_sequence_with_separate_check_collision():
    """The reference's own sequence `_parse_obs -> _check_collision -> (read is_collide) -> _solve`
    (agents/pure_mpc.py:68-78 spelled out by a caller) equals `predict()`: the stand-alone `_check_collision` refreshes
    the flag BEFORE the solve and the detector is advanced once per observation, not twice."""
    from mpc_rl_for_avs_amd.pure_mpc import PureMPC_Agent
    a, b = PureMPC_Agent(Env(), dict(CFG)), PureMPC_Agent(Env(), dict(CFG))
    obs = np.zeros((10, 8), np.float32)
    obs[0] = [1, 2.0, 30.0, 0.0, -10.0, -np.pi / 2, -1.0, 0.0]
    obs[1] = [1, -15.0, 12.0, 8.0, 0.0, 0.0, 0.0, 1.0]              # crosses the ego path ahead
    gone = obs.copy()
    gone[1:] = 0
    for step, o in enumerate([obs, gone, gone, gone, obs, gone]):
        want = a.predict(o, return_numpy=False)
        b._parse_obs(o)
        assert b._detected is False
        b._check_collision()
        flag_before_solve = (b.is_collide, b.collision_memory, list(b.conflict_index))
        got = b._solve()
        assert flag_before_solve == (a.is_collide, a.collision_memory, list(a.conflict_index)), step
        assert (b.is_collide, b.collision_memory) == (a.is_collide, a.collision_memory), step
        assert got.acceleration == want.acceleration and got.steer == want.steer, step
        assert got.success and got.status in (0, 5) and got.iters > 0
        again = b._solve()                                           # a second _solve never touches the detector
        assert b.collision_memory == a.collision_memory and again.acceleration == want.acceleration
    # an RL speed override given to _solve after a stand-alone detection is honoured (the profile is derived in _solve,
    # agents/pure_mpc.py:113-117, and takes precedence over the collision profile, :683-688)
    c, d = PureMPC_Agent(Env(), dict(CFG)), PureMPC_Agent(Env(), dict(CFG))
    rs = np.array([[0.7]])
    want = c.predict(obs, ref_speed=rs)
    d._parse_obs(obs)
    d._check_collision()
    assert d.is_collide
    got = d._solve(ref_speed_from_RL=rs)
    assert got.acceleration == want[0] and got.steer == want[1] and want[0] < -1.0


def test_single_agent_attributes_and_checkpoint():
    """`predict()` of the product agent (one device call, B = 1) leaves the attributes the reference's callers and plots
    read (agents/pure_mpc.py:38-43, 589-593), equal to the numpy mirror's; the detector record can be saved and put back
    (`mpc_save_env_state` / `mpc_set_env_state`), after which the agent repeats itself exactly."""
    from mpc_rl_for_avs_amd.pure_mpc import PureMPC_Agent
    from host_preamble import HostPreambleAgent
    a = PureMPC_Agent(Env(), dict(CFG))
    m = HostPreambleAgent(Env(), dict(CFG), engine=a._engine)
    obs = np.zeros((10, 8), np.float32)
    obs[0] = [1, 2.0, 30.0, 0.0, -10.0, -np.pi / 2, -1.0, 0.0]      # ego driving down the approach lane
    obs[1] = [1, -15.0, 12.0, 8.0, 0.0, 0.0, 0.0, 1.0]              # crosses the ego path at (2, 12) ahead
    obs[2] = [1, 40.0, -2.0, -8.0, 0.0, np.pi, 0.0, -1.0]           # far away on another lane: no crossing
    act = a.predict(obs)
    want = m.predict(obs)
    assert act.shape == (2,) and np.allclose(act, want, rtol=0, atol=1e-7)
    assert a.is_collide and m.is_collide and a.collision_memory == m.collision_memory == 10
    assert a.ego_index == m.ego_index == 19
    assert a.conflict_index == m.conflict_index == [37, None]
    assert a.agent_collide == [True, False]
    assert np.allclose(a.conflict_points[0], m.conflict_points[0], atol=1e-9) and a.conflict_points[1] is None
    assert np.allclose(a.stop_point, m.stop_point)
    assert len(a.agent_current_locations) == 2 and np.allclose(a.agent_current_locations[0], [-15.0, 12.0])
    fut = a.agent_future_locations
    assert len(fut) == 2 and len(fut[0]) == 31 and np.allclose(fut[0][-1], [-15.0 + 30 * 0.8, 12.0], atol=1e-4)
    assert a.last_acc == act[0]
    # the vehicle disappears: memory counts down on both sides
    obs2 = obs.copy()
    obs2[1:] = 0
    for i in range(3):
        assert np.allclose(a.predict(obs2), m.predict(obs2), rtol=0, atol=1e-7)
        assert a.is_collide and a.collision_memory == m.collision_memory == 9 - i
    # checkpoint, run on, restore, repeat
    saved = a.save_env_state(1)
    assert saved.shape == (1, a._engine._lib.mpc_env_state_bytes())
    run1 = [a.predict(obs2).copy() for _ in range(9)]
    mem1 = a.collision_memory
    a.load_env_state(saved)
    run2 = [a.predict(obs2).copy() for _ in range(9)]
    assert all(np.array_equal(x, y) for x, y in zip(run1, run2)) and a.collision_memory == mem1
    # a corrupted record is refused as a whole (counts are loop bounds, indices subscripts on the device)
    from mpc_rl_for_avs_amd.engine import EngineError
    for word, val in ((3, 99), (5, 85), (8, 10 ** 6), (0, -1)):      # n_conflict, ego_index, conflict[0], collision_memory
        bad = saved.copy()
        bad.view(np.int32)[0, word] = val
        with pytest.raises(EngineError, match="not a valid detector state"):
            a.load_env_state(bad)
    # into another engine
    b = PureMPC_Agent(Env(), dict(CFG))
    b.load_env_state(saved)
    assert np.array_equal(b.predict(obs2), run1[0])
    b.reset_env_state()
    b.predict(obs2)
    assert not b.is_collide and b.conflict_index == []


def test_fused_environment_step_on_the_gpu_equals_its_host_build_and_the_torch_ops():
    """`mpc_synth_env_step` (one launch per policy step: models, respawn, reward, termination, terminal observation,
    auto-reset, next observation) against (a) the same source compiled for the host, same seed: same episodes, resets
    and respawns included; (b) the torch implementation it replaces on the GPU, for the deterministic part."""
    import ctypes
    import torch
    from mpc_rl_for_avs_amd import rollout
    from test_synth_env_cpu import HostEnv
    import subprocess, os
    from conftest import BUILD_DIR, HOST_CXXFLAGS, ROOT
    out = os.path.join(BUILD_DIR, "libcpu_synth_env.so")
    if not os.path.exists(out):
        os.makedirs(BUILD_DIR, exist_ok=True)
        subprocess.run(["g++"] + HOST_CXXFLAGS + ["-o", out, os.path.join(ROOT, "tests", "cpu_synth_env_harness.cpp")], check=True)
    hostlib = ctypes.CDLL(out)
    B, K = 512, 4
    dev = torch.device("cuda:0")
    g = rollout.SyntheticIntersectionEnv(B, device=dev, seed=21, n_others=K, spawn_probability=0.3)
    assert g.backend == "hip"
    h = HostEnv(hostlib, B, K, seed=21, spawn_probability=0.3)
    assert np.allclose(g.reset().cpu().numpy(), h.reset(), rtol=0, atol=1e-5)
    rng = np.random.default_rng(0)
    ended = 0
    for step in range(150):
        act = np.stack([rng.uniform(-3, 5, B), 0.03 * rng.uniform(-1, 1, B)], axis=1)
        o_g, r_g, d_g, info = g.step(torch.as_tensor(act, device=dev))
        o_h, r_h, d_h = h.step(act)
        assert np.array_equal(d_g.cpu().numpy(), d_h), step
        for k in ("crashed", "arrived", "truncated"):
            assert np.array_equal(info[k].cpu().numpy(), h.flags[k].astype(bool)), (step, k)
        assert np.allclose(r_g.cpu().numpy(), r_h, rtol=0, atol=1e-4)
        assert np.allclose(o_g.cpu().numpy(), o_h, rtol=0, atol=1e-4) and np.allclose(info["terminal_obs"].cpu().numpy(), h.tobs, rtol=0, atol=1e-4)
        assert np.allclose(g.ego.cpu().numpy(), h.ego, rtol=0, atol=1e-8) and np.array_equal(g.oactive.cpu().numpy(), h.oactive.astype(bool))
        ended += int(d_h.sum())
    assert ended >= 50 and np.array_equal(g.rng_counter.cpu().numpy(), h.ctr)
    # (b) against the torch ops, no randomness in the step
    gt = rollout.SyntheticIntersectionEnv(B, device=dev, seed=5, n_others=K, spawn_probability=0.0, backend="torch")
    gh = rollout.SyntheticIntersectionEnv(B, device=dev, seed=5, n_others=K, spawn_probability=0.0, backend="hip")
    gh.reset()
    for a, b in ((gt.ego, gh.ego), (gt.opos, gh.opos), (gt.ospeed, gh.ospeed), (gt.ohead, gh.ohead), (gt.oactive, gh.oactive), (gt.t, gh.t)):
        a.copy_(b)
    alive = torch.ones(B, dtype=torch.bool, device=dev)
    for step in range(60):
        act = torch.as_tensor(np.stack([rng.uniform(-3, 5, B), 0.03 * rng.uniform(-1, 1, B)], axis=1), device=dev)
        o1, r1, d1, i1 = gt.step(act)
        o2, r2, d2, i2 = gh.step(act)
        assert torch.equal(d1[alive], d2[alive]) and torch.allclose(r1[alive], r2[alive], rtol=0, atol=1e-4)
        assert torch.allclose(i1["terminal_obs"][alive], i2["terminal_obs"][alive], rtol=0, atol=1e-4)
        alive = alive & ~d2
        assert torch.allclose(gt.ego[alive], gh.ego[alive], rtol=0, atol=1e-9)
    assert int((~alive).sum()) >= 5


@pytest.mark.parametrize("K, B", [(0, 5), (1, 66), (9, 131), (4, 256)])
def test_sixteen_lanes_per_environment_step_for_every_vehicle_count(K, B):
    """`mpc_synth_env_rows_kernel` (round 5: lane j of a 16-lane group = vehicle j, four environments per wave) against the
    host build of the one-thread statement `env::step_env`, same seed: no traffic, one vehicle, the most the observation
    holds (9), and batch sizes that leave the last wave partly empty.  Episodes end and vehicles respawn along the way."""
    import ctypes
    import subprocess
    import torch
    from mpc_rl_for_avs_amd import rollout
    from test_synth_env_cpu import HostEnv
    from conftest import BUILD_DIR, HOST_CXXFLAGS, ROOT
    out = os.path.join(BUILD_DIR, "libcpu_synth_env.so")
    if not os.path.exists(out):
        os.makedirs(BUILD_DIR, exist_ok=True)
        subprocess.run(["g++"] + HOST_CXXFLAGS + ["-o", out, os.path.join(ROOT, "tests", "cpu_synth_env_harness.cpp")], check=True)
    hostlib = ctypes.CDLL(out)
    dev = torch.device("cuda:0")
    g = rollout.SyntheticIntersectionEnv(B, device=dev, seed=9, n_others=K, spawn_probability=0.5)
    assert g.backend == "hip"
    h = HostEnv(hostlib, B, K, seed=9, spawn_probability=0.5)
    assert np.allclose(g.reset().cpu().numpy(), h.reset(), rtol=0, atol=1e-5)
    rng = np.random.default_rng(K)
    ended = 0
    for step in range(120):
        act = np.stack([rng.uniform(-3, 5, B), 0.05 * rng.uniform(-1, 1, B)], axis=1)
        o_g, r_g, d_g, info = g.step(torch.as_tensor(act, device=dev))
        o_h, r_h, d_h = h.step(act)
        assert np.array_equal(d_g.cpu().numpy(), d_h), step
        for k in ("crashed", "arrived", "truncated"):
            assert np.array_equal(info[k].cpu().numpy(), h.flags[k].astype(bool)), (step, k)
        assert np.allclose(r_g.cpu().numpy(), r_h, rtol=0, atol=1e-4)
        assert np.allclose(o_g.cpu().numpy(), o_h, rtol=0, atol=1e-4) and np.allclose(info["terminal_obs"].cpu().numpy(), h.tobs, rtol=0, atol=1e-4)
        assert np.allclose(g.ego.cpu().numpy(), h.ego, rtol=0, atol=1e-8) and np.array_equal(g.oactive.cpu().numpy(), h.oactive.astype(bool))
        assert np.allclose(g.opos.cpu().numpy(), h.opos, rtol=0, atol=1e-8) and np.array_equal(g.t.cpu().numpy(), h.t)
        ended += int(d_h.sum())
    assert (ended >= 1 or B < 16) and np.array_equal(g.rng_counter.cpu().numpy(), h.ctr)


def test_v1_rollout_converges_like_v0():
    """the config-4 rollout with v1 actions (the three cost weights anywhere in [-1, 1]^3, agents/ppo_mpc.py:
    407-420, an untrained Gaussian policy) - at least 99.5 % of the rollout's solves end converged at the default settings
    (round 4: 97.7 %)."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    pol = rollout.ActorCritic(3).to(dev)
    eng = engine.MPCEngine(horizon=20, max_iter=100)
    env = rollout.SyntheticIntersectionEnv(256, device=dev, seed=0, n_others=4)
    col = rollout.BatchedCollector(env, pol, eng, version="v1", algorithm="ppo", n_steps=32, collision_cost=False, seed=0)
    col.collect_rollouts()
    stats = col.collect_rollouts()
    assert stats["mpc_unconverged"] <= 0.005 * 256 * 32, stats
    eng.close()


def test_graph_and_eager_collectors_produce_the_same_rollout():
    """capturing the step as a hipGraph really steps the environment and the detector three times; all of that
    is put back (episodes, both random streams, the observation, the engine's records), so a graph collector and an eager
    one built from the same seeds collect the same rollout - and detector records a caller restored before building the
    collector (resume) are still there afterwards."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout
    dev = torch.device("cuda", 0)
    B, T = 64, 10
    bufs = []
    for use_graph in (True, False):
        torch.manual_seed(7)
        pol = rollout.ActorCritic(1).to(dev)
        eng = engine.MPCEngine(horizon=20, max_iter=100)
        env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=5, n_others=4)
        col = rollout.BatchedCollector(env, pol, eng, version="v0", algorithm="ppo", n_steps=T, seed=3, use_graph=use_graph)
        assert (col._graph is not None) == use_graph
        col.collect_rollouts()
        b = col.buffer
        bufs.append([x.clone() for x in (b.obs, b.actions, b.mpc_actions, b.rewards, b.values, b.log_probs)])
        second = col.collect_rollouts()               # and the rollout after it
        bufs[-1] += [b.obs.clone(), b.mpc_actions.clone()]
        assert second["steps"] == B * T
        eng.close()
    for g, e in zip(*bufs):
        assert torch.equal(g, e)
    # resume: records restored BEFORE the collector is built survive its construction
    eng = engine.MPCEngine(horizon=20, max_iter=100)
    env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=5, n_others=4)
    eng.predict_batch_torch(env.reset().clone(), torch.ones((B, 3), dtype=torch.float64, device=dev), sync=True)
    recs = eng.save_env_state(B)
    assert recs.view(np.int32)[:, 0].max() == 10      # collision memories running
    eng2 = engine.MPCEngine(horizon=20, max_iter=100)
    eng2.load_env_state(recs)
    torch.manual_seed(7)
    rollout.BatchedCollector(env, rollout.ActorCritic(1).to(dev), eng2, version="v0", n_steps=T, seed=3, use_graph=True)
    assert np.array_equal(eng2.save_env_state(B), recs)
    eng.close()
    eng2.close()


@pytest.mark.parametrize("version,algorithm", [("v0", "ppo"), ("v1", "ppo"), ("v0", "a2c")])
def test_fused_glue_step_equals_the_torch_step(version, algorithm):
    """mpc_policy_act + mpc_rollout_record (csrc/mpc_rollout_glue.hpp) against the torch ops they replace: same seeds, the
    SAME random draws (the torch path is fed the noise the kernel drew), the same rollout - up to the summation order of three
    float32 matrix products."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout
    dev = torch.device("cuda", 0)
    B, T = 96, 6
    out = []
    for fused in (True, False):
        torch.manual_seed(11)
        pol = rollout.ActorCritic(3 if version == "v1" else 1).to(dev)
        with torch.no_grad():
            pol.log_std.fill_(-1.0)
        eng = engine.MPCEngine(horizon=20, max_iter=100)
        env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=9, n_others=4)
        col = rollout.BatchedCollector(env, pol, eng, version=version, algorithm=algorithm, n_steps=T, seed=4,
                                       use_graph=False, fused_glue=fused)
        assert col.fused_glue == fused
        if fused:
            col._begin_rollout()
            noises = []
            for _ in range(T):
                col._step()
                noises.append(col._fg["noise"].clone())
            assert int(col._fg["step"]) == T
            col._finish_rollout()
            stats = col._rollout_stats(T)
            n = torch.stack(noises)
            assert abs(float(n.mean())) < 0.15 and abs(float(n.std()) - 1.0) < 0.15 and not torch.equal(noises[0], noises[1])
        else:
            col.noise_feed = [x.clone() for x in noises]
            stats = col.collect_rollouts()
            assert not col.noise_feed
        b = col.buffer
        out.append(dict(stats=stats, obs=b.obs.clone(), actions=b.actions.clone(), values=b.values.clone(),
                        logp=b.log_probs.clone(), rewards=b.rewards.clone(), starts=b.episode_starts.clone(),
                        mpc=b.mpc_actions.clone(), adv=b.advantages.clone(), last=col._last_obs.clone(),
                        term=None if b.terminal_obs is None else b.terminal_obs.clone(),
                        trunc=None if b.truncated is None else b.truncated.clone()))
        eng.close()
    f, t = out
    # step 0: the same observation on both sides - the policy's outputs agree to float32 summation order, the MPC's actions
    # to what a 1e-7 change of its inputs does
    assert torch.equal(f["obs"][0], t["obs"][0]) and torch.equal(f["starts"][0], t["starts"][0])
    assert torch.allclose(f["actions"][0], t["actions"][0], atol=2e-5) and torch.allclose(f["values"][0], t["values"][0], atol=2e-5)
    assert torch.allclose(f["logp"][0], t["logp"][0], atol=1e-4)
    d0 = (f["mpc"][0] - t["mpc"][0]).abs().amax(dim=-1)
    assert d0.quantile(0.9) < 1e-4
    assert torch.allclose(f["rewards"][0], t["rewards"][0], atol=1e-2) or version == "v1"
    # later steps: the same rollout as long as no solve lands in another minimum (v1 hands the MPC negative cost weights:
    # non-convex problems, where a 1e-7 change of an input may do that) - required of most environments, not of all
    same = (f["obs"] - t["obs"]).abs().amax(dim=(2, 3)) < 1e-3                # [T, B]
    assert same.float().mean() > (0.6 if version == "v1" else 0.97), same.float().mean()
    da = (f["actions"] - t["actions"]).abs().amax(dim=-1)
    assert da[same].max() < 1e-3
    assert torch.equal(f["starts"][same], t["starts"][same])
    if version == "v0":
        assert f["stats"]["episodes"] == t["stats"]["episodes"] and f["stats"]["crashed"] == t["stats"]["crashed"]
        assert abs(f["stats"]["mpc_unconverged"] - t["stats"]["mpc_unconverged"]) <= 3
    if f["term"] is not None:
        assert torch.equal(f["trunc"][same], t["trunc"][same])
    assert torch.isfinite(f["adv"]).all() and torch.isfinite(t["adv"]).all()


@pytest.mark.parametrize("algorithm", ["ppo", "a2c"])
def test_rollout_finish_on_the_gpu_is_the_torch_gae_bit_for_bit(algorithm):
    """mpc_rollout_finish (truncation bootstrap + generalised advantage estimation in one launch) against the torch form of
    the same float32 arithmetic (RolloutBuffer.bootstrap_truncated + compute_returns_and_advantage) on the buffer a real
    rollout left behind - same buffer, same value estimates: advantages, returns and bootstrapped rewards identical."""
    import torch
    from mpc_rl_for_avs_amd import engine, rollout
    dev = torch.device("cuda", 0)
    B, T = 160, 48
    torch.manual_seed(3)
    pol = rollout.ActorCritic(1).to(dev)
    eng = engine.MPCEngine(horizon=20, max_iter=60)
    env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=2, n_others=4)
    col = rollout.BatchedCollector(env, pol, eng, algorithm=algorithm, n_steps=T, seed=1, fused_glue=True)
    col._begin_rollout()
    for _ in range(T):
        col._step()
    buf = col.buffer
    if buf.truncated is not None:                  # 48 steps truncate nothing by themselves: mark some
        buf.truncated.copy_((torch.rand(T, B, device=dev) < 0.1).float())
    before = buf._row.clone()
    col._finish_rollout()                          # the kernel
    got = dict(row=buf._row.clone(), adv=buf.advantages.clone(), ret=buf.returns.clone())
    buf._row.copy_(before)
    with torch.no_grad():
        last_values = pol.predict_values(col._last_obs)
        buf.bootstrap_truncated(pol.predict_values)
        buf.compute_returns_and_advantage(last_values, col._roll["dones"])
    assert torch.equal(got["row"], buf._row)
    assert torch.equal(got["adv"], buf.advantages) and torch.equal(got["ret"], buf.returns)
    if algorithm == "ppo":
        assert not torch.equal(before, buf._row)   # the bootstrap did change rewards
    assert got["adv"].abs().max() > 0
    eng.close()
    # the longest rollout the entry point takes (8192 steps: 64 KB of LDS per environment) on a synthetic buffer
    import ctypes
    T2, B2 = 8192, 3
    big = rollout.RolloutBuffer(T2, B2, 1, dev, keep_terminal=False)
    ref = rollout.RolloutBuffer(T2, B2, 1, dev, keep_terminal=False)
    big._row.copy_(torch.randn(big._row.shape, device=dev) * 0.1)
    big.episode_starts.copy_((torch.rand(T2, B2, device=dev) < 0.01).float())
    ref._row.copy_(big._row)
    lv, dn = torch.randn(B2, device=dev), torch.zeros(B2, dtype=torch.uint8, device=dev)
    q = lambda t: ctypes.c_void_p(t.data_ptr())
    lib = engine.load_library()
    rc = lib.mpc_rollout_finish(0, T2, B2, 1, big._cols, 0, q(big._row), q(lv), q(dn), None, 0.99, 0.95, q(big.advantages),
                                q(big.returns), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0, lib.mpc_last_error()
    ref.compute_returns_and_advantage(lv, dn.bool())
    assert torch.equal(big.advantages, ref.advantages) and torch.equal(big.returns, ref.returns)


@pytest.mark.gpu
def test_concurrent_streams_are_probed_not_assumed():
    """engine.concurrent_streams (mpc_streams_overlap): the streams it returns run kernels side by side pairwise - the probe's
    own statement, checked again stream by stream - a stream does not "overlap" with itself, and two batches in flight on two
    of them give the results of the same batches solved one after the other."""
    import torch
    from mpc_rl_for_avs_amd import engine, synth
    dev = torch.device("cuda", 0)
    streams = engine.concurrent_streams(4, dev)
    assert len({s.cuda_stream for s in streams}) == 4
    for i in range(4):
        assert not engine.streams_overlap(streams[i], streams[i], 0)
        for j in range(i + 1, 4):
            assert engine.streams_overlap(streams[i], streams[j], 0)
    inp = synth.solver_inputs(1024, 4, seed=5)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    args = dict(state=t(inp["state"], torch.float64), ego_index=t(inp["ego_index"], torch.int32), weights=t(inp["weights"], torch.float64),
                is_collide=t(inp["is_collide"], torch.uint8), vref=t(inp["vref"], torch.float64), others=t(inp["others"], torch.float64),
                collision_cost=True)
    e = engine.MPCEngine(horizon=20, max_iter=60)
    want = e.solve_batch_torch(**args, throughput=True, sync=True)
    outs = []
    for s in streams[:2]:
        with torch.cuda.stream(s):
            outs.append(e.solve_batch_torch(**args, throughput=True))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o["u0"], want["u0"]) and torch.equal(o["status"], want["status"]) and torch.equal(o["iters"], want["iters"])
    e.close()
