"""Generates the golden fixtures under tests/golden/ (run in the build container; the GPU box only reads them).

1. reference_numpy.npz - outputs of the numpy-only pieces of the reference itself, imported from
   /root/reference with *empty* stand-in modules for the third-party packages that are absent offline
   (gymnasium is used only in type annotations, casadi / shapely / matplotlib only inside `_solve`,
   `_check_collision` and the plotting methods, none of which is executed here).  Covers
   `Agent.reference_states`, `_parse_obs`, `normalize_angle`, `update_reference_states` (all three branches),
   `predict_ego_future_positions` and `predict_future_positions`.
2. oracle_solutions.npz - inputs and KKT-certified solutions of the CPU oracle for 32 synthetic instances per
   configuration (the reference's own solver stack cannot run here, so these pin OUR oracle, not IPOPT).

3. ltv_reference_numpy.npz - outputs of the numpy-only helpers of the reference's iterative-linear agent
   (agents/pure_mpc_linear.py: calc_nearest_index_in_direction, linear_model_matrix, predict_motion), imported with an
   empty stand-in for cvxpy (used only inside `_linear_mpc_control`, which is not executed).

4. ltv_oracle_solutions.npz - inputs and solutions of the CPU oracle of the iterative-linear agent's QP
   (oracle/ltv_oracle.py) for 48 synthetic instances, first call and a call re-linearised about that solution, with
   the objective value of an independent scipy SLSQP run for the first 6 (these pin OUR oracle, not ECOS).

The fixtures are data (inputs / expected outputs) only.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def reference_vectors():
    ref_root = "/root/reference"
    for name in ("gymnasium", "casadi", "shapely", "shapely.errors", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["gymnasium"].Env = object
    sys.modules["shapely"].LineString = object
    sys.modules["shapely.errors"].GEOSException = Exception
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.path.insert(0, ref_root)
    from agents.pure_mpc import PureMPC_Agent  # noqa: E402

    class Env:
        unwrapped = None
        config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}
    Env.unwrapped = Env
    cfg = dict(horizon=20, render=False, ttc_threshold=3, weight_speed=1, weight_control=1, weight_input_diff=1,
               speed_override=0)
    ag = PureMPC_Agent(Env, cfg)
    out = {"reference_states": ag.reference_states}

    rng = np.random.default_rng(123)
    from mpc_rl_for_avs_amd import synth
    obs = synth.make_obs_batch(6, 5, seed=42)
    obs[1, 0, 5] = np.float32(3.5)      # heading beyond +pi -> wrapped
    obs[2, 0, 5] = np.float32(-3.3)     # heading beyond -pi -> wrapped
    obs[3, 3:, 0] = 0                   # fewer vehicles present
    parsed = []
    for b in range(6):
        ag._parse_obs(obs[b])
        row = [ag.ego_vehicle.position[0], ag.ego_vehicle.position[1], ag.ego_vehicle.heading, ag.ego_vehicle.speed,
               ag.observed_vehicles_count]
        oth = np.zeros((9, 4))
        for j, v in enumerate(ag.agent_vehicles):
            oth[j] = (v.position[0], v.position[1], v.speed, v.heading)
        parsed.append((np.array(row, dtype=np.float64), oth))
    out["parse_obs_in"] = obs
    out["parse_ego"] = np.stack([p[0] for p in parsed])
    out["parse_others"] = np.stack([p[1] for p in parsed])
    ang = np.array([0.0, 3.2, -3.2, 7.0, -7.0, np.pi, -np.pi, 10.0])
    out["normalize_in"] = ang
    out["normalize_out"] = np.array([ag.normalize_angle(a) for a in ang])

    # update_reference_states: RL override, no collision, collision (several index patterns)
    cases_in, cases_out = [], []
    ag._parse_obs(obs[0])
    for (ego_index, is_collide, conflict, mem, rl, speed) in [
            (4, False, [None], 0, None, 8.0), (4, False, [None], 0, 0.7, 8.0), (10, True, [30, None], 0, None, 9.5),
            (10, True, [12, 40], 10, None, 3.25), (50, True, [52], 0, None, 11.0), (83, True, [84], 0, None, 5.0),
            (20, True, [None, None], 0, None, 6.0), (7, True, [60], 0, 45.0, 6.0), (7, True, [8], 0, None, 0.0)]:
        ag.ego_index = ego_index
        ag.is_collide = is_collide
        ag.conflict_index = conflict
        ag.collision_memory = mem
        ag.memorized_conflict_indices = conflict if mem > 0 else None
        ag.ego_vehicle.speed = speed
        ag.last_valid_stop_point = None
        rs = None if rl is None else np.array([[rl]])
        ref = ag.update_reference_states(speed_override=0, speed_overide_from_RL=rs)
        cases_in.append([ego_index, int(is_collide), mem, np.nan if rl is None else rl, speed] +
                        [(-1 if c is None else c) for c in (conflict + [None])[:2]])
        cases_out.append(ref[:, 2].copy())
    out["update_ref_in"] = np.array(cases_in, dtype=np.float64)
    out["update_ref_speed_out"] = np.stack(cases_out)

    # predictors
    ego_in, ego_out, ego_len = [], [], []
    for (x, y, sp, vref) in [(2.0, 45.0, 10.0, 10.0), (2.3, 30.2, 0.0, 10.0), (1.8, 12.0, 14.0, 10.0),
                             (-3.0, 0.4, 5.0, 10.0), (-30.0, -2.2, 9.0, 10.0), (-36.0, -2.2, 10.0, 10.0),
                             (2.0, 20.5, 3.0, 0.0)]:
        pos = np.array([x, y], dtype=np.float32)
        fut = ag.predict_ego_future_positions(pos, sp, -1.5, 3.5, 0.1, 30, vref)
        arr = np.full((31, 2), np.nan)
        arr[:len(fut)] = np.asarray([np.asarray(p, dtype=np.float64) for p in fut])
        ego_in.append([x, y, sp, vref])
        ego_out.append(arr)
        ego_len.append(len(fut))
    out["ego_future_in"] = np.array(ego_in)
    out["ego_future_out"] = np.stack(ego_out)
    out["ego_future_len"] = np.array(ego_len)
    # the same predictor driven the way `_check_collision` drives it: float32 position and float32 speed straight
    # from a parsed observation (numpy 2 keeps the speed ramp and the first metres of arc length in float32)
    obs32 = synth.make_obs_batch(24, 2, seed=77)
    e32_in, e32_out, e32_len = [], [], []
    for b in range(24):
        ag._parse_obs(obs32[b])
        ego = ag.ego_vehicle
        idx = int(np.argmin(np.linalg.norm(ag.reference_trajectory - ego.position, axis=1)))
        vref = ag.global_reference_states[idx, 2] if b % 6 else 0.5 * float(ego.speed)
        fut = ag.predict_ego_future_positions(ego.position, ego.speed, ego.heading, ego.max_acceleration, ag.dt, 30, vref)
        arr = np.full((31, 2), np.nan)
        arr[:len(fut)] = np.asarray([np.asarray(p, dtype=np.float64) for p in fut])
        e32_in.append([ego.position[0], ego.position[1], ego.speed, vref])
        e32_out.append(arr)
        e32_len.append(len(fut))
    out["ego_future32_in"] = np.array(e32_in, dtype=np.float64)      # float32 values, exactly representable
    out["ego_future32_out"] = np.stack(e32_out)
    out["ego_future32_len"] = np.array(e32_len)
    # stop profile with a float32 ego speed (np.linspace then runs in float32)
    ag._parse_obs(obs32[0])
    lin_in, lin_out = [], []
    for (ego_index, conflict, speed) in [(10, [30], np.float32(9.3)), (3, [40, 12], np.float32(3.3)),
                                         (50, [52], np.float32(11.7)), (20, [27], np.float32(0.0))]:
        ag.ego_index, ag.is_collide, ag.conflict_index, ag.collision_memory = ego_index, True, conflict, 0
        ag.memorized_conflict_indices = None
        ag.ego_vehicle.speed = speed
        ag.last_valid_stop_point = None
        ref = ag.update_reference_states(speed_override=0, speed_overide_from_RL=None)
        lin_in.append([ego_index, min(conflict), float(speed)])
        lin_out.append(ref[:, 2].copy())
    out["stop_profile32_in"] = np.array(lin_in, dtype=np.float64)
    out["stop_profile32_out"] = np.stack(lin_out)
    fut = ag.predict_future_positions(np.array([-20.0, 2.0], dtype=np.float32), np.float32(8.0), np.float32(0.1), 0.1, 30)
    out["agent_future_out"] = np.asarray(fut, dtype=np.float64)
    del rng
    np.savez_compressed(os.path.join(HERE, "reference_numpy.npz"), **out)
    print("wrote reference_numpy.npz", {k: v.shape for k, v in out.items()})


# ---------------------------------------------------------------------------------------------------------------
# round 4: >= 500 randomised cases per function, closed-loop sequences through the reference's own predict(), and the
# reference's own NLP-building statements evaluated numerically (tests/golden/standins.py says what stands in for the
# absent casadi / shapely and what the fixtures therefore pin)
# ---------------------------------------------------------------------------------------------------------------
class _RefEnv:
    unwrapped = None
    config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}


_RefEnv.unwrapped = _RefEnv
_REF_CFG = dict(horizon=20, render=False, ttc_threshold=3, weight_speed=1, weight_control=1, weight_input_diff=1,
                speed_override=0)


def _reference_agent(horizon=20):
    """The reference's live PureMPC_Agent, imported from /root/reference with the stand-ins installed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, HERE)
    import standins
    standins.install()
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from agents.pure_mpc import PureMPC_Agent
    cfg = dict(_REF_CFG, horizon=horizon)
    return PureMPC_Agent(_RefEnv, cfg), standins


def _run_solve_head(ag, standins, weights_from_RL, ref_speed, point=None):
    """The reference's `_solve` (agents/pure_mpc.py:80-300) executed up to its nlpsol call: its own statements compute the
    ego index, call update_reference_states, build objective / constraints / bounds / initial guess - evaluated at
    `point` = (X[N+1, 4], U[N, 2]) by the numeric casadi stand-in.  Returns dict(ego_index, ref (85 x 4), f, g, x0, lbx,
    ubx, lbg, ubg, components, stop_point)."""
    import contextlib
    import copy
    import io
    N = ag.horizon
    if point is None:
        point = (np.zeros((N + 1, 4)), np.zeros((N, 2)))
    standins.POINT.clear()
    standins.POINT.update(x=np.asarray(point[0], np.float64).T.copy(), u=np.asarray(point[1], np.float64).T.copy())
    standins.CAPTURED.clear()
    got = {}
    orig = ag.update_reference_states

    def spy(*a, **k):
        got["ref"] = orig(*a, **k)
        return got["ref"]
    ag.update_reference_states = spy
    saved = copy.deepcopy(ag.agent_vehicles_mpc)      # the cost loop advances these copies (agents/pure_mpc.py:190-191)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ag._solve(weights_from_RL, ref_speed)
        raise AssertionError("the solver stand-in must end _solve")
    except standins.Captured:
        pass
    finally:
        ag.update_reference_states = orig
        ag.agent_vehicles_mpc = saved
    c = standins.CAPTURED
    out = dict(ego_index=int(ag.ego_index), ref=np.array(got["ref"], np.float64), f=c["f"], g=c["g"].copy(),
               x0=c["x0"].copy(), lbx=c["lbx"].copy(), ubx=c["ubx"].copy(), lbg=c["lbg"].copy(), ubg=c["ubg"].copy(),
               components=np.array(c["functions"]["cost_fn"], np.float64), opts=dict(c["opts"]))
    sp = getattr(ag, "stop_point", None)
    out["stop_index"] = -1 if sp is None else int(np.argmin(np.linalg.norm(ag.reference_trajectory - np.asarray(sp), axis=1)))
    return out


def _ego_obs(x, y, heading, speed, rows=10):
    """One observation with only the ego row present; vx, vy chosen so that the float32 |v| is exactly `speed`."""
    o = np.zeros((rows, 8), np.float32)
    o[0] = [1.0, x, y, speed, 0.0, heading, np.sin(heading), np.cos(heading)]
    return o


def reference_random_vectors(n=512):
    """>= 500 randomised cases for each numpy-only function of the reference on the path, every one produced by
    calling the reference's own code, stored with the float32 observation that makes the device reproduce the case."""
    ag, standins = _reference_agent()
    from mpc_rl_for_avs_amd import synth
    ref = ag.reference_states
    rng = np.random.default_rng(20260104)
    out = {}

    # ---- _parse_obs (agents/base_agent.py:81-116) and normalize_angle (:156-170)
    obs = np.concatenate([synth.make_obs_batch(n // 8, V, seed=900 + V) for V in (0, 1, 2, 3, 5, 7, 9, 9)])[:n]
    wrap = rng.uniform(size=n) < 0.3
    obs[wrap, 0, 5] = rng.uniform(-9.5, 9.5, int(wrap.sum())).astype(np.float32)       # heading far outside [-pi, pi]
    obs[5, 0, 5] = np.float32(np.pi)                                                    # float32(pi) > pi: wrapped once
    obs[6, 0, 5] = np.float32(-np.pi)
    for b in np.nonzero(rng.uniform(size=n) < 0.25)[0]:                                 # fewer vehicles present
        obs[b, rng.integers(1, 10):, 0] = 0
    ego = np.zeros((n, 5))
    oth = np.zeros((n, 9, 4))
    for b in range(n):
        ag._parse_obs(obs[b])
        e = ag.ego_vehicle
        ego[b] = (e.position[0], e.position[1], e.heading, e.speed, ag.observed_vehicles_count)
        for j, v in enumerate(ag.agent_vehicles):
            oth[b, j] = (v.position[0], v.position[1], v.speed, v.heading)
    out.update(parse_obs=obs, parse_ego=ego, parse_others=oth)
    ang = rng.uniform(-14.0, 14.0, n)
    ang[:6] = [np.pi, -np.pi, 3 * np.pi, -3 * np.pi, 0.0, 2 * np.pi]
    out.update(normalize_in=ang, normalize_out=np.array([ag.normalize_angle(a) for a in ang]),
               normalize32_in=ang.astype(np.float32).astype(np.float64),
               normalize32_out=np.array([float(ag.normalize_angle(np.float32(a))) for a in ang]))

    # ---- predict_ego_future_positions (agents/pure_mpc.py:459-527), driven as _check_collision drives it
    vsel = np.array([10.0, 0.0, 2.5, 5.75, 8.0, 12.5, 20.0, 30.0])
    e_obs = np.zeros((n, 10, 8), np.float32)
    e_vref = vsel[rng.integers(0, len(vsel), n)]
    e_out = np.full((n, 31, 2), np.nan)
    e_len = np.zeros(n, np.int64)
    for b in range(n):
        i = int(rng.integers(0, 85))
        off = rng.uniform(-1.5, 1.5, 2) if b % 7 else rng.uniform(-30.0, 30.0, 2)      # some far off the path
        sp = float(rng.uniform(0.0, 16.0)) if b % 11 else 0.0
        e_obs[b] = _ego_obs(ref[i, 0] + off[0], ref[i, 1] + off[1], ref[i, 3] + rng.uniform(-0.2, 0.2), sp)
        ag._parse_obs(e_obs[b])
        e = ag.ego_vehicle
        fut = ag.predict_ego_future_positions(current_position=e.position, speed=e.speed, heading=e.heading,
                                              max_acceleration=e.max_acceleration, dt=ag.dt, prediction_horizon=30,
                                              reference_speed=e_vref[b])
        e_len[b] = len(fut)
        e_out[b, :len(fut)] = np.asarray([np.asarray(q, dtype=np.float64) for q in fut])
    out.update(ego_future_obs=e_obs, ego_future_vref=e_vref, ego_future_out=e_out, ego_future_len=e_len)

    # ---- predict_future_positions (agents/pure_mpc.py:529-550): the constant-velocity polyline of an observed vehicle
    a_obs = synth.make_obs_batch(n, 1, seed=4242)
    a_obs[1::2, 1, 5] += rng.uniform(-0.5, 0.5, n // 2).astype(np.float32)             # every other heading off the lane axes
    a_obs[::9, 1, 3:5] = 0.0                                                            # standing vehicles
    a_out = np.zeros((n, 31, 2))
    for b in range(n):
        ag._parse_obs(a_obs[b])
        v = ag.agent_vehicles[0]
        fut = ag.predict_future_positions(current_position=np.array(v.position), speed=v.speed, heading=v.heading,
                                          dt=ag.dt, prediction_horizon=30)
        assert all(np.asarray(q).dtype == np.float32 for q in fut)
        a_out[b] = np.asarray(fut, dtype=np.float64)
    out.update(agent_future_obs=a_obs, agent_future_out=a_out)

    # ---- update_reference_states (agents/pure_mpc.py:678-724) through the head of _solve (:95-113), from detector states
    #      set by hand: every branch, 0 - 4 vehicles, None entries, memory replay, RL override, last valid stop point
    K = 4
    u_obs = np.zeros((n, 10, 8), np.float32)
    u_in = dict(is_collide=np.zeros(n, np.int32), mem=np.zeros(n, np.int32), n_conf=np.zeros(n, np.int32),
                conflict=np.full((n, K), -1, np.int32), has_mem=np.zeros(n, np.int32), n_mem=np.zeros(n, np.int32),
                memorized=np.full((n, K), -1, np.int32), last_valid=np.full(n, -1, np.int32), rl=np.full(n, np.nan))
    u_col = np.zeros((n, 85))
    u_ego = np.zeros(n, np.int32)
    u_stop = np.zeros(n, np.int32)
    u_last = np.zeros(n, np.int32)

    def draw_list():
        m = int(rng.integers(0, K + 1))
        lst = [None if rng.uniform() < 0.3 else int(rng.integers(0, 85)) for _ in range(m)]
        return lst
    for b in range(n):
        i = int(rng.integers(0, 85))
        off = rng.uniform(-0.45, 0.45, 2)
        sp = float(rng.uniform(0.0, 15.0)) if b % 10 else 0.0
        u_obs[b] = _ego_obs(ref[i, 0] + off[0], ref[i, 1] + off[1], ref[i, 3], sp)
        ag._parse_obs(u_obs[b])
        col = bool(rng.uniform() < 0.8)
        conf = draw_list()
        mem = int(rng.integers(0, 11)) if rng.uniform() < 0.5 else 0
        memo = draw_list() if rng.uniform() < 0.6 else None
        if b % 16 == 0:                                   # the stop index clamps at ego_index + 1 / at the table end
            conf = [min(i + 1, 84)] if b % 32 else [84, None]
        ag.is_collide, ag.conflict_index, ag.collision_memory = col, conf, mem
        ag.memorized_conflict_indices = memo
        ag.memorized_conflict_points = None if memo is None else [None] * len(memo)
        lv = int(rng.integers(0, 85)) if rng.uniform() < 0.4 else -1
        ag.last_valid_stop_point = None if lv < 0 else ag.reference_trajectory[lv]
        ag.stop_point = None
        rl = None if rng.uniform() < 0.7 else float(rng.uniform(-5.0, 40.0))
        r = _run_solve_head(ag, standins, None, None if rl is None else np.array([[rl]]))
        u_in["is_collide"][b], u_in["mem"][b], u_in["n_conf"][b] = int(col), mem, len(conf)
        u_in["conflict"][b, :len(conf)] = [-1 if c is None else c for c in conf]
        u_in["has_mem"][b] = 0 if memo is None else 1
        if memo is not None:
            u_in["n_mem"][b] = len(memo)
            u_in["memorized"][b, :len(memo)] = [-1 if c is None else c for c in memo]
        u_in["last_valid"][b] = lv
        u_in["rl"][b] = np.nan if rl is None else rl
        u_col[b], u_ego[b], u_stop[b] = r["ref"][:, 2], r["ego_index"], r["stop_index"]
        lvp = ag.last_valid_stop_point
        u_last[b] = -1 if lvp is None else int(np.argmin(np.linalg.norm(ag.reference_trajectory - lvp, axis=1)))
        assert np.array_equal(r["ref"][:, [0, 1, 3]], ref[:, [0, 1, 3]])
    out.update({f"update_ref_{k}": v for k, v in u_in.items()})
    out.update(update_ref_obs=u_obs, update_ref_speed_out=u_col, update_ref_ego_index=u_ego, update_ref_stop_out=u_stop,
               update_ref_last_valid_out=u_last)
    np.savez_compressed(os.path.join(HERE, "reference_random.npz"), **out)
    print("wrote reference_random.npz", {k: v.shape for k, v in out.items()})


def _record_closed_loop(n_env, n_others, steps, mode, seed):
    """Observation sequences of MPC-driven episodes on the synthetic environment (CPU: numpy mirror of the preamble + C
    oracle as the solver, as tests/golden/make_closed_loop.py does).  Returns obs [T, E, 10, 8], reset [T, E] (the
    environment began a new episode before this step), rl [T, E] (reference-speed override or NaN), and the oracle's
    solution (X, U) of every step."""
    import torch
    from mpc_rl_for_avs_amd import rollout
    from mpc_rl_for_avs_amd.reference_path import reference_states
    import oracle_lib
    from host_preamble import HostPreambleAgent
    REF = reference_states(0.1)
    sols = {}

    class OracleEngine:
        def solve_batch(self, state, ego_index, weights, is_collide, vref=None, others=None, collision_cost=False,
                        want_trajectories=False):
            r = oracle_lib.solve_batch(REF, state, ego_index, weights, is_collide, vref=vref, others=others,
                                       collision_cost=collision_cost, max_iter=200, xy_bounds=False, nthreads=8)
            sols["X"], sols["U"] = r["X"], r["U"]
            return r
    env = rollout.SyntheticIntersectionEnv(n_env, device="cpu", seed=seed, n_others=n_others)
    agent = HostPreambleAgent(_RefEnv, dict(_REF_CFG), engine=OracleEngine())
    pol = rollout.ActorCritic(1)
    gen = torch.Generator().manual_seed(seed)
    obs = env.reset()
    O, R, RL, XS, US = [], [], [], [], []
    reset = np.ones(n_env, bool)
    for _ in range(steps):
        o = obs.numpy().astype(np.float32)
        rs = None
        if mode == "v0":
            with torch.no_grad():
                a, _, _ = pol(obs, generator=gen)
            rs = torch.clamp(a, -1.0, 1.0)[:, :1].to(torch.float64).numpy()
        act = agent.predict_batch_host(o, None, rs)
        O.append(o)
        R.append(reset.copy())
        RL.append(np.full(n_env, np.nan) if rs is None else rs[:, 0].copy())
        XS.append(sols["X"].copy())
        US.append(sols["U"].copy())
        obs, _, done, _ = env.step(torch.as_tensor(act, dtype=torch.float64))
        reset = done.numpy().astype(bool)
        ids = np.nonzero(reset)[0]
        if ids.size:
            agent.reset_env_state([int(i) for i in ids])
    return np.array(O), np.array(R), np.array(RL), np.array(XS), np.array(US)


def reference_sequences():
    """Closed-loop observation sequences pushed through the REFERENCE's own `predict()` (agents/pure_mpc.py:68-78): its
    `_parse_obs`, its `_check_collision` state machine (LineString.intersection served by the geometry stand-in), and its
    `_solve` up to the nlpsol call, which yields the ego index, the rewritten reference table and the reference's own
    objective / constraint values at given points.  One reference agent per environment, constructed anew at episode
    boundaries (the reference keeps its detector state in the agent object, agents/pure_mpc.py:38-43)."""
    import contextlib
    import io
    parts = [_record_closed_loop(16, 4, 40, "mpc", seed=31), _record_closed_loop(16, 9, 40, "mpc", seed=32),
             _record_closed_loop(8, 4, 40, "v0", seed=33)]
    obs = np.concatenate([p[0] for p in parts], axis=1)
    reset = np.concatenate([p[1] for p in parts], axis=1)
    rl = np.concatenate([p[2] for p in parts], axis=1)
    Xs = np.concatenate([p[3] for p in parts], axis=1)
    Us = np.concatenate([p[4] for p in parts], axis=1)
    T, E = obs.shape[:2]
    rng = np.random.default_rng(77)
    # v1-style weights (RL action) on a third of the environments; negative values included (agents/ppo_mpc.py:407-417)
    w_env = np.full((E, 3), np.nan)
    w_env[::3] = rng.uniform(-1.0, 1.0, (len(range(0, E, 3)), 3))
    ag0, standins = _reference_agent()
    agents = [None] * E
    K = 9
    o = dict(state=np.zeros((T, E, 4)), nveh=np.zeros((T, E), np.int32), others=np.zeros((T, E, K, 4)),
             ego_index=np.zeros((T, E), np.int32), is_collide=np.zeros((T, E), np.uint8),
             collision_memory=np.zeros((T, E), np.int32), conflict_index=np.full((T, E, K), -1, np.int32),
             conflict_points=np.full((T, E, K, 2), np.nan), speed_col=np.zeros((T, E, 85)),
             stop_index=np.full((T, E), -1, np.int32))
    NP = 3
    nlp_sel = np.zeros((T, E), bool)
    nlp = dict(z=[], f=[], g=[], comp=[], t=[], e=[])
    first = {}
    from agents.pure_mpc import PureMPC_Agent
    for t in range(T):
        for e in range(E):
            if reset[t, e] or agents[e] is None:
                agents[e] = PureMPC_Agent(_RefEnv, dict(_REF_CFG))
            ag = agents[e]
            w = None if np.isnan(w_env[e, 0]) else w_env[e:e + 1].copy()
            rs = None if np.isnan(rl[t, e]) else np.array([[rl[t, e]]])
            with contextlib.redirect_stdout(io.StringIO()):
                ag._parse_obs(obs[t, e])
                ag._check_collision()
            take = rng.uniform() < 0.25
            X0 = np.tile([ag.ego_vehicle.position[0], ag.ego_vehicle.position[1], ag.ego_vehicle.heading,
                          ag.ego_vehicle.speed], (21, 1)).astype(np.float64)
            pts = [(X0, np.zeros((20, 2)))]
            if take:
                pts.append((Xs[t, e], Us[t, e]))            # the oracle's solution of this step (default weights)
                lo = np.array([-40.0, -10.0, -np.pi, 0.0])
                hi = np.array([10.0, 55.0, np.pi, 30.0])
                pts.append((rng.uniform(lo, hi, (21, 4)), rng.uniform([-5.0, -np.pi / 3], [5.0, np.pi / 3], (20, 2))))
            for q, pt in enumerate(pts):
                r = _run_solve_head(ag, standins, w, rs, pt)
                if q == 0:
                    first = r
                if take:
                    nlp["z"].append(np.concatenate([pt[0].ravel(), pt[1].ravel()]))
                    nlp["f"].append(r["f"])
                    nlp["g"].append(r["g"])
                    nlp["comp"].append(r["components"])
                    nlp["t"].append(t)
                    nlp["e"].append(e)
            nlp_sel[t, e] = take
            r = first
            ev = ag.ego_vehicle
            o["state"][t, e] = (ev.position[0], ev.position[1], ev.heading, ev.speed)
            o["nveh"][t, e] = len(ag.agent_vehicles)
            for j, v in enumerate(ag.agent_vehicles):
                o["others"][t, e, j] = (v.position[0], v.position[1], v.speed, v.heading)
            o["ego_index"][t, e] = r["ego_index"]
            o["is_collide"][t, e] = 1 if ag.is_collide else 0
            o["collision_memory"][t, e] = ag.collision_memory
            for j, c in enumerate(ag.conflict_index):
                o["conflict_index"][t, e, j] = -1 if c is None else int(c)
            for j, c in enumerate(ag.conflict_points):
                if c is not None:
                    o["conflict_points"][t, e, j] = np.asarray(c, np.float64)
            o["speed_col"][t, e] = r["ref"][:, 2]
            o["stop_index"][t, e] = r["stop_index"]
            if t == 0 and e == 0:
                o["x0"], o["lbx"], o["ubx"], o["lbg"], o["ubg"] = r["x0"], r["lbx"], r["ubx"], r["lbg"], r["ubg"]
                o["ipopt_max_iter"] = np.array(r["opts"]["ipopt.max_iter"])
                o["ipopt_tol"] = np.array(r["opts"]["ipopt.tol"])
            assert np.array_equal(r["x0"], np.concatenate([X0.ravel(), np.zeros(40)]))
    out = {f"seq_{k}": v for k, v in o.items()}
    out.update(seq_obs=obs, seq_reset=reset, seq_ref_speed=rl, seq_weights=w_env, nlp_z=np.array(nlp["z"]),
               nlp_f=np.array(nlp["f"]), nlp_g=np.array(nlp["g"]), nlp_components=np.array(nlp["comp"]),
               nlp_t=np.array(nlp["t"], np.int32), nlp_e=np.array(nlp["e"], np.int32))
    np.savez_compressed(os.path.join(HERE, "reference_sequences.npz"), **out)
    print("wrote reference_sequences.npz", {k: np.shape(v) for k, v in out.items()},
          "collide steps", int(o["is_collide"].sum()), "of", T * E)


def reference_predict_tail():
    """The reference's `predict()` executed END TO END (agents/pure_mpc.py:68-78, 80-318) on closed-loop observation
    sequences, with the solver stand-in handing back a given solution - the C oracle's of that step, flagged alternately as
    found / not found - so that the statements BEHIND the nlpsol call run too: sol['x'] unpacked into the control sequence,
    `last_acc`, `MPC_Action(...).numpy()`, the "NOTICE: Not found solution" print of a failed solve (the last iterate is
    used all the same).  Pins a1 (orchestration) and a12 (result) of SURVEY section 8."""
    import contextlib
    import io
    for d in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if d not in sys.path:
            sys.path.insert(0, d)
    obs, reset, rl, Xs, Us = _record_closed_loop(8, 4, 40, "v0", seed=41)
    T, E = obs.shape[:2]
    _, standins = _reference_agent()
    from agents.pure_mpc import PureMPC_Agent
    agents = [None] * E
    act = np.zeros((T, E, 2))
    last_acc = np.zeros((T, E))
    notice = np.zeros((T, E), bool)
    success = np.zeros((T, E), bool)
    try:
        for t in range(T):
            for e in range(E):
                if reset[t, e] or agents[e] is None:
                    agents[e] = PureMPC_Agent(_RefEnv, dict(_REF_CFG))
                ok = (t + e) % 3 != 0
                z = np.concatenate([Xs[t, e].ravel(), Us[t, e].ravel()])
                standins.SOLVE_HOOK[0] = lambda cap, z=z, ok=ok: (z, ok)
                standins.POINT.clear()      # the numeric SX needs a point to evaluate the (unused) cost at: the same one
                standins.POINT.update(x=Xs[t, e].T.copy(), u=Us[t, e].T.copy())
                standins.CAPTURED.clear()
                rs = None if np.isnan(rl[t, e]) else np.array([[rl[t, e]]])
                out = io.StringIO()
                with contextlib.redirect_stdout(out):
                    a = agents[e].predict(obs[t, e], weights_from_RL=None, ref_speed=rs)
                assert isinstance(a, np.ndarray) and a.shape == (2,)
                act[t, e] = a
                last_acc[t, e] = agents[e].last_acc
                notice[t, e] = "NOTICE: Not found solution" in out.getvalue()
                success[t, e] = ok
    finally:
        standins.SOLVE_HOOK[0] = None
    np.savez_compressed(os.path.join(HERE, "reference_predict_tail.npz"), obs=obs, reset=reset, ref_speed=rl, X=Xs, U=Us,
                        success=success, action=act, last_acc=last_acc, notice=notice)
    print("wrote reference_predict_tail.npz", act.shape, "failed solves", int((~success).sum()), "notices", int(notice.sum()))


def reference_distance_cost(n=512):
    """The "collision cost on" term.  Its only LIVE definition in the reference is the distance cost of the
    archived agent (agents/archive/pure_mpc.py:189-206: per stage and observed vehicle (d < 1 ? 1000 : 100) / (d + 1e-6)^2,
    the vehicles advanced by the constant-velocity model after every stage); the live file keeps the same statements in a
    dead branch (agents/pure_mpc.py:169-191).  The archived agent's own `_solve` is executed up to its nlpsol call with the
    numeric casadi stand-in; `Function('cost_fn', ...)` hands over its cost components, of which `distance_cost` is stored."""
    import contextlib
    import copy
    import io
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, HERE)
    import standins
    standins.install()
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from agents.archive.pure_mpc import PureMPC_Agent as ArchiveAgent
    from mpc_rl_for_avs_amd import synth
    cfg = dict(horizon=20, render=False, speed_override=0, weight_state=10, weight_control=1, weight_distance=10,
               weight_collision=1, weight_input_diff=1, weight_final_state=0)
    ag = ArchiveAgent(_RefEnv, cfg)
    rng = np.random.default_rng(808)
    obs = np.concatenate([synth.make_obs_batch(n // 4, V, seed=1300 + V) for V in (1, 4, 8, 9)])
    Z = np.zeros((n, 124))
    dist = np.zeros(n)
    comps = np.zeros((n, 6))
    for b in range(n):
        ag._parse_obs(obs[b])
        ag.is_collide = False
        e = ag.ego_vehicle
        X = np.tile([e.position[0], e.position[1], e.heading, e.speed], (21, 1)).astype(np.float64)
        X[:, :2] += np.cumsum(rng.normal(0.0, 0.8, (21, 2)), axis=0)          # a wandering path near the ego
        if b % 3 == 0 and ag.agent_vehicles:                                   # some nodes within 1 m of a vehicle
            v = ag.agent_vehicles[int(rng.integers(0, len(ag.agent_vehicles)))]
            k = int(rng.integers(0, 20))
            pk = np.asarray(v.position, np.float64) + k * float(v.speed) * 0.1 * np.array([np.cos(float(v.heading)),
                                                                                          np.sin(float(v.heading))])
            X[k, :2] = pk + rng.uniform(-0.9, 0.9, 2)
        U = rng.uniform([-5.0, -1.0], [5.0, 1.0], (20, 2))
        standins.POINT.clear()
        standins.POINT.update(x=X.T.copy(), u=U.T.copy())
        standins.CAPTURED.clear()
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ag._solve(None)
            raise AssertionError("the solver stand-in must end _solve")
        except standins.Captured:
            pass
        comps[b] = standins.CAPTURED["functions"]["cost_fn"]      # state, control, final_state, input_diff, distance, collision
        dist[b] = comps[b, 4]
        Z[b] = np.concatenate([X.ravel(), U.ravel()])
    out = dict(obs=obs, z=Z, distance_cost=dist, components=comps)
    np.savez_compressed(os.path.join(HERE, "reference_distance_cost.npz"), **out)
    print("wrote reference_distance_cost.npz", {k: v.shape for k, v in out.items()},
          "cases with a node inside 1 m:", int((dist > 2000).sum()))


def ltv_reference_vectors():
    ref_root = "/root/reference"
    for name in ("gymnasium", "cvxpy", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["gymnasium"].Env = object
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    from agents import pure_mpc_linear as RL  # noqa: E402
    from mpc_rl_for_avs_amd import synth
    from mpc_rl_for_avs_amd.reference_path import reference_states
    ref = reference_states()
    rng = np.random.default_rng(321)
    out = {"constants": np.array([RL.MAX_STEER, RL.MAX_DSTEER, RL.MAX_ACCEL, RL.MAX_DECEL, RL.MAX_SPEED,
                                  RL.R[0, 0], RL.R[1, 1], RL.Rd[0, 0], RL.Rd[1, 1], RL.Q_v_yaw[0, 0], RL.Q_v_yaw[1, 1],
                                  RL.Qf[0, 0], RL.Qf[1, 1], RL.Qf[2, 2], RL.Qf[3, 3]])}
    # nearest index: synthetic ego positions (float32 like parsed observations) + exact ties between two path points
    inp = synth.solver_inputs(40, 2, seed=9)
    pos = inp["state"][:, :2].astype(np.float32).astype(np.float64)
    pos = np.concatenate([pos, [[2.0, 30.5], [-20.5, -2.22585], [100.0, 100.0], [-60.0, 0.0]]])
    out["nearest_in"] = pos
    out["nearest_out"] = np.array([RL.calc_nearest_index_in_direction(p[0], p[1], ref[:, 0], ref[:, 1], 0) for p in pos])
    # linear model at random operating points
    vb = rng.uniform(0, 11, 16)
    yb = rng.uniform(-3.5, 3.5, 16)
    A, Bm = [], []
    for v, y in zip(vb, yb):
        a, b, c = RL.linear_model_matrix(v, y, 0.0, 0.1, 2.5)
        assert not c.any()
        A.append(a)
        Bm.append(b)
    out["linmodel_in"] = np.stack([vb, yb], axis=1)
    out["linmodel_A"] = np.stack(A)
    out["linmodel_B"] = np.stack(Bm)
    # nominal rollouts: zero profile, random profiles (speed clamp active at both ends)
    x0s, oas, ods, xbars = [], [], [], []
    for i in range(12):
        T = 20
        x0 = np.array([rng.uniform(-30, 5), rng.uniform(-5, 50), rng.uniform(0, 11), rng.uniform(-3.1, 3.1)])
        if i == 0:
            oa, od = np.zeros(T), np.zeros(T)
        else:
            oa = rng.uniform(-5, 2, T) * (3.0 if i % 3 == 0 else 1.0)
            od = rng.uniform(-0.52, 0.52, T)
        x0s.append(x0)
        oas.append(oa)
        ods.append(od)
        xbars.append(RL.predict_motion(x0, oa, od, 0.1, 2.5).T)      # [T+1, 4]
    out["nominal_x0"] = np.stack(x0s)
    out["nominal_oa"] = np.stack(oas)
    out["nominal_od"] = np.stack(ods)
    out["nominal_xbar"] = np.stack(xbars)
    np.savez_compressed(os.path.join(HERE, "ltv_reference_numpy.npz"), **out)
    print("wrote ltv_reference_numpy.npz", {k: v.shape for k, v in out.items()})


def ltv_reference_random_vectors(n=512):
    """Round 4: >= 500 randomised cases for each numpy-only helper of the reference's iterative-linear agent
    (agents/pure_mpc_linear.py: calc_nearest_index_in_direction :38-60, linear_model_matrix :62-82, predict_motion
    :84-110), the per-stage models along nominal rollouts (what _linear_mpc_control linearises about, :222-228), and the
    agent's own `_solve` / `_linear_mpc_control` statements (:153-257) evaluated numerically at given (x, u) with the cvxpy
    stand-in of tests/golden/standins.py: objective value, equality residuals, inequality slacks."""
    import contextlib
    import io
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, HERE)
    import standins
    standins.install()
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from agents import pure_mpc_linear as RL
    from mpc_rl_for_avs_amd import synth
    from mpc_rl_for_avs_amd.reference_path import reference_states
    ref = reference_states()
    rng = np.random.default_rng(4321)
    out = {}
    # nearest index: parsed-observation positions, ties, far-away points
    inp = synth.solver_inputs(n, 2, seed=19)
    pos = inp["state"][:, :2].astype(np.float32).astype(np.float64)
    pos[::17] += rng.uniform(-40.0, 40.0, (len(pos[::17]), 2))
    pos[:3] = [[2.0, 30.5], [-20.5, float(ref[84, 1])], [2.0, 9.5]]           # exact ties between two path points
    out["nearest_in"] = pos
    out["nearest_out"] = np.array([RL.calc_nearest_index_in_direction(p[0], p[1], ref[:, 0], ref[:, 1], 0) for p in pos])
    # linear model at random operating points
    vb, yb = rng.uniform(0.0, 12.0, n), rng.uniform(-3.6, 3.6, n)
    AB = [RL.linear_model_matrix(v, y, 0.0, 0.1, 2.5) for v, y in zip(vb, yb)]
    assert not any(c.any() for _, _, c in AB)
    out.update(linmodel_in=np.stack([vb, yb], axis=1), linmodel_A=np.stack([a for a, _, _ in AB]),
               linmodel_B=np.stack([b for _, b, _ in AB]))
    # nominal rollouts and the stage models along them
    T = 20
    x0 = np.stack([rng.uniform(-35.0, 5.0, n), rng.uniform(-5.0, 50.0, n), rng.uniform(0.0, 11.0, n),
                   rng.uniform(-3.14, 3.14, n)], axis=1)
    x0 = x0.astype(np.float32).astype(np.float64)               # what a parsed observation can hold
    oa = rng.uniform(-5.0, 2.0, (n, T)) * np.where(np.arange(n) % 3 == 0, 3.0, 1.0)[:, None]
    od = rng.uniform(-0.52, 0.52, (n, T))
    oa[::8], od[::8] = 0.0, 0.0                                   # first calls: zero profile
    xbar = np.stack([RL.predict_motion(x0[i], oa[i], od[i], 0.1, 2.5).T for i in range(n)])       # [n, T+1, 4]
    out.update(nominal_x0=x0, nominal_oa=oa, nominal_od=od, nominal_xbar=xbar)
    ns = 128
    out["stage_A"] = np.stack([[RL.linear_model_matrix(xbar[i, t, 2], xbar[i, t, 3], 0.0, 0.1, 2.5)[0] for t in range(T)]
                               for i in range(ns)])
    out["stage_B"] = np.stack([[RL.linear_model_matrix(xbar[i, t, 2], xbar[i, t, 3], 0.0, 0.1, 2.5)[1] for t in range(T)]
                               for i in range(ns)])
    # the agent's own QP statements at given points
    ag = RL.IterativeLinearMPC_Agent(_RefEnv, dict(horizon=T, render=False))
    nq = 256
    q_obs = synth.make_obs_batch(nq, 0, seed=77)
    q_u = rng.uniform([-6.0, -0.6], [3.0, 0.6], (nq, T, 2))
    q_x = np.zeros((nq, T + 1, 4))
    q_cost, q_eq, q_le, q_xref, q_target = np.zeros(nq), [], [], np.zeros((nq, T + 1, 4)), np.zeros(nq, np.int32)
    q_oa, q_od = oa[:nq].copy(), od[:nq].copy()
    for i in range(nq):
        ag._parse_obs(q_obs[i])
        first = i % 8 == 0
        ag.oa, ag.od = (None, None) if first else (q_oa[i].copy(), q_od[i].copy())
        e = ag.ego_vehicle
        xs = np.array([e.position[0], e.position[1], e.speed, e.heading], dtype=float)
        xb = RL.predict_motion(xs, np.zeros(T) if first else q_oa[i], np.zeros(T) if first else q_od[i], 0.1, 2.5)
        x = np.zeros((4, T + 1))
        x[:, 0] = xs
        for t in range(T):                                          # a point ON the linear dynamics: residuals vanish
            A, B, C = RL.linear_model_matrix(xb[2, t], xb[3, t], 0.0, 0.1, 2.5)
            x[:, t + 1] = A @ x[:, t] + B @ q_u[i, t] + C
        if i % 5 == 4:
            x[:, 1:] += rng.normal(0.0, 0.3, (4, T))                # ... and some off them
        standins.POINT["cvx"] = [x.copy(), q_u[i].T.copy()]
        standins.CAPTURED.clear()
        got = {}
        orig = ag._linear_mpc_control

        def spy(xref, xbar_, x0_):
            got["xref"], got["xbar"] = np.array(xref), np.array(xbar_)
            return orig(xref, xbar_, x0_)
        ag._linear_mpc_control = spy
        with contextlib.redirect_stdout(io.StringIO()):
            act = ag._solve()
        ag._linear_mpc_control = orig
        assert act.acceleration == 0.0 and act.steer == 0.0                      # the reference's "solver failed" exit
        assert np.array_equal(got["xbar"], xb)
        c = standins.CAPTURED
        q_x[i], q_cost[i], q_xref[i], q_target[i] = x.T, c["cvx_cost"], got["xref"].T, ag.target_ind
        q_eq.append(c["cvx_eq"])
        q_le.append(c["cvx_le"])
        if first:
            q_oa[i], q_od[i] = 0.0, 0.0
    out.update(qp_obs=q_obs, qp_oa=q_oa, qp_od=q_od, qp_x=q_x, qp_u=q_u, qp_cost=q_cost, qp_eq=np.array(q_eq),
               qp_le=np.array(q_le), qp_xref=q_xref, qp_target=q_target)
    np.savez_compressed(os.path.join(HERE, "ltv_reference_random.npz"), **out)
    print("wrote ltv_reference_random.npz", {k: v.shape for k, v in out.items()})


def ltv_oracle_vectors():
    import ltv_oracle as L
    from scipy.optimize import minimize
    from mpc_rl_for_avs_amd import synth
    from mpc_rl_for_avs_amd.reference_path import reference_states
    ref = reference_states()
    inp = synth.solver_inputs(48, 2, seed=303)
    st = np.ascontiguousarray(inp["state"][:, [0, 1, 3, 2]].astype(np.float32).astype(np.float64))
    nom = np.zeros((48, 20, 2))
    a = L.solve_batch(ref, st, nom)
    b = L.solve_batch(ref, st, a["U"])
    out = dict(state=st, U0=nom, u0_first=a["u0"], U_first=a["U"], status_first=a["status"], iters_first=a["iters"],
               target_index=a["target_index"], u0_second=b["u0"], U_second=b["U"], status_second=b["status"])
    fs = np.full(6, np.nan)
    fo = np.full(6, np.nan)
    for i in range(6):
        if a["status"][i] != 0:
            continue
        f = lambda u: L.objective_loops(u.reshape(20, 2), st[i], a["xref"][i], a["xbar"][i], 0.1)
        cons = {"type": "ineq", "fun": lambda u: L.constraint_loops(u.reshape(20, 2), st[i], a["xbar"][i], 0.1)}
        r = minimize(f, np.full(40, 0.05), constraints=[cons], method="SLSQP", options=dict(ftol=1e-15, maxiter=800))
        fs[i], fo[i] = r.fun, f(a["U"][i].ravel())
    out["slsqp_objective"], out["oracle_objective"] = fs, fo
    print("ltv oracle: status", np.bincount(a["status"], minlength=4), np.bincount(b["status"], minlength=4),
          "objective oracle - slsqp", np.nanmax(fo - fs))
    np.savez_compressed(os.path.join(HERE, "ltv_oracle_solutions.npz"), **out)


def oracle_vectors():
    import oracle_lib
    import nlp_batch as nb
    import kkt_batch as kb
    from mpc_rl_for_avs_amd import synth
    from mpc_rl_for_avs_amd.reference_path import reference_states
    ref = reference_states()
    out = {}
    for name, V, cc, seed in (("cfg2", 4, False, 101), ("cfg3", 8, True, 202)):
        inp = synth.solver_inputs(32, V, seed=seed)
        sol = oracle_lib.solve_batch(ref, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                                     vref=inp["vref"], others=inp["others"], collision_cost=cc, max_iter=100,
                                     xy_bounds=False)       # what the engine solves (the |x|, |y| <= 500 bounds never bind)
        p = nb.Batch.build(ref, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                           others=inp["others"], collision_cost=cc)
        cert = kb.certify(p, sol["X"], sol["U"])
        conv = (sol["status"] == 0) | (sol["status"] == 5)
        stat = np.where(conv, cert["stationarity"], np.nan)
        for k in ("state", "ego_index", "vref", "weights", "is_collide", "others"):
            out[f"{name}_{k}"] = inp[k]
        for k in ("u0", "U", "X", "status", "iters"):
            out[f"{name}_{k}"] = sol[k]
        out[f"{name}_kkt_rel_stationarity"] = stat
        print(name, "status", np.bincount(sol["status"], minlength=6), "worst certified rel stationarity", np.nanmax(stat))
    np.savez_compressed(os.path.join(HERE, "oracle_solutions.npz"), **out)


if __name__ == "__main__":
    if "--predict-tail-only" in sys.argv:
        reference_predict_tail()
        sys.exit(0)
    if "--round4-only" in sys.argv:
        reference_predict_tail()
        reference_random_vectors()
        reference_sequences()
        reference_distance_cost()
        ltv_reference_random_vectors()
        sys.exit(0)
    if "--oracle-only" not in sys.argv:      # the reference's own numpy code (needs /root/reference)
        reference_vectors()
        ltv_reference_vectors()
        reference_random_vectors()
        reference_sequences()
        reference_distance_cost()
        reference_predict_tail()
        ltv_reference_random_vectors()
    if "--reference-only" not in sys.argv:
        oracle_vectors()
        ltv_oracle_vectors()
