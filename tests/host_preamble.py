"""TEST INFRASTRUCTURE - numpy mirror of the reference's observation preamble (never imported by the product).

What `PureMPC_Agent.predict()` does before its NLP solve - observation parsing, the path-crossing "collision" detector
with its 10-step memory and the rewrite of the reference speed profile (reference agents/pure_mpc.py:552-724,
agents/base_agent.py:81-116) - restated statement by statement in numpy with the reference's float32 / float64
promotion behaviour.  It is pinned by outputs of the reference's own functions (tests/golden/reference_numpy.npz,
tests/test_host.py) and is the checker of the device preamble (csrc/mpc_preamble.hpp: tests/test_preamble_cpu.py on the
host build, tests/test_predict_gpu.py on the GPU).  Up to round 1 this code lived in the product package and served
`predict()`; the product now runs one preamble only, the device one (`mpc_predict_batch`).

`HostPreambleAgent(env, cfg, engine)` keeps the reference's call sequence `predict = _parse_obs -> _check_collision ->
_solve`, the solve going to `engine.solve_batch` (a real MPCEngine or a recording fake)."""
from __future__ import annotations

import numpy as np

from mpc_rl_for_avs_amd.pure_mpc import (COLLISION_MEMORY_STEPS, DEFAULT_MAX_SPEED, PREDICTION_HORIZON, SAFETY_BUFFER_POINTS,
                                         TIME_THRESHOLD, MPC_Action, Vehicle, normalize_angle)
from mpc_rl_for_avs_amd.reference_path import reference_states as _reference_states


# ------------------------------------------------------------------------------------------------------
# geometry that the reference delegates to shapely (agents/pure_mpc.py:583,608-633)
# ------------------------------------------------------------------------------------------------------
def _seg_intersections(p, q, a, b, eps=1e-12):
    """Intersection of segments p-q and a-b: list of (t_on_pq, point); two entries = collinear overlap ends."""
    r = q - p
    s = b - a
    rxs = r[0] * s[1] - r[1] * s[0]
    ap = a - p
    scale = max(1.0, float(np.abs(r).max()), float(np.abs(s).max())) ** 2
    if abs(rxs) > eps * scale:
        t = (ap[0] * s[1] - ap[1] * s[0]) / rxs
        u = (ap[0] * r[1] - ap[1] * r[0]) / rxs
        if -1e-12 <= t <= 1 + 1e-12 and -1e-12 <= u <= 1 + 1e-12:
            return [(min(max(t, 0.0), 1.0), p + min(max(t, 0.0), 1.0) * r)]
        return []
    if abs(ap[0] * r[1] - ap[1] * r[0]) > eps * scale:
        return []                          # parallel, not collinear
    rr = float(r @ r)
    if rr == 0.0:                          # p-q is a point
        ss = float(s @ s)
        if ss == 0.0:
            return [(0.0, p.copy())] if np.allclose(p, a) else []
        u = float((p - a) @ s) / ss
        return [(0.0, p.copy())] if -1e-12 <= u <= 1 + 1e-12 else []
    t0 = float(ap @ r) / rr
    t1 = float((b - p) @ r) / rr
    lo, hi = max(0.0, min(t0, t1)), min(1.0, max(t0, t1))
    if lo > hi:
        return []
    if hi - lo < 1e-15:
        return [(lo, p + lo * r)]
    return [(lo, p + lo * r), (hi, p + hi * r)]


def path_pieces(ego_path, agent_path, max_pieces=4):
    """The intersection of the two polylines as a list of pieces along the ego's direction of travel: ("point", p) for a
    transversal crossing, ("line", [p0, p1, ...]) for a collinear overlap with every node of the overlapping stretch (its
    ends on each ego segment it covers and the agent's own vertices inside it), de-duplicated and ordered along the ego."""
    ego = np.asarray(ego_path, dtype=np.float64)
    ag = np.asarray(agent_path, dtype=np.float64)
    out = []
    if len(ego) < 2 or len(ag) < 2:
        return out
    a, b = ag[0], ag[-1]                   # constant-velocity prediction: the agent polyline is one straight segment
    i = 0
    while i < len(ego) - 1 and len(out) < max_pieces:
        hits = _seg_intersections(ego[i], ego[i + 1], a, b)
        if not hits:
            i += 1
            continue
        if len(hits) == 1:
            pt = np.asarray(hits[0][1], dtype=np.float64)
            same = lambda q: abs(q[0] - pt[0]) <= 1e-12 + 1e-5 * abs(pt[0]) and abs(q[1] - pt[1]) <= 1e-12 + 1e-5 * abs(pt[1])
            # a crossing exactly at an ego vertex is found by both segments that share it
            dup = bool(out) and out[-1][0] == "point" and same(out[-1][1])
            # the intersection is a point SET: where the ego joins or leaves the other path's line at a vertex, the segment
            # before / behind the collinear stretch touches the line in the stretch's end point, which is part of that piece
            if not dup and out and out[-1][0] == "line" and same(out[-1][1][-1]):
                dup = True
            if not dup and i + 2 < len(ego):
                nxt = _seg_intersections(ego[i + 1], ego[i + 2], a, b)
                dup = len(nxt) == 2 and same(nxt[0][1])
            if not dup:
                out.append(("point", pt))
            i += 1
            continue
        # collinear overlap starting on this ego segment: collect the overlapping stretch over following segments
        pts = [hits[0][1], hits[1][1]]
        j = i + 1
        while j < len(ego) - 1:
            h2 = _seg_intersections(ego[j], ego[j + 1], a, b)
            if len(h2) != 2:
                break
            pts.append(h2[1][1])
            j += 1
        s = b - a
        ss = float(s @ s)
        k0, k1 = float((pts[0] - a) @ s), float((pts[-1] - a) @ s)
        for v in ag:                        # the agent's own vertices inside the overlap are nodes too
            if ss > 0 and min(k0, k1) - 1e-12 <= float((v - a) @ s) <= max(k0, k1) + 1e-12:
                pts.append(v)
        d = ego[i + 1] - ego[i]
        keyed = sorted(((float((pt - ego[i]) @ d), tuple(pt)) for pt in pts))
        uniq = []
        for _, pt in keyed:
            if not uniq or not np.allclose(uniq[-1], pt, atol=1e-12):
                uniq.append(pt)
        out.append(("line", [np.array(u, dtype=np.float64) for u in uniq]))
        i = j                               # go on behind the overlap
    return out


def path_crossings(ego_path, agent_path, max_candidates=4):
    """Intersection points of two polylines as the candidate list the reference builds from shapely's result
    (agents/pure_mpc.py:615-633): every transversal crossing, and for a collinear overlap the middle vertex of the
    overlapping stretch (`coords[len(coords) // 2]`).  ORDER: along the ego's direction of travel - GEOS returns the
    members of a Multi* geometry in the iteration order of a hash map of its overlay graph, which cannot be reproduced; a
    straight agent path meets the ego path more than once only when it cuts the arc twice."""
    return [p if kind == "point" else p[len(p) // 2] for kind, p in path_pieces(ego_path, agent_path, max_candidates)]


def first_path_crossing(ego_path, agent_path):
    """The first candidate of `path_crossings` (the only one in all but double-crossing scenes); None if none."""
    c = path_crossings(ego_path, agent_path, 1)
    return c[0] if c else None


class _EnvState:
    """Per-environment persistent state of the collision detector (agents/pure_mpc.py:38-43,63)."""
    __slots__ = ("collision_memory", "memorized_conflict_points", "memorized_conflict_indices",
                 "last_valid_stop_point", "stop_point", "is_collide", "conflict_points", "conflict_index",
                 "ego_index", "last_acc")

    def __init__(self):
        self.collision_memory = 0
        self.memorized_conflict_points = None
        self.memorized_conflict_indices = None
        self.last_valid_stop_point = None
        self.stop_point = None
        self.is_collide = False
        self.conflict_points = []
        self.conflict_index = []
        self.ego_index = 0
        self.last_acc = 0


class HostPreambleAgent:
    """The reference agent with its preamble in numpy on the host (checker of the device preamble)."""

    weight_components = ["speed", "control", "input_diff"]   # agents/pure_mpc.py:15-22

    def __init__(self, env, cfg: dict, engine=None, collision_cost: bool = False,
                 device: int = 0, max_iter: int = 100, tol: float = 1e-8, warm_start: bool = False) -> None:
        # agents/base_agent.py:28-49
        self.env = env.unwrapped if hasattr(env, "unwrapped") else env
        self.env_config = self.env.config
        self.config = cfg
        self.simulate_freq = self.env_config["simulation_frequency"]
        self.policy_freq = self.env_config["policy_frequency"]
        self.total_vehicles_count = self.env_config["observation"]["vehicles_count"]
        self.observed_vehicles_count = 0
        self.ego_vehicle = None
        self.agent_vehicles = list()
        self.horizon = self.config["horizon"]
        self.dt = 1 / self.policy_freq
        self.global_reference_states = self.reference_states
        self.reference_trajectory = self.global_reference_states[:, :2]
        self.render = self.config.get("render", False)
        self.num_frames_in_dt = self.simulate_freq // self.policy_freq
        # agents/pure_mpc.py:38-63
        self.collision_memory_steps = COLLISION_MEMORY_STEPS
        self.ttc_threshold = self.config.get("ttc_threshold", 3)
        self.default_weights = {f"weight_{k}": self.config[f"weight_{k}"] for k in HostPreambleAgent.weight_components}
        self.collision_cost = bool(collision_cost)
        self.warm_start = bool(warm_start)       # predict_batch only; the reference always starts cold
        if engine is None:
            from mpc_rl_for_avs_amd.engine import MPCEngine
            engine = MPCEngine(horizon=self.horizon, dt=self.dt, max_iter=max_iter, tol=tol,
                               w_distance=float(self.config.get("weight_distance", 10.0)),
                               w_collision=float(self.config.get("weight_collision", 1.0)), device=device,
                               ref_table=self.global_reference_states)
        self._engine = engine
        self._states = [_EnvState()]
        self.last_solve = None

    def __str__(self) -> str:
        return "Pure MPC agent [Receding Horizon Control], solved by the MI355X batched engine"

    # ---- single-environment attributes of the reference, served from environment 0
    def _s0(self):
        return self._states[0]

    is_collide = property(lambda self: self._s0().is_collide)
    conflict_points = property(lambda self: self._s0().conflict_points)
    conflict_index = property(lambda self: self._s0().conflict_index)
    ego_index = property(lambda self: self._s0().ego_index)
    stop_point = property(lambda self: self._s0().stop_point)
    last_acc = property(lambda self: self._s0().last_acc)
    collision_memory = property(lambda self: self._s0().collision_memory)

    @property
    def reference_states(self):
        return _reference_states(self.dt)          # agents/base_agent.py:118-154 (fresh copy per access)

    normalize_angle = staticmethod(normalize_angle)

    def other_vehicle_model(self, other_vehicle, dt):      # agents/base_agent.py:172-174
        return other_vehicle.position + other_vehicle.speed * dt * np.array(
            [np.cos(other_vehicle.heading), np.sin(other_vehicle.heading)])

    def reset_env_state(self, env_ids=None):
        """Forget the collision memory of the given environments (episode boundaries), in the engine
        (`predict_batch`) and in the host-side states (`predict`, `predict_batch_host`)."""
        ids = range(len(self._states)) if env_ids is None else env_ids
        for i in ids:
            if i < len(self._states):
                self._states[i] = _EnvState()
        if hasattr(self._engine, "reset_env_state"):
            self._engine.reset_env_state(env_ids)

    # ------------------------------------------------------------------ reference API
    def predict(self, obs, return_numpy=True, weights_from_RL=None, ref_speed=None):
        """agents/pure_mpc.py:68-78"""
        self._parse_obs(obs)
        self._check_collision()
        mpc_action = self._solve(weights_from_RL, ref_speed)
        return mpc_action.numpy() if return_numpy else mpc_action

    def _parse_obs(self, obs: np.ndarray) -> None:
        """agents/base_agent.py:81-116"""
        if not isinstance(obs, np.ndarray):
            raise TypeError(f"Expect observation type np.ndarray, but got {type(obs)}.")
        if obs.shape != (self.total_vehicles_count, 8):
            raise ValueError(
                f"Expect observation's shape of ({(self.total_vehicles_count, 8)}), but got {obs.shape}")
        self.ego_vehicle, self.agent_vehicles = self._vehicles_from_obs(obs)
        self.observed_vehicles_count = len(self.agent_vehicles)
        self.agent_vehicles_mpc = [Vehicle(v.index, v.position.copy(), v.vectorized_speed, v.heading, v.sinh, v.cosh)
                                   for v in self.agent_vehicles]

    def _vehicles_from_obs(self, obs):
        observed = int(np.sum(obs[:, 0] == 1)) - 1
        ego = Vehicle(0, obs[0, 1:3], obs[0, 3:5], normalize_angle(obs[0, 5]), obs[0, 6], obs[0, 7])
        others = [Vehicle(i + 1, obs[i + 1, 1:3], obs[i + 1, 3:5], obs[i + 1, 5], obs[0, 6], obs[0, 7])
                  for i in range(max(observed, 0))]
        return ego, others

    def _check_collision(self):
        self._check_collision_env(self._states[0], self.ego_vehicle, self.agent_vehicles)

    def _solve(self, weights_from_RL=None, ref_speed_from_RL=None) -> MPC_Action:
        """agents/pure_mpc.py:80-318 for the single environment parsed by `_parse_obs`."""
        w = None if weights_from_RL is None else np.asarray(weights_from_RL, dtype=np.float64).reshape(1, -1)[:, :3]
        rs = None if ref_speed_from_RL is None else np.asarray(ref_speed_from_RL, dtype=np.float64).reshape(1, 1)
        act = self._solve_envs([self._states[0]], [self.ego_vehicle], [self.agent_vehicles], w, rs)
        self._states[0].last_acc = act[0, 0]
        return MPC_Action(acceleration=act[0, 0], steer=act[0, 1])

    # ------------------------------------------------------------------ batched API
    def predict_batch(self, obs, weights_from_RL=None, ref_speed=None) -> np.ndarray:
        """B parallel environments: obs[B, vehicles_count, 8] -> actions[B, 2] (acceleration, steer).

        `weights_from_RL` is [B, 3] (speed, control, input_diff) or None, `ref_speed` [B, 1] or None;
        environment b keeps its own collision memory across calls (same B every call, `reset_env_state` at
        episode ends).  Equivalent to looping `predict` over the environments; everything runs on the device
        (`mpc_predict_batch`)."""
        obs, w, rs = self._check_batch_args(obs, weights_from_RL, ref_speed)
        B = obs.shape[0]
        if w is None:
            w = np.tile([float(self.default_weights[f"weight_{k}"]) for k in HostPreambleAgent.weight_components], (B, 1))
        out = self._engine.predict_batch(obs, w, None if rs is None else rs[:, 0], collision_cost=self.collision_cost,
                                         warm_start=self.warm_start)
        self.last_solve = out
        bad = int(np.count_nonzero(out["status"]))
        if bad:                                             # agents/pure_mpc.py:303-305
            print(f"NOTICE: Not found solution ({bad} of {B} instances)")
        return out["act"]

    def batch_env_state(self, B):
        """is_collide / ego_index / collision_memory / stop_index / conflict_index of environments 0..B-1 after
        `predict_batch` (the batched counterpart of the attributes `predict` leaves on the agent)."""
        return self._engine.env_state(B)

    def predict_batch_host(self, obs, weights_from_RL=None, ref_speed=None) -> np.ndarray:
        """`predict` looped over the environments with one batched solve: the preamble of every environment runs in
        numpy on the host (the single-environment code path), the solve on the device.  Kept as the cross-check of
        `predict_batch`; it has its own per-environment states, separate from the engine's."""
        obs, w, rs = self._check_batch_args(obs, weights_from_RL, ref_speed)
        B = obs.shape[0]
        while len(self._states) < B:
            self._states.append(_EnvState())
        egos, others = [], []
        for b in range(B):
            e, o = self._vehicles_from_obs(obs[b])
            self._check_collision_env(self._states[b], e, o)
            egos.append(e)
            others.append(o)
        act = self._solve_envs(self._states[:B], egos, others, w, rs)
        for b in range(B):
            self._states[b].last_acc = act[b, 0]
        return act

    def _check_batch_args(self, obs, weights_from_RL, ref_speed):
        if not isinstance(obs, np.ndarray):
            raise TypeError(f"Expect observation type np.ndarray, but got {type(obs)}.")
        if obs.ndim != 3 or obs.shape[1:] != (self.total_vehicles_count, 8):
            raise ValueError(f"Expect observations of shape (B, {self.total_vehicles_count}, 8), but got {obs.shape}")
        B = obs.shape[0]
        w = None if weights_from_RL is None else np.asarray(weights_from_RL, dtype=np.float64).reshape(B, -1)[:, :3]
        rs = None if ref_speed is None else np.asarray(ref_speed, dtype=np.float64).reshape(B, 1)
        return obs, w, rs

    # ------------------------------------------------------------------ preamble pieces
    def _nearest_ref_index(self, position):
        d = self.reference_trajectory - np.asarray(position)[None, :]     # float64 - float32 -> float64
        return int(np.argmin(np.sqrt(np.sum(d * d, axis=1))))

    def predict_ego_future_positions(self, current_position, speed, heading, max_acceleration, dt,
                                     prediction_horizon, reference_speed):
        """agents/pure_mpc.py:459-527: walk along the reference path by arc length."""
        future_positions = [current_position]
        current_speed = speed
        start_index = self._nearest_ref_index(current_position)
        ref_points = self.reference_trajectory[start_index:, :2]
        if len(ref_points) < 2:
            return future_positions
        seg = np.linalg.norm(ref_points[1:] - ref_points[:-1], axis=1)
        cumulative = np.concatenate([[0.0], np.cumsum(seg)])
        current_distance = 0
        for _ in range(prediction_horizon):
            if current_speed < reference_speed:
                current_speed = min(current_speed + max_acceleration * dt, reference_speed)
            else:
                current_speed = reference_speed
            current_distance += current_speed * dt
            next_idx = int(np.searchsorted(cumulative, current_distance))
            if next_idx >= len(ref_points):
                break
            if next_idx == 0:
                next_position = ref_points[0]
            else:
                prev_dist, next_dist = cumulative[next_idx - 1], cumulative[next_idx]
                alpha = (current_distance - prev_dist) / (next_dist - prev_dist) if next_dist != prev_dist else 1.0
                alpha = np.clip(alpha, 0, 1)
                next_position = ref_points[next_idx - 1] + alpha * (ref_points[next_idx] - ref_points[next_idx - 1])
            future_positions.append(next_position)
        if len(future_positions) <= 1:
            return [current_position] * prediction_horizon
        return future_positions

    def predict_future_positions(self, current_position, speed, heading, dt, prediction_horizon):
        """agents/pure_mpc.py:529-550: constant-velocity prediction."""
        future_positions = [current_position]
        step = speed * dt * np.array([np.cos(heading), np.sin(heading)])
        for _ in range(prediction_horizon):
            future_positions.append(future_positions[-1] + step)
        return future_positions

    def _check_collision_env(self, st: _EnvState, ego: Vehicle, agents):
        """agents/pure_mpc.py:552-676 for one environment."""
        if st.collision_memory > 0 and st.memorized_conflict_points is not None:
            st.conflict_points = st.memorized_conflict_points
            st.conflict_index = st.memorized_conflict_indices
            st.is_collide = True
            st.collision_memory -= 1
            return
        st.ego_index = self._nearest_ref_index(ego.position)
        ego_future = self.predict_ego_future_positions(
            current_position=ego.position, speed=ego.speed, heading=ego.heading,
            max_acceleration=ego.max_acceleration, dt=self.dt, prediction_horizon=PREDICTION_HORIZON,
            reference_speed=self.global_reference_states[st.ego_index, 2])
        ego_arr = np.asarray([np.asarray(p, dtype=np.float64) for p in ego_future])
        if len(ego_arr) < 2:
            # the ego stands on the last reference point: LineString of ONE point raises GEOSException, the reference prints a
            # warning and returns with the detector state as it was (agents/pure_mpc.py:582-587)
            return
        st.conflict_points, st.conflict_index = [], []
        collide = []
        for veh in agents:
            agent_future = self.predict_future_positions(np.array(veh.position), veh.speed, veh.heading, self.dt,
                                                         PREDICTION_HORIZON)
            ag_arr = np.asarray(agent_future, dtype=np.float64)
            detected, conflict_idx, point = False, None, None
            for point in path_crossings(ego_arr, ag_arr):            # candidate loop of agents/pure_mpc.py:635-654
                ego_time = int(np.argmin(np.linalg.norm(ego_arr - point, axis=1)))
                agent_time = int(np.argmin(np.linalg.norm(ag_arr - point, axis=1)))
                if abs(ego_time - agent_time) < TIME_THRESHOLD:
                    detected = True
                    conflict_idx = int(np.argmin(np.linalg.norm(self.reference_trajectory - point, axis=1)))
                    break
            collide.append(detected)
            st.conflict_points.append(point if detected else None)
            st.conflict_index.append(conflict_idx if detected else None)
        st.is_collide = bool(np.any(collide))
        if st.is_collide:
            st.collision_memory = self.collision_memory_steps
            st.memorized_conflict_points = st.conflict_points.copy()
            st.memorized_conflict_indices = st.conflict_index.copy()
        elif st.collision_memory > 0:
            st.collision_memory -= 1
            st.is_collide = True
        else:
            st.memorized_conflict_points = None
            st.memorized_conflict_indices = None

    def update_reference_states(self, speed_override=None, speed_overide_from_RL=None, st: _EnvState | None = None,
                                ego_speed=None) -> np.ndarray:
        """agents/pure_mpc.py:678-724 (`speed_override` is ignored there too)."""
        st = self._states[0] if st is None else st
        if ego_speed is None:
            ego_speed = self.ego_vehicle.speed
        if speed_overide_from_RL is not None:
            new_ref = self.reference_states
            new_ref[:, 2] = np.clip(speed_overide_from_RL[0, 0], 0, DEFAULT_MAX_SPEED)
            return new_ref
        if not st.is_collide:
            return self.reference_states
        new_ref = self.reference_states
        conflict_indices = st.conflict_index
        if st.collision_memory > 0 and st.memorized_conflict_indices is not None:
            conflict_indices = st.memorized_conflict_indices
        valid = [i for i in conflict_indices if i is not None]
        if not valid:
            return new_ref
        stop_index = max(st.ego_index + 1, min(valid) - SAFETY_BUFFER_POINTS)
        stop_index = min(stop_index, len(self.reference_trajectory) - 1)
        points_to_stop = stop_index - st.ego_index
        if points_to_stop > 0:
            new_ref[st.ego_index:stop_index, 2] = np.linspace(ego_speed, 0, points_to_stop)
            new_ref[stop_index:, 2] = 0.0
            st.stop_point = self.reference_trajectory[stop_index]
            st.last_valid_stop_point = st.stop_point
        elif st.last_valid_stop_point is not None:
            st.stop_point = st.last_valid_stop_point
        return new_ref

    # ------------------------------------------------------------------ problem data -> engine
    def build_solver_inputs(self, states, egos, others, weights_from_RL=None, ref_speed=None):
        """The arguments of `mpc_solve_batch` for a list of parsed environments (agents/pure_mpc.py:95-117)."""
        B = len(egos)
        N = self.horizon
        M = self.global_reference_states.shape[0]
        state = np.empty((B, 4))
        ego_index = np.empty(B, dtype=np.int32)
        vref = np.empty((B, N + 1))
        weights = np.empty((B, 3))
        is_collide = np.zeros(B, dtype=np.uint8)
        V = max([len(o) for o in others] + [0])
        oth = np.zeros((B, max(V, 1), 4))
        oth[:, :, 0] = 1e6          # absent vehicles are parked far away (their distance cost is ~1e-9)
        oth[:, :, 1] = 1e6
        for b in range(B):
            st, ego = states[b], egos[b]
            state[b] = (ego.position[0], ego.position[1], ego.heading, ego.speed)
            st.ego_index = self._nearest_ref_index(ego.position)            # agents/pure_mpc.py:106-109
            ego_index[b] = st.ego_index
            rs = None if ref_speed is None else ref_speed[b:b + 1]
            ref = self.update_reference_states(self.config.get("speed_override"), rs, st, ego.speed)
            vref[b] = ref[np.minimum(st.ego_index + np.arange(N + 1), M - 1), 2]
            if weights_from_RL is None:
                weights[b] = [self.default_weights[f"weight_{k}"] for k in HostPreambleAgent.weight_components]
            else:
                weights[b] = weights_from_RL[b, :3]
            is_collide[b] = 1 if st.is_collide else 0
            for j, veh in enumerate(others[b]):
                oth[b, j] = (veh.position[0], veh.position[1], veh.speed, veh.heading)
        return dict(state=state, ego_index=ego_index, vref=vref, weights=weights, is_collide=is_collide,
                    others=oth if (self.collision_cost and V > 0) else None)

    def _solve_envs(self, states, egos, others, weights_from_RL, ref_speed):
        inp = self.build_solver_inputs(states, egos, others, weights_from_RL, ref_speed)
        self.last_inputs = inp
        out = self._engine.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                                       vref=inp["vref"], others=inp["others"],
                                       collision_cost=self.collision_cost and inp["others"] is not None,
                                       want_trajectories=False)
        self.last_solve = out
        bad = int(np.count_nonzero(out["status"]))
        if bad:                                             # agents/pure_mpc.py:303-305
            print(f"NOTICE: Not found solution ({bad} of {len(egos)} instances)")
        return out["u0"]

    # plotting of the reference is outside the hot path
    def plot(self):
        return None

    def visualize_predictions(self):
        return None
