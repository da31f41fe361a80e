"""Host-side mirror of the reference's MPC agent interface, backed by the MI355X engine.

`PureMPC_Agent(env, cfg)` keeps the constructor / `predict()` / `_solve()` surface of the reference
(agents/pure_mpc.py:13-78, agents/base_agent.py:12-116) so that `PPO_MPC` / `A2C_MPC`
(agents/ppo_mpc.py:197-200,422-427; agents/a2c_mpc.py:103-106,145-150) and the run scripts work unchanged,
and adds `predict_batch()` for many parallel environments.  What happens where:

  predict() / _solve()      one environment, the reference's call sequence.  `_parse_obs` validates the observation and
                            builds the Vehicle objects callers read; the path-crossing "collision" detection with
                            its 10-step memory, the rewrite of the reference speed profile (agents/pure_mpc.py:552-724)
                            and the NLP solve (agents/pure_mpc.py:80-318) run on the MI355X in ONE C-ABI call,
                            `mpc_predict_batch` with B = 1; the attributes the reference leaves on the agent
                            (is_collide, conflict_points, conflict_index, agent_collide, stop_point, ego_index, ...) are
                            read back from the engine's per-environment record (`mpc_get_env_state`)
  predict_batch()           B environments, the same call with B > 1 (csrc/mpc_preamble.hpp: 16 lanes per environment,
                            detector memory inside the engine)

There is exactly one implementation of the preamble in the product, the device one; its numpy mirror - the checker,
pinned by outputs of the reference's own functions - lives in tests/host_preamble.py.

There is no CPU solve path: without the HIP library / a GPU the constructor raises `EngineError`.
Plotting (`plot`, `visualize_predictions`) is not part of the hot path and is a no-op here.
"""
from __future__ import annotations

import numpy as np

from .engine import MPCEngine, converged
from .reference_path import reference_states as _reference_states

PREDICTION_HORIZON = 30     # agents/pure_mpc.py:554
TIME_THRESHOLD = 30         # agents/pure_mpc.py:555
COLLISION_MEMORY_STEPS = 10  # agents/pure_mpc.py:39
DEFAULT_MAX_SPEED = 30.0    # agents/pure_mpc.py:680
SAFETY_BUFFER_POINTS = 5    # agents/pure_mpc.py:681


class MPC_Action:
    """agents/utils.py:4-12, plus what the reference only prints (`solver.stats()['success']`, agents/pure_mpc.py:303-305):
    `success` (solved to tolerance), the engine's `status` code and iteration count - for callers that hold nothing but
    the action (trainer.predict-style code)."""

    def __init__(self, acceleration, steer, success=True, status=0, iters=0) -> None:
        self.acceleration = acceleration
        self.steer = steer
        self.success = bool(success)
        self.status = int(status)
        self.iters = int(iters)

    def numpy(self) -> np.ndarray:
        return np.array([self.acceleration, self.steer])


class Vehicle:
    """agents/utils.py:16-40"""
    LENGTH = 2.5
    LENGTH_REAR = LENGTH / 2

    def __init__(self, index, position, vectorized_speed, heading, sinh, cosh):
        self.index = index
        self.is_ego = index == 0
        self.position = position
        self.vectorized_speed = vectorized_speed
        self.heading = heading
        self.speed = np.linalg.norm(self.vectorized_speed)
        self.sinh = sinh
        self.cosh = cosh
        self.max_acceleration = 3.5
        self.max_deceleration = -10


def normalize_angle(angle):
    """agents/base_agent.py:156-170"""
    while angle > np.pi:
        angle -= 2 * np.pi
    while angle < -np.pi:
        angle += 2 * np.pi
    return angle


class PureMPC_Agent:
    """Drop-in for the reference `PureMPC_Agent` (agents/pure_mpc.py:13) running on an MI355X."""

    weight_components = ["speed", "control", "input_diff"]   # agents/pure_mpc.py:15-22

    # the solver options the reference passes to IPOPT (agents/pure_mpc.py:291-296; read back from its own `opts` dict by
    # tests/golden/make_golden.py: reference_sequences.npz seq_ipopt_max_iter / seq_ipopt_tol)
    REFERENCE_MAX_ITER = 1000
    REFERENCE_TOL = 1e-6

    def __init__(self, env, cfg: dict, engine: MPCEngine | None = None, collision_cost: bool = False,
                 device: int = 0, max_iter: int = 100, tol: float = 1e-8, warm_start: bool = False,
                 reference_settings: bool = False, stall_window: int = 0) -> None:
        """reference_settings=True: `ipopt.max_iter 1000`, `ipopt.tol 1e-6` as in agents/pure_mpc.py:294-295 instead of the
        engine's defaults (100 iterations, tol 1e-8 - a tighter tolerance under a latency budget: a batched call lasts as
        long as its slowest instance, and an instance at the cap returns its last iterate like the reference's failed
        solve does, :303-305).  stall_window (not an IPOPT option, off by default, include/mpc_mi355x.h): ends solves whose
        KKT error has stopped halving, which is what keeps a max_iter-1000 batch from waiting for the one instance in a
        thousand that never converges."""
        if reference_settings:
            max_iter, tol = PureMPC_Agent.REFERENCE_MAX_ITER, PureMPC_Agent.REFERENCE_TOL
        self.solver_settings = dict(max_iter=int(max_iter), tol=float(tol), stall_window=int(stall_window),
                                    reference_settings=bool(reference_settings))
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
        self.default_weights = {f"weight_{k}": self.config[f"weight_{k}"] for k in PureMPC_Agent.weight_components}
        self.collision_cost = bool(collision_cost)
        self.warm_start = bool(warm_start)       # not in the reference, which always starts cold
        self._engine = engine if engine is not None else MPCEngine(
            horizon=self.horizon, dt=self.dt, max_iter=max_iter, tol=tol, stall_window=stall_window,
            w_distance=float(self.config.get("weight_distance", 10.0)),
            w_collision=float(self.config.get("weight_collision", 1.0)), device=device,
            ref_table=self.global_reference_states)
        self._obs = None
        self._detected = False     # the detector has already seen self._obs (a stand-alone _check_collision or a _solve)
        self._env0 = None          # detector record of environment 0 after the last predict()
        self.last_acc = 0          # agents/pure_mpc.py:63
        self.last_solve = None

    def __str__(self) -> str:
        return "Pure MPC agent [Receding Horizon Control], solved by the MI355X batched engine"

    # ---- single-environment attributes of the reference (agents/pure_mpc.py:38-43, 589-593), served from the engine's
    #      record of environment 0 (mpc_get_env_state) after predict()
    def _rec(self, key, default):
        return default if self._env0 is None else self._env0[key]

    is_collide = property(lambda self: bool(self._rec("is_collide", False)))
    ego_index = property(lambda self: int(self._rec("ego_index", 0)))
    collision_memory = property(lambda self: int(self._rec("collision_memory", 0)))

    @property
    def conflict_index(self):
        """Reference index of the conflict point per observed vehicle, None where the paths do not cross."""
        return [None if c < 0 else int(c) for c in self._rec("conflict_index", [])]

    @property
    def conflict_points(self):
        """Where the ego's and each vehicle's predicted paths cross (agents/pure_mpc.py:590, 656), None if they do not."""
        ci, cp = self._rec("conflict_index", []), self._rec("conflict_points", [])
        return [None if c < 0 else np.array(p) for c, p in zip(ci, cp)]

    @property
    def agent_collide(self):
        """agents/pure_mpc.py:593, 659: per observed vehicle, does its predicted path cross the ego's."""
        return [bool(c >= 0) for c in self._rec("conflict_index", [])]

    @property
    def stop_point(self):
        s = int(self._rec("stop_index", -1))
        return None if s < 0 else self.reference_trajectory[s]

    @property
    def agent_current_locations(self):
        """agents/pure_mpc.py:591: positions of the observed vehicles."""
        return [np.array(v.position) for v in self.agent_vehicles]

    @property
    def agent_future_locations(self):
        """agents/pure_mpc.py:592, 605: the 30-step constant-velocity polylines of the observed vehicles (display data,
        derived from the observation alone; the detector's own copies live on the device)."""
        return [self.predict_future_positions(np.array(v.position), v.speed, v.heading, self.dt, PREDICTION_HORIZON)
                for v in self.agent_vehicles]

    @property
    def reference_states(self):
        return _reference_states(self.dt)          # agents/base_agent.py:118-154 (fresh copy per access)

    normalize_angle = staticmethod(normalize_angle)

    def other_vehicle_model(self, other_vehicle, dt):      # agents/base_agent.py:172-174
        return other_vehicle.position + other_vehicle.speed * dt * np.array(
            [np.cos(other_vehicle.heading), np.sin(other_vehicle.heading)])

    def predict_future_positions(self, current_position, speed, heading, dt, prediction_horizon):
        """agents/pure_mpc.py:529-550: constant-velocity prediction."""
        future_positions = [current_position]
        step = speed * dt * np.array([np.cos(heading), np.sin(heading)])
        for _ in range(prediction_horizon):
            future_positions.append(future_positions[-1] + step)
        return future_positions

    def reset_env_state(self, env_ids=None):
        """Forget the collision memory of the given environments (episode boundaries; None = all) - what constructing a
        new agent does in the reference."""
        self._engine.reset_env_state(env_ids)
        if env_ids is None or 0 in list(env_ids):
            self._env0 = None

    def save_env_state(self, B=1):
        """Checkpoint of the detector state of environments 0..B-1 (opaque bytes); `load_env_state` restores it."""
        return self._engine.save_env_state(B)

    def load_env_state(self, records):
        self._engine.load_env_state(records)

    # ------------------------------------------------------------------ reference API
    def predict(self, obs, return_numpy=True, weights_from_RL=None, ref_speed=None):
        """agents/pure_mpc.py:68-78"""
        self._parse_obs(obs)
        self._check_collision(_within_predict=True)
        mpc_action = self._solve(weights_from_RL, ref_speed)
        return mpc_action.numpy() if return_numpy else mpc_action

    def _parse_obs(self, obs: np.ndarray) -> None:
        """agents/base_agent.py:81-116: validation and the Vehicle objects callers read; the engine parses the same
        observation again on the device."""
        if not isinstance(obs, np.ndarray):
            raise TypeError(f"Expect observation type np.ndarray, but got {type(obs)}.")
        if obs.shape != (self.total_vehicles_count, 8):
            raise ValueError(
                f"Expect observation's shape of ({(self.total_vehicles_count, 8)}), but got {obs.shape}")
        self._obs = obs
        self._detected = False
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

    def _check_collision(self, _within_predict=False):
        """agents/pure_mpc.py:552-676.  The detector and its memory live in the engine.  Inside `predict()` it runs in
        the same device call as the solve (`_solve`), so this step has nothing to do there.  Called on its own - the
        reference's sequence `_parse_obs -> _check_collision -> read is_collide -> _solve` - it runs the detector for
        the parsed observation now (`MPC_FLAG_DETECT_ONLY`), refreshes `is_collide`, `conflict_index`, ... and tells
        the following `_solve` not to advance the detector a second time (`MPC_FLAG_DETECTED`)."""
        if _within_predict or self._detected:
            return None
        if self._obs is None:
            raise RuntimeError("_check_collision: call _parse_obs(obs) first")
        self._engine.detect_batch(np.ascontiguousarray(self._obs, dtype=np.float32)[None])
        self._detected = True
        self._read_env0()
        return None

    def _read_env0(self):
        rec = self._engine.env_state(1)
        n = self.observed_vehicles_count
        self._env0 = dict(is_collide=rec["is_collide"][0], ego_index=rec["ego_index"][0],
                          collision_memory=rec["collision_memory"][0], stop_index=rec["stop_index"][0],
                          conflict_index=rec["conflict_index"][0, :n].copy(),
                          conflict_points=rec["conflict_points"][0, :n].copy())

    def _solve(self, weights_from_RL=None, ref_speed_from_RL=None) -> MPC_Action:
        """agents/pure_mpc.py:80-318 for the environment parsed by `_parse_obs`: one `mpc_predict_batch` with B = 1."""
        if self._obs is None:
            raise RuntimeError("_solve: call _parse_obs(obs) first")
        w = None if weights_from_RL is None else np.asarray(weights_from_RL, dtype=np.float64).reshape(1, -1)[:, :3]
        rs = None if ref_speed_from_RL is None else np.asarray(ref_speed_from_RL, dtype=np.float64).reshape(1, 1)
        act = self._predict_device(np.ascontiguousarray(self._obs, dtype=np.float32)[None], w, rs,
                                   detected=self._detected)
        self._detected = True          # a second _solve for the same observation must not advance the detector either
        self._read_env0()
        self.last_acc = act[0, 0]
        st = int(self.last_solve["status"][0])
        return MPC_Action(acceleration=act[0, 0], steer=act[0, 1], success=bool(converged(np.int32(st))), status=st,
                          iters=int(self.last_solve["iters"][0]))

    # ------------------------------------------------------------------ batched API
    def predict_batch(self, obs, weights_from_RL=None, ref_speed=None) -> np.ndarray:
        """B parallel environments: obs[B, vehicles_count, 8] -> actions[B, 2] (acceleration, steer).

        `weights_from_RL` is [B, 3] (speed, control, input_diff) or None, `ref_speed` [B, 1] or None;
        environment b keeps its own collision memory across calls (same B every call, `reset_env_state` at
        episode ends).  Equivalent to looping `predict` over B agents; everything runs on the device
        (`mpc_predict_batch`)."""
        obs, w, rs = self._check_batch_args(obs, weights_from_RL, ref_speed)
        return self._predict_device(obs, w, rs)

    def _predict_device(self, obs, w, rs, detected=False):
        B = obs.shape[0]
        if w is None:
            w = np.tile([float(self.default_weights[f"weight_{k}"]) for k in PureMPC_Agent.weight_components], (B, 1))
        out = self._engine.predict_batch(obs, w, None if rs is None else rs[:, 0], collision_cost=self.collision_cost,
                                         warm_start=self.warm_start, detected=detected)
        self.last_solve = out
        bad = int(np.count_nonzero(~converged(out["status"])))
        if bad:                                             # agents/pure_mpc.py:303-305
            print(f"NOTICE: Not found solution ({bad} of {B} instances)")
        return out["act"]

    def batch_env_state(self, B):
        """is_collide / ego_index / collision_memory / stop_index / conflict_index / conflict_points of environments
        0..B-1 after `predict_batch` (the batched counterpart of the attributes `predict` leaves on the agent)."""
        return self._engine.env_state(B)

    def _check_batch_args(self, obs, weights_from_RL, ref_speed):
        if not isinstance(obs, np.ndarray):
            raise TypeError(f"Expect observation type np.ndarray, but got {type(obs)}.")
        if obs.ndim != 3 or obs.shape[1:] != (self.total_vehicles_count, 8):
            raise ValueError(f"Expect observations of shape (B, {self.total_vehicles_count}, 8), but got {obs.shape}")
        B = obs.shape[0]
        w = None if weights_from_RL is None else np.asarray(weights_from_RL, dtype=np.float64).reshape(B, -1)[:, :3]
        rs = None if ref_speed is None else np.asarray(ref_speed, dtype=np.float64).reshape(B, 1)
        return obs, w, rs

    # plotting of the reference is outside the hot path
    def plot(self):
        return None

    def visualize_predictions(self):
        return None
