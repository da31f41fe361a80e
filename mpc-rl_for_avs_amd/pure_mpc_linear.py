"""Host-side mirror of the reference's iterative-linear MPC agent, backed by the MI355X engine.

`IterativeLinearMPC_Agent(env, cfg)` keeps the constructor / `predict()` / `_solve()` surface of the reference
(agents/pure_mpc_linear.py:112-203, agents/base_agent.py:12-79) so that run_pure_mpc_linear.py works unchanged, and
adds `predict_batch()` for many parallel environments.

  predict() / _solve()   one environment: observation parsing in numpy (so that `ego_vehicle`, `agent_vehicles`,
                         `target_ind`, `oa`, `od` exist as the attributes the reference exposes); nearest reference
                         point, forward simulation of the stored profile, linearisation and the QP solve run on the
                         MI355X (`mpc_ltv_solve_batch`)
  predict_batch()        B environments, everything on the device, stored profiles inside the engine
                         (`mpc_ltv_predict_batch`)

The reference solves the QP with cvxpy/ECOS; the engine solves the same QP (strictly convex: one minimiser) with a
Riccati-based primal-dual interior-point method (csrc/mpc_ltv.hpp).  There is no CPU solve path: without the HIP
library / a GPU the constructor raises `EngineError`.  `plot` is a no-op.
"""
from __future__ import annotations

import math

import numpy as np

from .engine import MPCEngine
from .pure_mpc import MPC_Action, Vehicle, normalize_angle
from .reference_path import reference_states as _reference_states

NX = 4  # state = (x, y, v, heading)                         agents/pure_mpc_linear.py:23
NU = 2  # input = (acceleration, steering)                   :24
MAX_STEER = math.radians(30.0)      # :33
MAX_DSTEER = math.radians(30.0)     # :34
MAX_ACCEL = 2.0                     # :35
MAX_DECEL = -5.0                    # :36
MAX_SPEED = 40 / 3.6                # :37


class IterativeLinearMPC_Agent:
    """Drop-in for the reference `IterativeLinearMPC_Agent` (agents/pure_mpc_linear.py:112) solving on an MI355X."""

    def __init__(self, env, cfg: dict, engine: MPCEngine | None = None, device: int = 0, max_iter: int = 100) -> None:
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
        # agents/pure_mpc_linear.py:126-134
        self.oa = None
        self.od = None
        self.wheelbase = self.config.get("wheelbase", 2.5)
        if float(self.wheelbase) != 2.5:
            raise ValueError("the engine is built for the reference's wheelbase of 2.5 m (agents/pure_mpc_linear.py:131)")
        self.target_ind = 0
        # trip count of the linearisation loop (agents/pure_mpc_linear.py:189 hard-codes 1); cfg key of this package
        self.linearization_passes = int(self.config.get("linearization_passes", 1))
        self._engine = engine if engine is not None else MPCEngine(
            horizon=self.horizon, dt=self.dt, max_iter=max_iter, device=device, ref_table=self.global_reference_states,
            ltv_passes=self.linearization_passes)
        self.last_solve = None

    def __str__(self):
        return "Iterative Linear MPC Agent, solved by the MI355X batched engine"

    @property
    def reference_states(self):
        return _reference_states(self.dt)          # agents/base_agent.py:118-154

    normalize_angle = staticmethod(normalize_angle)

    # ------------------------------------------------------------------ reference API
    def predict(self, obs, return_numpy=True):
        """agents/base_agent.py:54-73"""
        self._parse_obs(obs)
        mpc_action = self._solve()
        return mpc_action.numpy() if return_numpy else mpc_action

    def _parse_obs(self, obs: np.ndarray) -> None:
        """agents/base_agent.py:81-116"""
        if not isinstance(obs, np.ndarray):
            raise TypeError(f"Expect observation type np.ndarray, but got {type(obs)}.")
        if obs.shape != (self.total_vehicles_count, 8):
            raise ValueError(
                f"Expect observation's shape of ({(self.total_vehicles_count, 8)}), but got {obs.shape}")
        observed = int(np.sum(obs[:, 0] == 1)) - 1
        self.ego_vehicle = Vehicle(0, obs[0, 1:3], obs[0, 3:5], normalize_angle(obs[0, 5]), obs[0, 6], obs[0, 7])
        self.agent_vehicles = [Vehicle(i + 1, obs[i + 1, 1:3], obs[i + 1, 3:5], obs[i + 1, 5], obs[0, 6], obs[0, 7])
                               for i in range(max(observed, 0))]
        self.observed_vehicles_count = len(self.agent_vehicles)

    def _solve(self) -> MPC_Action:
        """agents/pure_mpc_linear.py:153-203 for the environment parsed by `_parse_obs`."""
        T_ = self.horizon
        ego = self.ego_vehicle
        if self.oa is None or self.od is None:                 # :190-192
            self.oa = np.zeros(T_)
            self.od = np.zeros(T_)
        state = np.array([[ego.position[0], ego.position[1], ego.speed, ego.heading]], dtype=np.float64)
        U = np.stack([self.oa, self.od], axis=1)[None]
        out = self._engine.ltv_solve_batch(state, U)
        self.last_solve = out
        self.target_ind = int(out["target_index"][0])
        self.oa = out["U"][0, :, 0].copy()                      # the profile of the last pass that solved (:197-198);
        self.od = out["U"][0, :, 1].copy()                      # the stored one if the first pass failed
        if out["status"][0] != 0:                               # :193-196 solver failed => fallback
            return MPC_Action(0.0, 0.0)
        return MPC_Action(self.oa[0], self.od[0])

    # ------------------------------------------------------------------ batched API
    def predict_batch(self, obs) -> np.ndarray:
        """B parallel environments: obs[B, vehicles_count, 8] -> actions[B, 2].  Environment b keeps its own stored
        profile inside the engine across calls (same B every call; `reset_env_state` at episode ends).  Equivalent to
        looping `predict` over B agents."""
        obs = np.asarray(obs)
        if obs.ndim != 3 or obs.shape[1:] != (self.total_vehicles_count, 8):
            raise ValueError(f"Expect observations of shape (B, {self.total_vehicles_count}, 8), but got {obs.shape}")
        out = self._engine.ltv_predict_batch(obs.astype(np.float32, copy=False))
        self.last_solve = out
        return out["act"]

    def reset_env_state(self, env_ids=None):
        """Episode boundaries: forget the stored profiles (a new agent starts with oa = od = None)."""
        if env_ids is None or 0 in list(env_ids):
            self.oa = None
            self.od = None
        self._engine.reset_env_state(env_ids)

    def plot(self):
        return None
