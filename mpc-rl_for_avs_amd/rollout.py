"""MPC-in-the-loop rollout collection for many parallel environments (BASELINE configs 4-5).

The reference collects experience with ONE environment and one MPC solve per step:
    policy(obs) -> clip -> mpc_agent.predict(obs, weights_from_RL | ref_speed) -> env.step(mpc_action) -> buffer.add
(`PPO_MPC.collect_rollouts` agents/ppo_mpc.py:353-483, `A2C_MPC.collect_rollouts` agents/a2c_mpc.py:111-180, both
copies of stable-baselines3's on-policy loop with the MPC call spliced in).  `BatchedCollector` is that loop for B
environments at once with every tensor resident on the GPU: the policy is a torch module, the MPC call is
`MPCEngine.predict_batch_torch` (one C-ABI call: device preamble + solve, zero-copy on torch's stream), the
environment steps as vectorised torch ops, and the buffer is a set of device tensors.  With `torch.distributed`
initialised the environments are sharded over the ranks and the MPC actions are all-gathered (RCCL) every step,
the path's only exchange.

highway-env, gymnasium and stable-baselines3 are not available offline, so
  * `SyntheticIntersectionEnv` is a stand-in with the reference's observation layout (config/config.py:10-26:
    10 rows x [presence, x, y, vx, vy, heading, sin_h, cos_h], absolute, sorted by distance), its ego route (the
    reference path of agents/base_agent.py:118-154), the MPC's own vehicle model for the ego (agents/pure_mpc.py:
    220-228), constant-velocity traffic on the four approach lanes, and the reward / termination *shape* of
    envs/intersection_env_Feb2025_v1.py:80-155 (collision, speed, arrival, lane centring, off-road; crash or arrival
    terminates, 200 steps truncate).  It is scaffolding for throughput measurements, not a re-implementation of
    highway-env.
  * `ActorCritic` has the shape of SB3's default `MlpPolicy` used by the reference's `create_ppo_policy /
    create_a2c_policy` (trainers/trainer_utils.py:6-45): flattened 80-d observation, separate 64-64 tanh towers
    for policy and value, diagonal Gaussian over a Box(-1, 1)^dim action (dim 1 = reference speed "v0",
    dim 3 = cost weights "v1", trainers/trainer.py:422-428).
  * `RolloutBuffer.compute_returns_and_advantage` is generalised advantage estimation as SB3 defines it, and
    `OnPolicyTrainer` the PPO / A2C parameter update of the reference's `train()` methods, so that a whole
    collect -> update loop runs without stable-baselines3.
"""
from __future__ import annotations

import os

import math

import numpy as np
import torch

from .reference_path import reference_states

VEHICLES_COUNT = 10        # config/cfg.yaml:2
EPISODE_STEPS = 200        # duration 10 s + 10 s at 10 Hz (envs/intersection_env_Feb2025_v1.py:153-155, config.py:41)
REWARD = dict(collision=-200.0, high_speed=15.0, arrived=50.0, center_bonus=5.0, off_road=-50.0)  # env :47-52,:96-107
LANE_HALF_WIDTH = 2.0
CRASH_DISTANCE = 2.5
WHEELBASE = 2.5


class SyntheticIntersectionEnv:
    """B independent intersection episodes stepped together on one device (auto-reset like an SB3 VecEnv).

    Two implementations of the same step behind this class: on a GPU the fused HIP kernel `mpc_synth_env_step`
    (csrc/mpc_synth_env.hpp: vehicle models, respawn, reward, termination, terminal observation, auto-reset and the next
    observation in ONE launch, counter-based random numbers) - `backend="hip"`, the default there; on the CPU, or with
    `backend="torch"`, the vectorised torch ops below (about a hundred small kernels per step on a GPU).  Same state
    tensors, same return values; the two draw different random streams (as torch's CPU and GPU generators do), the
    deterministic part of the step is identical (tests/test_rollout_cpu.py, tests/test_predict_gpu.py)."""

    def __init__(self, num_envs: int, device="cpu", seed: int = 0, n_others: int = 4, dt: float = 0.1,
                 spawn_probability: float = 0.3, backend: str = "auto", env_offset: int = 0):
        assert 0 <= n_others <= VEHICLES_COUNT - 1
        self.num_envs, self.K, self.dt = int(num_envs), int(n_others), float(dt)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            # an indexed device, resolved once: the fused step hands the ordinal to hipSetDevice, and "cuda" without an index
            # means the CURRENT device (torch.cuda.set_device(local_rank) under torchrun), not device 0
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.spawn_probability = float(spawn_probability)
        if backend not in ("auto", "hip", "torch"):
            raise ValueError("backend must be auto|hip|torch")
        self.backend = ("hip" if self.device.type == "cuda" else "torch") if backend == "auto" else backend
        if self.backend == "hip" and self.device.type != "cuda":
            raise ValueError("the fused environment step needs a GPU device")
        self.seed, self.env_offset = int(seed), int(env_offset)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        ref = reference_states(dt)
        self.ref_xy = torch.as_tensor(np.ascontiguousarray(ref[:, :2]), dtype=torch.float64, device=self.device)
        self.M = ref.shape[0]
        B, K = self.num_envs, max(self.K, 1)
        self.ego = torch.zeros((B, 4), dtype=torch.float64, device=self.device)      # x, y, heading, speed
        self.opos = torch.zeros((B, K, 2), dtype=torch.float64, device=self.device)
        self.ospeed = torch.zeros((B, K), dtype=torch.float64, device=self.device)
        self.ohead = torch.zeros((B, K), dtype=torch.float64, device=self.device)
        self.oactive = torch.zeros((B, K), dtype=torch.bool, device=self.device)
        self.t = torch.zeros(B, dtype=torch.int32, device=self.device)
        self._lane_h = torch.tensor([0.0, math.pi / 2, math.pi, -math.pi / 2], dtype=torch.float64, device=self.device)
        if self.backend == "hip":
            from . import engine as _engine
            self._lib = _engine.load_library()
            self.rng_counter = torch.zeros(B, dtype=torch.int64, device=self.device)
            u8 = lambda: torch.zeros(B, dtype=torch.uint8, device=self.device)
            # outputs of the step live at fixed addresses (a captured hipGraph replays against them)
            self._out = dict(obs=torch.zeros((B, VEHICLES_COUNT, 8), dtype=torch.float32, device=self.device),
                             terminal_obs=torch.zeros((B, VEHICLES_COUNT, 8), dtype=torch.float32, device=self.device),
                             reward=torch.zeros(B, dtype=torch.float32, device=self.device),
                             done=u8(), truncated=u8(), crashed=u8(), arrived=u8())

    # ---- fused implementation (csrc/mpc_synth_env.hpp) ---------------------------------------------------------------
    def _hip_call(self, action, reset_all):
        import ctypes
        o = self._out
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._lib.mpc_synth_env_step(
            self.device.index, self.num_envs, self.K, self.dt, self.spawn_probability, self.seed, self.env_offset,
            p(self.ref_xy), self.M, p(action), p(self.ego), p(self.opos), p(self.ospeed), p(self.ohead), p(self.oactive),
            p(self.t), p(self.rng_counter), p(o["obs"]), p(o["terminal_obs"]), p(o["reward"]), p(o["done"]),
            p(o["truncated"]), p(o["crashed"]), p(o["arrived"]), 1 if reset_all else 0, stream)
        if rc != 0:
            raise RuntimeError(f"mpc_synth_env_step failed ({rc}): {self._lib.mpc_last_error().decode()}")

    # ---- random helpers ------------------------------------------------------------------------------------
    def _u(self, shape, lo, hi):
        return lo + (hi - lo) * torch.rand(shape, generator=self.gen, device=self.device, dtype=torch.float64)

    def _spawn_others(self, shape, dlo, dhi):
        """Vehicles on the four approach lanes driving towards the centre (right-hand traffic, lane offset 2 m)."""
        lane = torch.randint(0, 4, shape, generator=self.gen, device=self.device)
        h = self._lane_h[lane]
        d = self._u(shape, dlo, dhi)
        x = -d * torch.cos(h) + torch.where(lane == 1, -2.0, 0.0) + torch.where(lane == 3, 2.0, 0.0)
        y = -d * torch.sin(h) + torch.where(lane == 0, 2.0, 0.0) + torch.where(lane == 2, -2.0, 0.0)
        sp = torch.clamp(8.0 + torch.randn(shape, generator=self.gen, device=self.device, dtype=torch.float64), min=0.0)
        return torch.stack([x, y], dim=-1), sp, h

    def _reset_where(self, mask):
        B, K = self.num_envs, max(self.K, 1)
        ego = torch.zeros((B, 4), dtype=torch.float64, device=self.device)
        ego[:, 0] = 2.0
        ego[:, 1] = 45.0 + self._u((B,), -5.0, 5.0)          # envs/intersection_env_Feb2025_v1.py:397-410
        ego[:, 2] = -math.pi / 2
        ego[:, 3] = 10.0
        pos, sp, h = self._spawn_others((B, K), 5.0, 60.0)
        m1, m2 = mask[:, None], mask[:, None, None]
        # state tensors are updated in place: a captured hipGraph (BatchedCollector(use_graph=True)) replays against
        # fixed addresses
        self.ego.copy_(torch.where(m1, ego, self.ego))
        self.opos.copy_(torch.where(m2, pos, self.opos))
        self.ospeed.copy_(torch.where(m1, sp, self.ospeed))
        self.ohead.copy_(torch.where(m1, h, self.ohead))
        active = torch.ones((B, K), dtype=torch.bool, device=self.device) if self.K > 0 else \
            torch.zeros((B, K), dtype=torch.bool, device=self.device)
        self.oactive.copy_(torch.where(m1, active, self.oactive))
        self.t.copy_(torch.where(mask, torch.zeros_like(self.t), self.t))

    # ---- observation (config/config.py:10-26) ------------------------------------------------------------------
    def observe(self) -> torch.Tensor:
        """(torch implementation; also the independent check of the fused kernel's observation)"""
        B, K = self.num_envs, max(self.K, 1)
        obs = torch.zeros((B, VEHICLES_COUNT, 8), dtype=torch.float32, device=self.device)
        x, y, th, v = self.ego.unbind(dim=1)
        obs[:, 0, 0] = 1.0
        obs[:, 0, 1], obs[:, 0, 2] = x.float(), y.float()
        obs[:, 0, 3], obs[:, 0, 4] = (v * torch.cos(th)).float(), (v * torch.sin(th)).float()
        obs[:, 0, 5], obs[:, 0, 6], obs[:, 0, 7] = th.float(), torch.sin(th).float(), torch.cos(th).float()
        if self.K > 0:
            d = torch.linalg.norm(self.opos - self.ego[:, None, :2], dim=-1)
            d = torch.where(self.oactive, d, torch.full_like(d, float("inf")))
            order = torch.argsort(d, dim=1, stable=True)                  # `order: sorted`: nearest first, absent last
            g = lambda t: torch.gather(t, 1, order)
            act = g(self.oactive)
            px = torch.gather(self.opos[..., 0], 1, order)
            py = torch.gather(self.opos[..., 1], 1, order)
            sp, hh = g(self.ospeed), g(self.ohead)
            rows = torch.stack([torch.ones_like(px), px, py, sp * torch.cos(hh), sp * torch.sin(hh), hh, torch.sin(hh),
                                torch.cos(hh)], dim=-1)
            rows = torch.where(act[..., None], rows, torch.zeros_like(rows))
            obs[:, 1:1 + K] = rows.float()
        return obs

    def reset(self) -> torch.Tensor:
        if self.backend == "hip":
            self._hip_call(None, True)
            return self._out["obs"]
        self._reset_where(torch.ones(self.num_envs, dtype=torch.bool, device=self.device))
        return self.observe()

    # ---- one policy step ---------------------------------------------------------------------------------------
    def step(self, action: torch.Tensor):
        """action[B, 2] = acceleration [m/s^2], steering angle [rad] (what the RL wrappers hand to env.step,
        agents/ppo_mpc.py:430-432).  Returns obs, reward, done, info like an SB3 VecEnv with auto-reset:
        info = dict(terminal_obs, truncated, crashed, arrived)."""
        if self.backend == "hip":
            if action.dtype != torch.float64 or not action.is_contiguous():
                action = action.to(torch.float64).contiguous()
            self._hip_call(action, False)
            o = self._out
            b = lambda k: o[k].view(torch.bool)
            return o["obs"], o["reward"], b("done"), dict(terminal_obs=o["terminal_obs"], truncated=b("truncated"),
                                                          crashed=b("crashed"), arrived=b("arrived"))
        a = torch.clamp(action[:, 0].to(torch.float64), -5.0, 5.0)                      # config/config.py:31
        delta = torch.clamp(action[:, 1].to(torch.float64), -math.pi / 4, math.pi / 4)  # config/config.py:30
        x, y, th, v = self.ego.unbind(dim=1)
        beta = torch.atan(0.5 * torch.tan(delta))
        dt = self.dt
        nx = x + v * torch.cos(th + beta) * dt
        ny = y + v * torch.sin(th + beta) * dt
        nth = th + v / WHEELBASE * torch.sin(beta) * dt
        nv = torch.clamp(v + a * dt, 0.0, 30.0)
        self.ego.copy_(torch.stack([nx, ny, nth, nv], dim=1))
        if self.K > 0:
            step = (self.ospeed * dt)[..., None] * torch.stack([torch.cos(self.ohead), torch.sin(self.ohead)], dim=-1)
            self.opos.add_(step)
            gone = (self.opos.abs().amax(dim=-1) > 65.0) | ~self.oactive
            respawn = gone & (torch.rand(gone.shape, generator=self.gen, device=self.device) < self.spawn_probability)
            pos, sp, h = self._spawn_others(gone.shape, 40.0, 60.0)
            self.opos.copy_(torch.where(respawn[..., None], pos, self.opos))
            self.ospeed.copy_(torch.where(respawn, sp, self.ospeed))
            self.ohead.copy_(torch.where(respawn, h, self.ohead))
            self.oactive.copy_((self.oactive & ~gone) | respawn)
            dist = torch.linalg.norm(self.opos - self.ego[:, None, :2], dim=-1)
            crashed = ((dist < CRASH_DISTANCE) & self.oactive).any(dim=1)
        else:
            crashed = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        dref = torch.linalg.norm(self.ref_xy[None] - self.ego[:, None, :2], dim=-1)
        lateral, idx = dref.min(dim=1)
        on_road = lateral <= LANE_HALF_WIDTH
        arrived = (idx >= self.M - 3) & on_road
        centering = 1.0 - torch.clamp(lateral / LANE_HALF_WIDTH, max=1.0)
        reward = (REWARD["collision"] * crashed + REWARD["high_speed"] * (self.ego[:, 3] / 10.0) +
                  REWARD["arrived"] * arrived + torch.where(on_road, REWARD["center_bonus"] * centering,
                                                            torch.full_like(centering, REWARD["off_road"])))
        self.t.add_(1)
        terminated = crashed | arrived
        truncated = (self.t >= EPISODE_STEPS) & ~terminated
        done = terminated | truncated
        terminal_obs = self.observe()
        self._reset_where(done)
        obs = torch.where(done[:, None, None], self.observe(), terminal_obs)
        return obs, reward.float(), done, dict(terminal_obs=terminal_obs, truncated=truncated, crashed=crashed,
                                               arrived=arrived)


class ActorCritic(torch.nn.Module):
    """SB3 `MlpPolicy`-shaped actor-critic: 80 -> 64 -> 64 (tanh) twice, Gaussian head of `action_dim`."""

    def __init__(self, action_dim: int, obs_dim: int = VEHICLES_COUNT * 8, hidden: int = 64):
        super().__init__()
        mk = lambda: torch.nn.Sequential(torch.nn.Linear(obs_dim, hidden), torch.nn.Tanh(),
                                         torch.nn.Linear(hidden, hidden), torch.nn.Tanh())
        self.pi, self.vf = mk(), mk()
        self.action_net = torch.nn.Linear(hidden, action_dim)
        self.value_net = torch.nn.Linear(hidden, 1)
        self.log_std = torch.nn.Parameter(torch.zeros(action_dim))
        self.action_dim = action_dim

    def _dist(self, obs):
        mean = self.action_net(self.pi(obs.flatten(1)))
        # validate_args=False: the default argument checks read a device boolean back on the host (a sync per step)
        return torch.distributions.Normal(mean, self.log_std.exp().expand_as(mean), validate_args=False)

    def forward(self, obs, deterministic: bool = False, generator=None):
        d = self._dist(obs)
        if deterministic:
            actions = d.mean
        else:
            actions = d.mean + d.stddev * torch.randn(d.mean.shape, generator=generator, device=d.mean.device,
                                                      dtype=d.mean.dtype)
        return actions, self.predict_values(obs), d.log_prob(actions).sum(dim=1)

    def predict_values(self, obs):
        return self.value_net(self.vf(obs.flatten(1)))[:, 0]

    # ---- rollout-time forward (no gradients): both towers as ONE 80 -> 128 -> 128 -> (action_dim + 1) network (first
    #      layers side by side, second layer block-diagonal, heads in one matrix) and the Gaussian sample / log-probability
    #      written out - 3 GEMMs, 2 tanh and 6 small kernels instead of 6 + 4 + 12.  Same function as forward(); the fused
    #      weights live in persistent buffers (a captured hipGraph reads them in place) refreshed by refresh_fused().
    @torch.no_grad()
    def refresh_fused(self):
        A, H = self.action_dim, self.pi[0].out_features
        dev, dt = self.log_std.device, self.log_std.dtype
        if getattr(self, "_fz", None) is None or self._fz["w1"].device != dev:
            z = lambda *shape: torch.zeros(shape, device=dev, dtype=dt)
            self._fz = dict(w1=z(self.pi[0].in_features, 2 * H), b1=z(2 * H), w2=z(2 * H, 2 * H), b2=z(2 * H),
                            wh=z(2 * H, A + 1), bh=z(A + 1), std=z(A), c0=z(1)[0])
        f = self._fz
        f["w1"][:, :H].copy_(self.pi[0].weight.t())
        f["w1"][:, H:].copy_(self.vf[0].weight.t())
        f["b1"][:H].copy_(self.pi[0].bias)
        f["b1"][H:].copy_(self.vf[0].bias)
        f["w2"][:H, :H].copy_(self.pi[2].weight.t())
        f["w2"][H:, H:].copy_(self.vf[2].weight.t())
        f["b2"][:H].copy_(self.pi[2].bias)
        f["b2"][H:].copy_(self.vf[2].bias)
        f["wh"][:H, :A].copy_(self.action_net.weight.t())
        f["wh"][H:, A].copy_(self.value_net.weight[0])
        f["bh"][:A].copy_(self.action_net.bias)
        f["bh"][A:].copy_(self.value_net.bias)
        f["std"].copy_(self.log_std.exp())
        f["c0"].copy_(self.log_std.sum() + 0.5 * A * math.log(2.0 * math.pi))

    @torch.no_grad()
    def act(self, obs, generator=None, noise=None):
        """(actions, values, log_probs) like forward(obs), through the fused weights (call refresh_fused() after every
        change of the parameters).  noise: the sample's standard-normal draws [B, action_dim] instead of the generator's."""
        f, A = self._fz, self.action_dim
        h = torch.tanh(torch.addmm(f["b1"], obs.flatten(1), f["w1"]))
        h = torch.tanh(torch.addmm(f["b2"], h, f["w2"]))
        out = torch.addmm(f["bh"], h, f["wh"])
        mean = out[:, :A]
        if noise is None:
            noise = torch.randn(mean.shape, generator=generator, device=mean.device, dtype=mean.dtype)
        actions = torch.addcmul(mean, noise, f["std"])
        log_probs = (noise * noise).sum(dim=1).mul_(-0.5).sub_(f["c0"])
        return actions, out[:, A], log_probs

    def evaluate_actions(self, obs, actions):
        d = self._dist(obs)
        return self.predict_values(obs), d.log_prob(actions).sum(dim=1), d.entropy().sum(dim=1)


class RolloutBuffer:
    """[n_steps, B, ...] device tensors + GAE (the arithmetic of SB3's RolloutBuffer, which the reference uses as is).

    The float32 fields of a step (observation, action, reward, episode start, value, log-probability and - for PPO - the
    terminal observation and the truncation flag) are columns of ONE [n_steps, B, F] tensor, so that a step writes its row
    with one concatenation and one copy instead of eight; `obs`, `actions`, ... are views of it."""

    def __init__(self, n_steps, num_envs, action_dim, device, gamma=0.99, gae_lambda=0.95, keep_terminal=False):
        T, B, A, O = int(n_steps), int(num_envs), int(action_dim), VEHICLES_COUNT * 8
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=device)
        self._cols = O + A + 4 + ((O + 1) if keep_terminal else 0)
        self._row = z(T, B, self._cols)
        r, c = self._row, O + A
        self.obs = r[..., :O].view(T, B, VEHICLES_COUNT, 8)
        self.actions = r[..., O:c]
        self.rewards, self.episode_starts, self.values, self.log_probs = r[..., c], r[..., c + 1], r[..., c + 2], r[..., c + 3]
        # PPO bootstraps truncated episodes with the value of their terminal observation (agents/ppo_mpc.py:451-461): the
        # observations and flags are kept per step and the values are computed for the whole rollout at once at its end
        # (bootstrap_truncated) - the policy does not change during a rollout - instead of one value tower per step
        self.terminal_obs = r[..., c + 4:c + 4 + O].view(T, B, VEHICLES_COUNT, 8) if keep_terminal else None
        self.truncated = r[..., c + 4 + O] if keep_terminal else None
        self.advantages, self.returns = z(T, B), z(T, B)
        self.mpc_actions = z(T, B, 2, dt=torch.float64)
        self.gamma, self.gae_lambda, self.n_steps, self.pos = float(gamma), float(gae_lambda), T, 0
        self.pos_dev = torch.zeros(1, dtype=torch.int64, device=device)     # the same counter for captured graphs

    def reset(self):
        self.pos = 0
        self.pos_dev.zero_()

    def _pack(self, obs, actions, rewards, episode_starts, values, log_probs, terminal_obs, truncated):
        f = lambda t: t.reshape(t.shape[0], -1)
        cols = [f(obs), f(actions), f(rewards), f(episode_starts), f(values), f(log_probs)]
        if self.terminal_obs is not None:
            cols += [f(terminal_obs), f(truncated)]
        row = torch.cat(cols, dim=1)
        assert row.shape[1] == self._cols
        return row

    def add(self, obs, actions, rewards, episode_starts, values, log_probs, mpc_actions=None, terminal_obs=None,
            truncated=None):
        i = self.pos
        self._row[i] = self._pack(obs, actions, rewards, episode_starts, values, log_probs, terminal_obs, truncated)
        if mpc_actions is not None:
            self.mpc_actions[i] = mpc_actions
        self.pos += 1
        self.pos_dev.add_(1)

    def add_at_device_pos(self, obs, actions, rewards, episode_starts, values, log_probs, mpc_actions, terminal_obs=None,
                          truncated=None):
        """`add` with the row taken from the device-side counter: no host value is baked into a captured graph."""
        i = self.pos_dev
        row = self._pack(obs, actions, rewards, episode_starts, values, log_probs, terminal_obs, truncated)
        self._row.index_copy_(0, i, row.unsqueeze(0))
        self.mpc_actions.index_copy_(0, i, mpc_actions.to(self.mpc_actions.dtype).unsqueeze(0))
        self.pos_dev.add_(1)

    @torch.no_grad()
    def bootstrap_truncated(self, value_fn):
        """rewards += gamma * V(terminal observation) where the episode was truncated, for the whole rollout in one
        evaluation of the value tower"""
        if self.terminal_obs is None:
            return
        T, B = self.rewards.shape
        tv = value_fn(self.terminal_obs.reshape((T * B,) + self.terminal_obs.shape[2:])).reshape(T, B)
        self.rewards += self.gamma * tv * self.truncated

    def compute_returns_and_advantage(self, last_values, dones):
        last_gae = torch.zeros_like(last_values)
        dones = dones.to(last_values.dtype)
        for step in reversed(range(self.n_steps)):
            if step == self.n_steps - 1:
                next_non_terminal, next_values = 1.0 - dones, last_values
            else:
                next_non_terminal, next_values = 1.0 - self.episode_starts[step + 1], self.values[step + 1]
            delta = self.rewards[step] + self.gamma * next_values * next_non_terminal - self.values[step]
            last_gae = delta + self.gamma * self.gae_lambda * next_non_terminal * last_gae
            self.advantages[step] = last_gae
        self.returns = self.advantages + self.values


class BatchedCollector:
    """`collect_rollouts` of the reference for B environments (see module docstring).

    version "v0": the policy's 1-d action is the reference-speed override (agents/ppo_mpc.py:410-414);
    version "v1": its first three components are the cost weights (agents/ppo_mpc.py:416-420).
    algorithm "ppo" clips the action to the Box(-1, 1) bounds before use and bootstraps truncated episodes with the
    value of the terminal observation (agents/ppo_mpc.py:399-407, 451-461); "a2c" does neither
    (agents/a2c_mpc.py:138-144).
    reset_mpc_on_done=False mirrors the reference, whose single MPC agent keeps its collision memory across
    episode boundaries; True forgets it (`mpc_reset_env_mask`) when an environment restarts.
    warm_start=True (not in the reference) starts every solve from the environment's previous solution advanced by
    one stage (`MPC_FLAG_WARM_START`).
    use_graph: replay the step as one captured hipGraph (default on a GPU; `False` runs it eagerly, ~100 launches from
    Python per step).
    """

    def __init__(self, env, policy: ActorCritic, engine, version: str = "v0", algorithm: str = "ppo",
                 n_steps: int = 64, gamma: float = 0.99, gae_lambda: float = 0.95,
                 default_weights=(1.0, 1.0, 1.0), collision_cost: bool = False, reset_mpc_on_done: bool = False,
                 gather_actions: bool = False, seed: int = 0, warm_start: bool = False,
                 use_graph: bool | None = None, throughput: bool = False, fused_glue: bool | None = None):
        if version not in ("v0", "v1") or algorithm not in ("ppo", "a2c"):
            raise ValueError("version must be v0|v1 and algorithm ppo|a2c")
        if version == "v1" and policy.action_dim < 3:
            raise ValueError("v1 needs an action of at least 3 components (speed, control, input_diff weights)")
        self.env, self.policy, self.engine = env, policy, engine
        self.version, self.algorithm = version, algorithm
        self.collision_cost, self.reset_mpc_on_done, self.gather_actions = collision_cost, reset_mpc_on_done, gather_actions
        self.warm_start = bool(warm_start)
        # several collectors stepping on their own streams (PipelinedCollector): MPC_FLAG_THROUGHPUT, the build of the solve
        # kernel for four resident waves per SIMD whatever this group's batch size (include/mpc_mi355x.h)
        self._mpc_kw = dict(throughput=True) if throughput else {}
        dev = env.device
        B = env.num_envs
        self.buffer = RolloutBuffer(n_steps, B, policy.action_dim, dev, gamma, gae_lambda, keep_terminal=algorithm == "ppo")
        self.default_weights = torch.tensor(default_weights, dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        self._last_obs = env.reset().clone()                 # persistent tensors, updated in place (graph replays)
        self._last_episode_starts = torch.ones(B, dtype=torch.float32, device=dev)
        # episode counters of the rollout in one tensor (one reduction + one add per step): finished, crashed, arrived
        # episodes and solves that did not converge
        counts = torch.zeros(5, dtype=torch.int64, device=dev)     # [4]: steps the record kernel refused (past the buffer's end)
        self._roll = dict(counts=counts, ep_done=counts[0], crashed=counts[1], arrived=counts[2], unconverged=counts[3],
                          dones=torch.zeros(B, dtype=torch.bool, device=dev))
        self.policy.refresh_fused()
        self.num_timesteps = 0
        self.last_mpc = None
        self._mpc_out = None
        self.gathered_actions = None
        self.gathered_status = None
        # fused_glue (None = whenever possible: fused HIP environment, SB3-shaped policy, real engine): policy forward + sample
        # + MPC inputs in ONE launch (mpc_policy_act) and buffer row + carry-over + counters in one (mpc_rollout_record)
        # instead of ~40 torch kernels - csrc/mpc_rollout_glue.hpp
        can_fuse = (dev.type == "cuda" and getattr(env, "backend", "") == "hip" and isinstance(policy, ActorCritic) and
                    hasattr(engine, "predict_batch_torch") and hasattr(engine, "_lib") and
                    policy.pi[0].in_features == VEHICLES_COUNT * 8 and 2 * policy.pi[0].out_features <= 256 and
                    policy.action_dim <= 8 and policy.log_std.dtype == torch.float32)
        if fused_glue and not can_fuse:
            raise ValueError("fused_glue needs the fused HIP environment, an ActorCritic policy (80 inputs, hidden <= 128) and the real engine")
        self.fused_glue = can_fuse if fused_glue is None else bool(fused_glue)
        if self.fused_glue:
            A = policy.action_dim
            z = lambda *sh, dt=torch.float32: torch.zeros(sh, dtype=dt, device=dev)
            # noise: the Gaussian sample's draws of the last step, made by the kernel itself (counter-based, keyed by seed,
            # global environment id and `step`, the number of policy steps taken so far, which mpc_rollout_record advances)
            self._fg = dict(act=z(B, A), val=z(B), logp=z(B), w=z(B, 3, dt=torch.float64), rs=z(B, dt=torch.float64),
                            ticket=z(1, dt=torch.int32), noise=z(B, A), step=z(1, dt=torch.int64))
            self._noise_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        # None = the default path: on a GPU with the real engine a step (policy -> MPC -> environment -> buffer row) is
        # captured once as a hipGraph and replayed; eager on the CPU, with stand-in engines, and when the actions are
        # all-gathered (the collective stays outside the graph)
        if use_graph is None:
            use_graph = dev.type == "cuda" and hasattr(engine, "reserve_envs") and hasattr(engine, "predict_batch_torch")
        self.use_graph = bool(use_graph)
        self._graph = None
        self.graph_fallback_reason = None
        if self.use_graph:
            if not gather_actions:
                self._capture()
            else:
                # sharded runs (config 5): the RCCL all-gather of actions + status is captured INSIDE the graph, so they keep
                # the one-launch step.  Should this RCCL / torch build refuse to capture a collective, the collector RAISES
                # (graph_fallback_reason says why); only with MPC_ALLOW_EAGER_COLLECTIVE_FALLBACK=1 does it step eagerly
                # instead.  The abort is PER RANK: a rank whose capture fails raises here, before any collective of the
                # rollout is issued, and its process ends; the other ranks block in their first all-gather until the process
                # group's timeout or the launcher (torchrun) tears the job down - the same behaviour as any rank dying.
                try:
                    self._capture()
                except Exception as e:      # noqa: BLE001 - whatever the capture raised is the reason reported
                    torch.cuda.synchronize(dev)
                    self._graph, self.use_graph = None, False
                    self.graph_fallback_reason = f"{type(e).__name__}: {e}"
                    # _capture() has put the collector's state back (its finally clause).  What it cannot vouch for is the
                    # communicator an aborted capture of a collective leaves behind: unless the caller asked for the eager
                    # fallback explicitly, a sharded run ends here instead of issuing collectives on it
                    if os.environ.get("MPC_ALLOW_EAGER_COLLECTIVE_FALLBACK", "0") != "1":
                        raise RuntimeError("hipGraph capture of the sharded rollout step failed (" + self.graph_fallback_reason +
                                           "); set MPC_ALLOW_EAGER_COLLECTIVE_FALLBACK=1 to step eagerly instead") from e

    def mpc_inputs(self, actions):
        """RL action -> (weights[B,3] float64, ref_speed[B] float64 or None) as the reference maps them."""
        clipped = torch.clamp(actions, -1.0, 1.0) if self.algorithm == "ppo" else actions
        if self.version == "v0":
            return self.default_weights, clipped[:, 0].to(torch.float64).contiguous()
        return clipped[:, :3].to(torch.float64).contiguous(), None

    # ---- one rollout = begin, n steps, finish; split so that the step can be captured in a hipGraph and several
    #      collectors can be interleaved (PipelinedCollector)
    @torch.no_grad()
    def _begin_rollout(self):
        self.policy.eval()
        self.policy.refresh_fused()        # the parameters may have been updated since the last rollout
        self.buffer.reset()
        self._roll["counts"].zero_()

    @torch.no_grad()
    def _rollout_step_fused(self):
        """The same step with the two glue kernels: identical data flow; the float32 sums of the three small matrix products
        are accumulated in index order instead of hipBLASLt's, and the Gaussian sample's noise is drawn by the kernel
        (counter-based; left in self._fg["noise"]) instead of by torch's generator - fed the same noise, ActorCritic.act
        returns the same actions (tests/test_predict_gpu.py::test_fused_glue_step_equals_the_torch_step)."""
        import ctypes
        lib, env, pol, buf, fg = self.engine._lib, self.env, self.policy, self.buffer, self._fg
        f, A, B = pol._fz, pol.action_dim, env.num_envs
        H2 = f["b1"].numel()
        dev = env.device
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        obs = self._last_obs
        v1 = self.version == "v1"
        rc = lib.mpc_policy_act(dev.index, B, A, H2, p(obs), p(f["w1"]), p(f["b1"]), p(f["w2"]), p(f["b2"]), p(f["wh"]),
                                p(f["bh"]), p(f["std"]), p(f["c0"]), p(fg["noise"]), self._noise_seed,
                                int(getattr(env, "env_offset", 0)), p(fg["step"]), 1 if v1 else 0,
                                1 if self.algorithm == "ppo" else 0, p(fg["act"]), p(fg["val"]), p(fg["logp"]),
                                p(fg["w"]) if v1 else None, None if v1 else p(fg["rs"]), stream)
        if rc != 0:
            raise RuntimeError(f"mpc_policy_act failed ({rc}): {lib.mpc_last_error().decode()}")
        weights, ref_speed = (fg["w"], None) if v1 else (self.default_weights, fg["rs"])
        self._mpc_out = self.engine.predict_batch_torch(obs, weights, ref_speed, collision_cost=self.collision_cost,
                                                        warm_start=self.warm_start, out=self._mpc_out, **self._mpc_kw)
        self.last_mpc = self._mpc_out
        mpc_action = self.last_mpc["act"]
        if self.gather_actions:
            from . import sharding
            self.gathered_actions, self.gathered_status = sharding.all_gather_results(mpc_action, self.last_mpc["status"])
        env.step(mpc_action)
        o = env._out
        if self.reset_mpc_on_done:
            self.engine.reset_env_mask_torch(o["done"])
        elif self.warm_start:
            self.engine.reset_env_mask_torch(o["done"], warm_only=True)
        keep = buf.terminal_obs is not None
        rc = lib.mpc_rollout_record(dev.index, buf.n_steps, B, A, buf._cols, 1 if keep else 0, p(buf._row), p(buf.mpc_actions), p(buf.pos_dev),
                                    p(fg["ticket"]), p(self._last_obs), p(self._last_episode_starts), p(fg["act"]), p(fg["val"]),
                                    p(fg["logp"]), p(mpc_action), p(self.last_mpc["status"]), p(o["obs"]), p(o["reward"]),
                                    p(o["done"]), p(o["terminal_obs"]) if keep else None, p(o["truncated"]) if keep else None,
                                    p(o["crashed"]), p(o["arrived"]), p(self._roll["counts"]), p(self._roll["dones"]),
                                    p(fg["step"]), stream)
        if rc != 0:
            raise RuntimeError(f"mpc_rollout_record failed ({rc}): {lib.mpc_last_error().decode()}")

    @torch.no_grad()
    def _rollout_step(self, device_pos: bool = False):
        if self.fused_glue:
            self._rollout_step_fused()
            if not device_pos:
                self.buffer.pos += 1
            return
        obs = self._last_obs
        # (noise_feed: a list of per-step draws to consume instead of the generator's - how the tests give this path the
        # draws the glue kernel made)
        feed = getattr(self, "noise_feed", None)
        actions, values, log_probs = self.policy.act(obs, generator=self.gen, noise=feed.pop(0) if feed else None)
        weights, ref_speed = self.mpc_inputs(actions)
        self._mpc_out = self.engine.predict_batch_torch(obs, weights, ref_speed, collision_cost=self.collision_cost,
                                                        warm_start=self.warm_start, out=self._mpc_out, **self._mpc_kw)
        self.last_mpc = self._mpc_out
        mpc_action = self.last_mpc["act"]
        if self.gather_actions:
            from . import sharding
            # every rank sees every environment's action and whether its solve converged
            self.gathered_actions, self.gathered_status = sharding.all_gather_results(mpc_action, self.last_mpc["status"])
        new_obs, rewards, dones, info = self.env.step(mpc_action)
        # (PPO, agents/ppo_mpc.py:451-461: a truncated episode's last reward is bootstrapped with the value of its terminal
        # observation - kept here, applied for the whole rollout by _finish_rollout)
        term = dict(terminal_obs=info["terminal_obs"], truncated=info["truncated"]) if self.algorithm == "ppo" else {}
        if self.reset_mpc_on_done:
            self.engine.reset_env_mask_torch(dones.view(torch.uint8))
        elif self.warm_start:            # a new episode must not start from the old one's plan
            self.engine.reset_env_mask_torch(dones.view(torch.uint8), warm_only=True)
        if device_pos:
            self.buffer.add_at_device_pos(obs, actions, rewards, self._last_episode_starts, values, log_probs, mpc_action,
                                          **term)
        else:
            self.buffer.add(obs, actions, rewards, self._last_episode_starts, values, log_probs, mpc_action, **term)
        self._last_obs.copy_(new_obs)
        self._last_episode_starts.copy_(dones)
        r = self._roll
        r["dones"].copy_(dones)
        # like the reference (agents/pure_mpc.py:303-305) an unconverged solve still acts with its last iterate; the
        # collector counts them so that a training run can see what fraction of its actions that was
        st = self.last_mpc["status"]
        r["counts"][:4] += torch.stack((dones, info["crashed"], info["arrived"], (st != 0) & ((st < 5) | (st > 7)))).sum(dim=1)

    def _capture(self):
        """Capture one rollout step (policy -> mpc_predict_batch -> environment step -> buffer row) as a hipGraph: a step
        is ~100 small launches, ~2 ms of host time, which a replay replaces by one launch.  Every tensor the step reads or
        writes across steps lives at a fixed address (environment state, last observation, buffer, counters); the buffer
        row comes from a device-side counter; both random generators are registered with the graph."""
        dev = self.env.device
        env, B = self.env, self.env.num_envs
        if hasattr(self.engine, "reserve_envs"):     # the handle's per-environment buffers must not grow inside the capture
            self.engine.reserve_envs(B)
        # The warm-up and capture steps below really step the environment and the detector.  Everything they touch is
        # snapshotted here and put back afterwards, so that building the collector changes nothing the caller can see:
        # episodes, random streams (both generators, the fused environment's counters), the observation, and the engine's
        # detector records - including records a caller restored with mpc_set_env_state before building the collector
        # (resume).  A graph collector and an eager one with the same seeds therefore produce the same rollouts
        # (tests/test_predict_gpu.py::test_graph_and_eager_collectors_produce_the_same_rollout).
        names = [n for n in ("ego", "opos", "ospeed", "ohead", "oactive", "t", "rng_counter") if hasattr(env, n)]
        snap = {n: getattr(env, n).clone() for n in names}
        snap_obs, snap_starts = self._last_obs.clone(), self._last_episode_starts.clone()
        gen_states = [self.gen.get_state(), env.gen.get_state()]
        snap_step = self._fg["step"].clone() if self.fused_glue else None
        records = self.engine.save_env_state(B) if hasattr(self.engine, "save_env_state") else None
        side = torch.cuda.Stream(dev)
        try:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):                       # allocations and lazy initialisation happen outside the capture
                    self._rollout_step(device_pos=True)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            for gen in (self.gen, self.env.gen):
                g.register_generator_state(gen)
            with torch.cuda.graph(g, stream=side):
                self._rollout_step(device_pos=True)
            self._graph = g
        except BaseException as capture_error:
            # the capture FAILED: put the collector back as if it had never tried, and should the restore itself fail
            # (a synchronize or a state load after an invalidated capture), chain it so that the capture's error stays visible
            try:
                self._restore_after_capture(names, snap, snap_obs, snap_starts, gen_states, snap_step, records)
            except Exception as restore_error:      # noqa: BLE001
                raise restore_error from capture_error
            raise
        else:
            self._restore_after_capture(names, snap, snap_obs, snap_starts, gen_states, snap_step, records)

    def _restore_after_capture(self, names, snap, snap_obs, snap_starts, gen_states, snap_step, records):
        """Put back what the warm-up / capture steps of _capture() touched (in place: a graph replays against these addresses)."""
        env = self.env
        torch.cuda.synchronize(env.device)
        self.buffer.reset()
        for n in names:
            getattr(env, n).copy_(snap[n])
        self._last_obs.copy_(snap_obs)
        self._last_episode_starts.copy_(snap_starts)
        self.gen.set_state(gen_states[0])
        env.gen.set_state(gen_states[1])
        if snap_step is not None:
            self._fg["step"].copy_(snap_step)
        if records is not None:
            self.engine.load_env_state(records)       # also forgets the warm-start memory the warm-up steps left
        self._roll["counts"].zero_()
        self._roll["dones"].zero_()

    def _step(self):
        if self._graph is not None:
            self._graph.replay()
            self.buffer.pos += 1
        else:
            self._rollout_step()
        self.num_timesteps += self.env.num_envs

    @torch.no_grad()
    def _finish_rollout(self):
        last_values = self.policy.predict_values(self._last_obs)
        if not self.fused_glue:
            self.buffer.bootstrap_truncated(self.policy.predict_values)
            self.buffer.compute_returns_and_advantage(last_values, self._roll["dones"])
            return
        # the same arithmetic in one launch (mpc_rollout_finish) instead of 11 torch kernels per step of the rollout
        import ctypes
        buf, dev, B = self.buffer, self.env.device, self.env.num_envs
        T = buf.n_steps
        tv = None
        if buf.terminal_obs is not None:
            tv = self.policy.predict_values(buf.terminal_obs.reshape((T * B,) + buf.terminal_obs.shape[2:])).contiguous()
        last_values = last_values.contiguous()
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        rc = self.engine._lib.mpc_rollout_finish(dev.index, T, B, self.policy.action_dim, buf._cols,
                                                 1 if buf.terminal_obs is not None else 0, p(buf._row), p(last_values),
                                                 p(self._roll["dones"]), p(tv), buf.gamma, buf.gae_lambda, p(buf.advantages),
                                                 p(buf.returns), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"mpc_rollout_finish failed ({rc}): {self.engine._lib.mpc_last_error().decode()}")

    def _rollout_stats(self, n):
        r = self._roll
        # counts[4]: steps mpc_rollout_record refused because the buffer was already full (the torch path raises IndexError
        # there); read back with the other counters at the end of the rollout and never silent
        refused = int(r["counts"][4])
        if refused:
            raise IndexError(f"{refused} rollout steps were recorded past the end of the buffer ({self.buffer.n_steps} steps)")
        return dict(steps=n * self.env.num_envs, episodes=int(r["ep_done"]), crashed=int(r["crashed"]),
                    arrived=int(r["arrived"]), mpc_unconverged=int(r["unconverged"]), refused_steps=refused)

    def collect_rollouts(self, n_rollout_steps: int | None = None):
        n = self.buffer.n_steps if n_rollout_steps is None else int(n_rollout_steps)
        assert n == self.buffer.n_steps
        self._begin_rollout()
        for _ in range(n):
            self._step()
        self._finish_rollout()
        return self._rollout_stats(n)

    def flat_buffer(self):
        """(obs, actions, log_probs, advantages, returns) with time and environment flattened, for the update"""
        b = self.buffer
        f = lambda t: t.reshape((-1,) + t.shape[2:])
        return f(b.obs), f(b.actions), f(b.log_probs), f(b.advantages), f(b.returns)

    def mean_reward(self):
        return float(self.buffer.rewards.mean())


class PipelinedCollector:
    """Several BatchedCollectors (disjoint environment groups, one engine handle each, one shared policy) stepped
    round-robin on their own HIP streams.  A batched MPC solve ends with a long tail - a few slow instances on an
    otherwise idle GPU - and the groups are independent, so the tail of one group's solve overlaps
    with the policy / preamble / bulk of the others'.  Pays only with `use_graph=True` collectors: in eager mode the host
    needs ~2 ms to launch one group's step and the groups serialise on that.  Same data as running the collectors one
    after the other (each group's chain of operations is unchanged and ordered on its stream); the update sees the
    concatenated buffers."""

    def __init__(self, collectors):
        if not collectors:
            raise ValueError("need at least one collector")
        self.collectors = list(collectors)
        c0 = self.collectors[0]
        if any(c.policy is not c0.policy or c.algorithm != c0.algorithm or c.buffer.n_steps != c0.buffer.n_steps
               for c in self.collectors):
            raise ValueError("the collectors must share the policy, the algorithm and the rollout length")
        if len({id(c.engine) for c in self.collectors}) != len(self.collectors):
            raise ValueError("every collector needs its own engine handle (calls on one handle must not overlap)")
        self.policy, self.algorithm, self.env = c0.policy, c0.algorithm, c0.env
        # streams that really overlap (two that share a hardware queue serialise; engine.concurrent_streams probes each): with
        # more groups than the runtime has queues the rest are taken as they come
        from . import engine as _engine
        dev = c0.env.device
        try:
            self.streams = _engine.concurrent_streams(len(self.collectors), dev)
        except _engine.EngineError:
            self.streams = [torch.cuda.Stream(dev) for _ in self.collectors]

    @property
    def num_timesteps(self):
        return sum(c.num_timesteps for c in self.collectors)

    def collect_rollouts(self, n_rollout_steps: int | None = None):
        n = self.collectors[0].buffer.n_steps if n_rollout_steps is None else int(n_rollout_steps)
        assert all(n == c.buffer.n_steps for c in self.collectors), "a rollout is exactly one buffer of steps"
        cur = torch.cuda.current_stream(self.env.device)
        for s in self.streams:
            s.wait_stream(cur)                      # the groups see everything enqueued so far (policy update, resets)
        for c, s in zip(self.collectors, self.streams):
            with torch.cuda.stream(s):
                c._begin_rollout()
        for _ in range(n):
            for c, s in zip(self.collectors, self.streams):
                with torch.cuda.stream(s):
                    c._step()
        for c, s in zip(self.collectors, self.streams):
            with torch.cuda.stream(s):
                c._finish_rollout()
        for s in self.streams:
            cur.wait_stream(s)                      # ... and the update sees every group's buffer
        stats = [c._rollout_stats(n) for c in self.collectors]
        return {k: sum(d[k] for d in stats) for k in stats[0]}

    def flat_buffer(self):
        parts = [c.flat_buffer() for c in self.collectors]
        return tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(5))

    def mean_reward(self):
        return float(torch.cat([c.buffer.rewards.reshape(-1) for c in self.collectors]).mean())


class OnPolicyTrainer:
    """collect -> update loop of the reference's MPC-RL agents for B environments (`PPO_MPC.learn/train`
    agents/ppo_mpc.py:218-330, `A2C_MPC.train` agents/a2c_mpc.py:182-226; both are stable-baselines3's on-policy
    algorithms with the MPC call inside `collect_rollouts`).  The update is what those `train()` methods compute:

      ppo  n_epochs passes over the buffer in shuffled minibatches; clipped surrogate
           -mean(min(A r, A clip(r, 1-eps, 1+eps))) with r = exp(logp - logp_old), advantages normalised per minibatch,
           + vf_coef * mse(returns, V) + ent_coef * (-mean entropy); Adam, gradient-norm clip
           (defaults of config/cfg.yaml:66-86: lr 3e-4, 10 epochs, clip 0.2, gamma 0.99, lambda 0.95)
      a2c  one step on the whole buffer: -mean(A logp) + vf_coef * mse + ent_coef * (-mean entropy); RMSprop
           (alpha 0.99, eps 1e-5), no advantage normalisation, lambda 1 (config/cfg.yaml:31-61)
    """

    def __init__(self, collector, learning_rate: float | None = None, n_epochs: int = 10,
                 batch_size: int = 256, clip_range: float = 0.2, ent_coef: float = 0.0, vf_coef: float = 0.5,
                 max_grad_norm: float = 0.5, normalize_advantage: bool | None = None, seed: int = 0):
        self.col = collector
        self.algorithm = collector.algorithm
        pol = collector.policy
        if self.algorithm == "ppo":
            self.opt = torch.optim.Adam(pol.parameters(), lr=3e-4 if learning_rate is None else learning_rate, eps=1e-5)
            self.normalize_advantage = True if normalize_advantage is None else normalize_advantage
        else:
            self.opt = torch.optim.RMSprop(pol.parameters(), lr=7e-4 if learning_rate is None else learning_rate,
                                           alpha=0.99, eps=1e-5)
            self.normalize_advantage = False if normalize_advantage is None else normalize_advantage
        self.n_epochs, self.batch_size, self.clip_range = int(n_epochs), int(batch_size), float(clip_range)
        self.ent_coef, self.vf_coef, self.max_grad_norm = float(ent_coef), float(vf_coef), float(max_grad_norm)
        self.gen = torch.Generator(device=collector.env.device)
        self.gen.manual_seed(int(seed))
        self.n_updates = 0

    def _flat(self):
        return self.col.flat_buffer()

    def _loss(self, obs, actions, old_logp, adv, ret):
        values, logp, entropy = self.col.policy.evaluate_actions(obs, actions)
        if self.normalize_advantage and adv.numel() > 1:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        if self.algorithm == "ppo":
            ratio = torch.exp(logp - old_logp)
            pg = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1.0 - self.clip_range, 1.0 + self.clip_range)).mean()
        else:
            pg = -(adv * logp).mean()
        vl = torch.nn.functional.mse_loss(ret, values)
        el = -entropy.mean()
        return pg + self.ent_coef * el + self.vf_coef * vl, pg, vl, el

    def train(self):
        """One update from the collector's (full) buffer; returns the mean losses."""
        pol = self.col.policy
        pol.train()
        obs, actions, old_logp, adv, ret = self._flat()
        n = obs.shape[0]
        stats = []
        if self.algorithm == "ppo":
            for _ in range(self.n_epochs):
                perm = torch.randperm(n, generator=self.gen, device=obs.device)
                for lo in range(0, n, self.batch_size):
                    idx = perm[lo:lo + self.batch_size]
                    loss, pg, vl, el = self._loss(obs[idx], actions[idx], old_logp[idx], adv[idx], ret[idx])
                    self.opt.zero_grad()
                    loss.backward()
                    torch.nn.utils.clip_grad_norm_(pol.parameters(), self.max_grad_norm)
                    self.opt.step()
                    stats.append(torch.stack([loss.detach(), pg.detach(), vl.detach(), el.detach()]))
                self.n_updates += 1
        else:
            loss, pg, vl, el = self._loss(obs, actions, old_logp, adv, ret)
            self.opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(pol.parameters(), self.max_grad_norm)
            self.opt.step()
            stats.append(torch.stack([loss.detach(), pg.detach(), vl.detach(), el.detach()]))
            self.n_updates += 1
        m = torch.stack(stats).mean(dim=0).tolist()
        return dict(loss=m[0], policy_loss=m[1], value_loss=m[2], entropy_loss=m[3])

    def learn(self, total_timesteps: int):
        """collect_rollouts / train until `total_timesteps` environment steps were taken (BaseAlgorithm.learn)."""
        log = []
        while self.col.num_timesteps < total_timesteps:
            roll = self.col.collect_rollouts()
            upd = self.train()
            log.append(dict(roll, **upd, timesteps=self.col.num_timesteps,
                            mean_reward=self.col.mean_reward()))
        return log
