"""Seeded synthetic MPC instances (SURVEY.md section 8d generator) for tests and bench.py.

highway-env is not available offline, so observations are drawn around the reference path:
ego scattered +-0.5 m / +-0.1 rad about a random reference point, speed U(0,12) with 5 % exactly 0,
other vehicles on the four approach lanes of the intersection at N(8,1) m/s, all rounded to
float32 like real `Kinematics` observations (reference config/config.py:10-26).
"""
from __future__ import annotations

import numpy as np

from .reference_path import reference_states, nearest_index

VEHICLES_COUNT = 10   # reference config/cfg.yaml:2  (observation rows)


def make_obs_batch(B: int, V: int, seed: int = 0, dt: float = 0.1) -> np.ndarray:
    """obs[B, 10, 8] float32, rows [presence, x, y, vx, vy, heading, sin_h, cos_h], row 0 = ego."""
    assert 0 <= V <= VEHICLES_COUNT - 1
    rng = np.random.default_rng(seed)
    ref = reference_states(dt)
    i = rng.integers(0, 80, size=B)
    h = ref[i, 3]
    e_para = rng.uniform(-0.5, 0.5, B)
    e_perp = rng.uniform(-0.5, 0.5, B)
    x = ref[i, 0] + np.cos(h) * e_para - np.sin(h) * e_perp
    y = ref[i, 1] + np.sin(h) * e_para + np.cos(h) * e_perp
    heading = h + rng.uniform(-0.1, 0.1, B)
    heading = (heading + np.pi) % (2 * np.pi) - np.pi
    speed = rng.uniform(0.0, 12.0, B)
    speed[rng.uniform(size=B) < 0.05] = 0.0
    obs = np.zeros((B, VEHICLES_COUNT, 8), dtype=np.float32)
    obs[:, 0, 0] = 1.0
    obs[:, 0, 1] = x
    obs[:, 0, 2] = y
    obs[:, 0, 3] = speed * np.cos(heading)
    obs[:, 0, 4] = speed * np.sin(heading)
    obs[:, 0, 5] = heading
    obs[:, 0, 6] = np.sin(heading)
    obs[:, 0, 7] = np.cos(heading)
    # approach lanes: heading 0 at y=+2 from the west, pi/2 at x=-2 from the north,
    # pi at y=-2 from the east, -pi/2 at x=+2 from the south (the ego's own lane)
    lane_h = np.array([0.0, np.pi / 2, np.pi, -np.pi / 2])
    for j in range(V):
        lane = rng.integers(0, 4, size=B)
        d = rng.uniform(5.0, 60.0, B)
        hh = lane_h[lane]
        lat = 2.0
        ox = -d * np.cos(hh) + lat * np.sin(hh) * np.where(lane % 2 == 0, -1.0, 1.0) * 0.0
        oy = -d * np.sin(hh)
        # lateral offset (drive on the right): heading 0 -> y=+2, pi -> y=-2, pi/2 -> x=-2, -pi/2 -> x=+2
        ox = ox + np.where(lane == 1, -lat, 0.0) + np.where(lane == 3, lat, 0.0)
        oy = oy + np.where(lane == 0, lat, 0.0) + np.where(lane == 2, -lat, 0.0)
        sp = np.maximum(rng.normal(8.0, 1.0, B), 0.0)
        obs[:, j + 1, 0] = 1.0
        obs[:, j + 1, 1] = ox
        obs[:, j + 1, 2] = oy
        obs[:, j + 1, 3] = sp * np.cos(hh)
        obs[:, j + 1, 4] = sp * np.sin(hh)
        obs[:, j + 1, 5] = hh
        obs[:, j + 1, 6] = np.sin(hh)
        obs[:, j + 1, 7] = np.cos(hh)
    return obs


def solver_inputs(B: int, V: int, seed: int = 0, N: int = 20, dt: float = 0.1,
                  collide_fraction: float = 0.5, rl_weights_fraction: float = 0.5) -> dict:
    """Solver-level inputs (the arguments of `mpc_solve_batch`) with a forced is_collide mix.

    The speed profile of colliding instances follows `update_reference_states`
    (reference agents/pure_mpc.py:694-716) for a conflict index drawn ahead of the ego."""
    obs = make_obs_batch(B, V, seed, dt)
    rng = np.random.default_rng(seed + 7919)
    ref = reference_states(dt)
    M = ref.shape[0]
    ego = obs[:, 0]
    heading = ego[:, 5].astype(np.float64)
    speed32 = np.sqrt(ego[:, 3] * ego[:, 3] + ego[:, 4] * ego[:, 4])  # float32 norm like np.linalg.norm
    state = np.stack([ego[:, 1].astype(np.float64), ego[:, 2].astype(np.float64), heading,
                      speed32.astype(np.float64)], axis=1)
    ego_index = nearest_index(ref[:, :2], ego[:, 1:3])
    is_collide = (rng.uniform(size=B) < collide_fraction).astype(np.uint8)
    weights = np.ones((B, 3))
    rl = rng.uniform(size=B) < rl_weights_fraction
    weights[rl] = rng.uniform(0.0, 1.0, size=(int(rl.sum()), 3))
    idx = np.minimum(ego_index[:, None] + np.arange(N + 1)[None, :], M - 1)
    vref = ref[idx, 2].copy()
    conflict = np.minimum(ego_index + rng.integers(3, 26, size=B), M - 1)
    for b in np.nonzero(is_collide)[0]:
        col = ref[:, 2].copy()
        stop = min(max(ego_index[b] + 1, conflict[b] - 5), M - 1)
        n = stop - ego_index[b]
        if n > 0:
            col[ego_index[b]:stop] = np.linspace(state[b, 3], 0.0, n)
            col[stop:] = 0.0
        vref[b] = col[idx[b]]
    others = np.zeros((B, max(V, 1), 4))
    if V > 0:
        o = obs[:, 1:V + 1]
        others[:, :V, 0] = o[:, :, 1]
        others[:, :V, 1] = o[:, :, 2]
        others[:, :V, 2] = np.sqrt(o[:, :, 3] * o[:, :, 3] + o[:, :, 4] * o[:, :, 4])
        others[:, :V, 3] = o[:, :, 5]
    return dict(obs=obs, state=state, ego_index=ego_index, vref=vref, weights=weights,
                is_collide=is_collide, others=others[:, :V] if V > 0 else None, N=N, dt=dt, V=V)
