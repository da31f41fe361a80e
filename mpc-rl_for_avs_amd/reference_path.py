"""Global reference path of the ego vehicle (host side of the MPC hot path).

Mirrors `Agent.reference_states` (reference agents/base_agent.py:118-154): an 85x4 table with
columns [x, y, v, heading]: 40 points straight down from (2, 50) at v*dt spacing, a 20-step
quarter turn (heading -pi/2 -> -pi) and 25 points straight in -x; v = 10 everywhere.
The reference rebuilds the table on every property access; here it is built once per dt.
"""
from __future__ import annotations

import functools

import numpy as np

NUM_REF_POINTS = 85


@functools.lru_cache(maxsize=8)
def _table(dt: float) -> np.ndarray:
    pts = np.empty((NUM_REF_POINTS, 4), dtype=np.float64)
    x, y, v, heading = 2.0, 50.0, 10.0, -np.pi / 2
    i = 0
    for _ in range(40):
        y += v * dt * np.sin(heading)
        pts[i] = (x, y, v, heading); i += 1
    step = (np.pi / 2) / 20
    for _ in range(20):
        heading -= step
        x += v * dt * np.cos(heading)
        y += v * dt * np.sin(heading)
        pts[i] = (x, y, v, heading); i += 1
    for _ in range(25):
        x += v * dt * np.cos(heading)
        pts[i] = (x, y, v, heading); i += 1
    pts.setflags(write=False)
    return pts


def reference_states(dt: float = 0.1) -> np.ndarray:
    """Fresh writable copy of the [85, 4] table (callers in the reference mutate their copy)."""
    return _table(float(dt)).copy()


def nearest_index(ref_xy: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """argmin_i |pos - ref_i| for a batch of positions [B, 2] (reference agents/pure_mpc.py:106-109).

    Distances are formed exactly like the reference's `np.linalg.norm(position - point)`:
    float32 positions minus float64 points -> float64, first minimum wins (np.argmin)."""
    d = pos[:, None, :].astype(np.float64) - ref_xy[None, :, :]
    return np.argmin(np.sqrt(np.sum(d * d, axis=2)), axis=1).astype(np.int32)
