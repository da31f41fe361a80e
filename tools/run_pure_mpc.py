#!/usr/bin/env python3
"""BASELINE config 1: the reference's stand-alone loop (main/run_pure_mpc.py:10-40: obs -> PureMPC_Agent.predict ->
env.step, one ego, one other vehicle) against the synthetic intersection environment, with the MPC solve on the
MI355X.  Prints the outcome and the per-step latency of `predict` (host preamble + one-instance solve)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Env:
    config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}


def run(steps=150, n_others=1, seed=0, horizon=20, verbose=True):
    import numpy as np
    import torch
    from mpc_rl_for_avs_amd import rollout
    from mpc_rl_for_avs_amd.pure_mpc import PureMPC_Agent
    cfg = dict(horizon=horizon, render=False, ttc_threshold=3, weight_speed=1, weight_control=1, weight_input_diff=1,
               speed_override=0)
    env = rollout.SyntheticIntersectionEnv(1, device="cpu", seed=seed, n_others=n_others)
    agent = PureMPC_Agent(_Env(), cfg)
    obs = env.reset()
    lat, log = [], []
    outcome = "running"
    for i in range(steps):
        o = obs[0].numpy()
        t0 = time.perf_counter()
        action = agent.predict(o, False)                       # MPC_Action, as the reference's loop asks for it
        lat.append(time.perf_counter() - t0)
        act = torch.tensor([[action.acceleration, action.steer]], dtype=torch.float64)
        log.append((float(o[0, 1]), float(o[0, 2]), float(np.hypot(o[0, 3], o[0, 4])), float(action.acceleration),
                    float(action.steer), bool(agent.is_collide), int(agent.last_solve["status"][0])))
        obs, reward, done, info = env.step(act)
        if bool(done[0]):
            outcome = "crashed" if bool(info["crashed"][0]) else ("arrived" if bool(info["arrived"][0]) else "timeout")
            break
    lat_ms = 1e3 * np.array(lat[1:] or lat)
    if verbose:
        print(f"{len(log)} steps, outcome {outcome}; predict latency ms: median {np.median(lat_ms):.2f} "
              f"p95 {np.percentile(lat_ms, 95):.2f}; converged {np.mean([r[6] == 0 for r in log]):.3f}")
    return outcome, log, lat_ms


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--others", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    run(a.steps, a.others, a.seed)
