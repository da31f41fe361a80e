#!/usr/bin/env python3
"""The reference's stand-alone loop for the iterative-linear agent (run_pure_mpc_linear.py:12-49: obs ->
IterativeLinearMPC_Agent.predict -> env.step with the action normalised by 5 and pi/3) against the synthetic
intersection environment, with the QP solve on the MI355X.  Prints the outcome and the per-step latency of `predict`."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Env:
    config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}


def run(steps=100, n_others=1, seed=0, horizon=20, verbose=True):
    import numpy as np
    import torch
    from mpc_rl_for_avs_amd import rollout
    from mpc_rl_for_avs_amd.pure_mpc_linear import IterativeLinearMPC_Agent
    env = rollout.SyntheticIntersectionEnv(1, device="cpu", seed=seed, n_others=n_others)
    agent = IterativeLinearMPC_Agent(_Env(), dict(horizon=horizon, render=False))
    obs = env.reset()
    lat, log = [], []
    outcome = "running"
    for i in range(steps):
        o = obs[0].numpy()
        t0 = time.perf_counter()
        action = agent.predict(o, return_numpy=False)
        lat.append(time.perf_counter() - t0)
        # the synthetic environment takes physical units (m/s^2, rad), i.e. what the reference's loop divides by 5, pi/3
        act = torch.tensor([[action.acceleration, action.steer]], dtype=torch.float64)
        log.append((float(o[0, 1]), float(o[0, 2]), float(np.hypot(o[0, 3], o[0, 4])), float(action.acceleration),
                    float(action.steer), int(agent.last_solve["status"][0]), int(agent.last_solve["iters"][0])))
        obs, reward, done, info = env.step(act)
        if bool(done[0]):
            outcome = "crashed" if bool(info["crashed"][0]) else ("arrived" if bool(info["arrived"][0]) else "timeout")
            break
    lat_ms = 1e3 * np.array(lat[1:] or lat)
    if verbose:
        print(f"{len(log)} steps, outcome {outcome}; predict latency ms: median {np.median(lat_ms):.2f} "
              f"p95 {np.percentile(lat_ms, 95):.2f}; QP solved {np.mean([r[5] == 0 for r in log]):.3f}, "
              f"mean iterations {np.mean([r[6] for r in log]):.1f}; final speed {log[-1][2]:.2f} m/s")
    return outcome, log, lat_ms


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--others", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    run(a.steps, a.others, a.seed)
