"""Times the REFERENCE's own solve path (PureMPC_Agent._solve incl. SX-graph build + nlpsol construction, which
the reference pays on every control step, agents/pure_mpc.py:80-318) on synthetic observations.

Shipped but NOT RUN in this project: casadi / gymnasium / shapely are not installed in the build or GPU images
(SURVEY.md section 8c), so no CasADi number appears in BASELINE.md or bench.py.  Run it in an environment that has
the reference's requirements.txt installed:

    PYTHONPATH=/path/to/MPC-RL_for_AVs:/path/to/this/repo python tools/time_reference_casadi.py --n 200
"""
import argparse
import time

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--vehicles", type=int, default=4)
    a = ap.parse_args()
    try:
        from agents.pure_mpc import PureMPC_Agent        # the reference (needs casadi, gymnasium, shapely)
        from mpc_rl_for_avs_amd import synth
    except ImportError as e:
        raise SystemExit(f"not runnable here ({e}); see the module docstring: the reference checkout and this repository "
                         f"must be on PYTHONPATH and the reference's requirements installed")

    class Env:
        unwrapped = None
        config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}
    Env.unwrapped = Env
    cfg = dict(horizon=20, render=False, ttc_threshold=3, weight_speed=1, weight_control=1, weight_input_diff=1,
               speed_override=0)
    obs = synth.make_obs_batch(a.n, a.vehicles, seed=0)
    times, acts = [], []
    for b in range(a.n):
        agent = PureMPC_Agent(Env, cfg)                   # fresh agent: no collision memory carried over
        t0 = time.perf_counter()
        acts.append(agent.predict(obs[b]))
        times.append(time.perf_counter() - t0)
    t = np.array(times)
    print(f"reference CasADi/IPOPT path: {a.n} solves, mean {t.mean()*1e3:.1f} ms, median {np.median(t)*1e3:.1f} ms "
          f"-> {1.0/t.mean():.1f} solves/s on 1 core")
    np.save("reference_actions.npy", np.array(acts))     # compare with PureMPC_Agent.predict_batch of this repo


if __name__ == "__main__":
    main()
