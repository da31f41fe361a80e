"""Generates tests/golden/closed_loop_ipopt.npz: problem data recorded from CLOSED-LOOP runs (the states an MPC-driven
ego actually visits, as opposed to the uniform synthetic draw of synth.solver_inputs) and their solutions by
oracle/ipopt_restated.py - the closest thing to the reference's CasADi/IPOPT that can run here - at the REFERENCE's
solver settings (ipopt tol 1e-6, max_iter 1000: agents/pure_mpc.py:294-295).

Run from the repository root:  python tests/golden/make_closed_loop.py [--keep-states]     (a few minutes on 8 cores, CPU only)

Scenarios (all on the CPU: numpy mirror of the agent's preamble, tests/host_preamble.py, with the C oracle as its solver,
driving mpc-rl_for_avs_amd/rollout.SyntheticIntersectionEnv):
  c1     BASELINE configs[0]: one ego, one other vehicle, the reference's stand-alone loop (main/run_pure_mpc.py:10-40),
         live objective, 4 episodes from different spawn offsets
  c1cc   the same with the collision-cost term on (agents/archive/pure_mpc.py:189-206)
  c4     BASELINE configs[3] shape: 48 environments x 16 steps, 4 other vehicles, v0 (RL action = reference speed,
         untrained policy, clipped to [-1, 1] like agents/ppo_mpc.py:399-407)
  c4mpc  the same environments driven by the MPC alone (no RL override): the speeds / crossings of a trained agent
  c4cc   c4mpc with the collision-cost term on
  c4v1   the v1 INPUT DOMAIN (round 4): the same environments with the RL action = the three cost weights, clipped to
         [-1, 1] as PPO's Box(-1, 1) action space makes them (agents/ppo_mpc.py:399-407, 416-420) - i.e. NEGATIVE cost weights
         half of the time: drawn uniformly from [-1, 1]^3 per environment and step, every fifth environment all-negative
Every step's problem data (state, ego_index, vref, weights, is_collide, others) is one instance.  A seeded subsample
keeps the file small.  Stored per scenario: the problem data, `u0/U/X/status/iters/kkt` of the independent solver, and
`oracle_u0/oracle_status/oracle_iters` of oracle/mpc_oracle.c (tol 1e-8) for the classification in
tools/parity_vs_ipopt.py.  status of the independent solver: 0 converged, 1 iteration limit, 2 inertia correction
failed, 5 restoration failed, 6 restoration converged to a point of local infeasibility (round 4: IPOPT's restoration
phase IS restated, oracle/ipopt_restated.py; until round 3 status 5 meant "would enter restoration").  Also stored:
`n_resto`, the number of restoration phases the solve went through.
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import torch  # noqa: E402
import mpc_rl_for_avs_amd  # noqa: E402,F401
from mpc_rl_for_avs_amd import rollout  # noqa: E402
from mpc_rl_for_avs_amd.reference_path import reference_states  # noqa: E402
import nlp_batch as nb  # noqa: E402
import ipopt_restated as ipr  # noqa: E402
import oracle_lib  # noqa: E402
from host_preamble import HostPreambleAgent  # noqa: E402

REF = reference_states(0.1)
CFG = dict(horizon=20, render=False, ttc_threshold=3, weight_speed=1, weight_control=1, weight_input_diff=1,
           speed_override=0, weight_distance=10.0, weight_collision=1.0)
KEEP = {"c1": 160, "c1cc": 160, "c4": 160, "c4mpc": 160, "c4cc": 160, "c4v1": 160}


class Env:
    config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}
    unwrapped = None


Env.unwrapped = Env


class OracleEngine:
    """The C oracle behind the engine interface the host mirror drives (solver of the closed loop, tol 1e-8)."""

    def solve_batch(self, state, ego_index, weights, is_collide, vref=None, others=None, collision_cost=False,
                    want_trajectories=False):
        return oracle_lib.solve_batch(REF, state, ego_index, weights, is_collide, vref=vref, others=others,
                                      collision_cost=collision_cost, max_iter=200, xy_bounds=False, nthreads=8)


def record(n_env, n_others, steps, cc, mode, seed):
    """mode 'mpc': default weights, no override; 'v0': reference speed from an untrained policy."""
    env = rollout.SyntheticIntersectionEnv(n_env, device="cpu", seed=seed, n_others=n_others)
    agent = HostPreambleAgent(Env, dict(CFG), engine=OracleEngine(), collision_cost=cc)
    pol = rollout.ActorCritic(1)
    gen = torch.Generator().manual_seed(seed)
    wrng = np.random.default_rng(1000 + seed)
    obs = env.reset()
    rows = []
    for _ in range(steps):
        o = obs.numpy().astype(np.float32)
        rs = None
        if mode == "v0":
            with torch.no_grad():
                a, _, _ = pol(obs, generator=gen)
            rs = torch.clamp(a, -1.0, 1.0)[:, :1].to(torch.float64).numpy()
        w = None
        if mode == "v1":
            w = wrng.uniform(-1.0, 1.0, (n_env, 3))
            w[::5] = -np.abs(w[::5])
        act = agent.predict_batch_host(o, w, rs)
        inp = agent.last_inputs
        oth = np.zeros((n_env, n_others, 4))
        oth[:, :, :2] = 1e6                       # absent vehicles are parked far away, as the host mirror does
        if inp["others"] is not None:
            got = np.asarray(inp["others"])
            oth[:, :got.shape[1]] = got[:, :n_others]
        for b in range(n_env):
            rows.append(dict(state=inp["state"][b].copy(), ego_index=int(inp["ego_index"][b]), vref=inp["vref"][b].copy(),
                             weights=inp["weights"][b].copy(), is_collide=int(inp["is_collide"][b]), others=oth[b].copy()))
        obs, _, done, _ = env.step(torch.as_tensor(act, dtype=torch.float64))
        ids = np.nonzero(done.numpy())[0]
        if ids.size:
            agent.reset_env_state([int(i) for i in ids])
    return rows


def scenario(name):
    if name in ("c1", "c1cc"):
        rows = []
        for ep in range(4):
            rows += record(1, 1, 110, name == "c1cc", "mpc", seed=ep)
        return rows
    if name == "c4":
        return record(48, 4, 16, False, "v0", seed=7)
    if name == "c4mpc":
        return record(48, 4, 16, False, "mpc", seed=8)
    if name == "c4cc":
        return record(48, 4, 16, True, "mpc", seed=9)
    if name == "c4v1":
        return record(48, 4, 16, False, "v1", seed=10)
    raise KeyError(name)


def pack(rows):
    return dict(state=np.array([r["state"] for r in rows]), ego_index=np.array([r["ego_index"] for r in rows], np.int32),
                vref=np.array([r["vref"] for r in rows]), weights=np.array([r["weights"] for r in rows]),
                is_collide=np.array([r["is_collide"] for r in rows], np.uint8), others=np.array([r["others"] for r in rows]))


def _one(args):
    d, cc, b = args
    p = nb.Batch.build(REF, d["state"][b:b + 1], d["ego_index"][b:b + 1], d["weights"][b:b + 1], d["is_collide"][b:b + 1],
                       vref=d["vref"][b:b + 1], others=d["others"][b:b + 1], collision_cost=cc)
    r = ipr.solve(p, tol=1e-6, max_iter=1000, sf_min=1e-2)          # the reference's settings, pure_mpc.py:294-295
    return r["U"], r["X"], r["status"], r["iters"], r["kkt"], r["n_resto"]


def main():
    # --keep-states (round 5): the problem data already in the file stays - the closed-loop states recorded with round 4's
    # engine - and only the two solvers' answers are recomputed, so that a change of the engine's globalisation is compared
    # on the SAME 960 instances (tools/parity_vs_ipopt.py) instead of on the slightly different states its own closed loop
    # would visit (a different local minimiser taken at one step changes the rest of the episode).
    keep_states = "--keep-states" in sys.argv
    path = os.path.join(ROOT, "tests", "golden", "closed_loop_ipopt.npz")
    old = np.load(path) if keep_states else None
    out = {}
    for name, keep in KEEP.items():
        cc = name.endswith("cc")
        if keep_states:
            d = {k: old[f"{name}_{k}"] for k in ("state", "ego_index", "vref", "weights", "is_collide", "others")}
            rows = sel = range(d["state"].shape[0])
        else:
            rows = scenario(name)
            rng = np.random.default_rng(len(name))
            sel = np.sort(rng.choice(len(rows), size=min(keep, len(rows)), replace=False))
            d = pack([rows[i] for i in sel])
        with Pool(min(8, os.cpu_count() or 1)) as pool:
            res = pool.map(_one, [(d, cc, b) for b in range(len(sel))])
        orc = oracle_lib.solve_batch(REF, d["state"], d["ego_index"], d["weights"], d["is_collide"], vref=d["vref"],
                                     others=d["others"], collision_cost=cc, max_iter=1000, xy_bounds=False, nthreads=8)
        for k, v in d.items():
            out[f"{name}_{k}"] = v
        out[f"{name}_U"] = np.array([r[0] for r in res])
        out[f"{name}_X"] = np.array([r[1] for r in res])
        out[f"{name}_u0"] = out[f"{name}_U"][:, 0].copy()
        out[f"{name}_status"] = np.array([np.ravel(r[2])[0] for r in res], dtype=np.int32)
        out[f"{name}_iters"] = np.array([np.ravel(r[3])[0] for r in res], dtype=np.int32)
        out[f"{name}_kkt"] = np.array([np.ravel(r[4])[0] for r in res])
        out[f"{name}_n_resto"] = np.array([int(r[5]) for r in res], dtype=np.int32)
        out[f"{name}_oracle_u0"] = orc["u0"]
        out[f"{name}_oracle_status"] = orc["status"]
        out[f"{name}_oracle_iters"] = orc["iters"]
        st = out[f"{name}_status"]
        ok = ((st == 0) | (st == 3)) & ((orc["status"] == 0) | ((orc["status"] >= 5) & (orc["status"] <= 7)))
        err = np.abs(orc["u0"] - out[f"{name}_u0"]).max(axis=1) / np.maximum(1.0, np.abs(out[f"{name}_u0"]).max(axis=1))
        print(f"{name}: {len(rows)} recorded, {len(sel)} kept; independent solver status histogram "
              f"{np.bincount(st, minlength=7).tolist()}, went through restoration {int((out[f'{name}_n_resto'] > 0).sum())}, iterations mean {out[f'{name}_iters'].mean():.1f} max "
              f"{out[f'{name}_iters'].max()}; oracle status {np.bincount(orc['status'], minlength=6).tolist()}; both converged "
              f"{int(ok.sum())}, of those within 1e-4: {int((err[ok] <= 1e-4).sum())}, max {err[ok].max():.2e}", flush=True)
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
