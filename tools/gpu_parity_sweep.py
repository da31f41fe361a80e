#!/usr/bin/env python3
"""Wider randomized parity check of the engine against the CPU oracle than the test-suite sizes: several seeds, vehicle
counts and both objective variants (run on a GPU box; the oracle uses the host cores)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_lib
from mpc_rl_for_avs_amd import engine, synth
from mpc_rl_for_avs_amd.reference_path import reference_states

ref = reference_states()
eng = engine.MPCEngine(horizon=20, max_iter=100)
worst = 0.0
for seed in range(1, 7):
    for V, cc in ((1, False), (4, False), (4, True), (9, True)):
        inp = synth.solver_inputs(2048, V, seed=seed)
        got = eng.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                              others=inp["others"], collision_cost=cc)
        want = oracle_lib.solve_batch(ref, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                                      vref=inp["vref"], others=inp["others"], collision_cost=cc, max_iter=100,
                                      xy_bounds=False)
        both = (got["status"] == 0) & (want["status"] == 0)
        err = np.abs(got["u0"] - want["u0"]).max(axis=1) / np.maximum(1.0, np.abs(want["u0"]).max(axis=1))
        bad = int((err[both] > 1e-4).sum())
        worst = max(worst, float(np.percentile(err[both], 99)))
        print(f"seed {seed} V={V} cc={int(cc)}: both converged {both.mean():.4f}, status equal {(got['status'] == want['status']).mean():.4f}, "
              f"iterations equal {(got['iters'] == want['iters'])[both].mean():.4f}, beyond 1e-4: {bad}, p99 err {np.percentile(err[both], 99):.2e}, "
              f"gpu status {np.bincount(got['status'], minlength=5)}, finite {np.isfinite(got['u0']).all()}", flush=True)
print("worst p99", worst)
