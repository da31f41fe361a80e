#!/usr/bin/env python3
"""Evaluated and NOT adopted: a portfolio of solver settings for the slowest instances of a batch.

One batch of 4096 takes as long as its slowest instance (chain of iterations x time per iteration of a lone wave), and which
instance is slow is chaotic in the solver's settings.  Idea: cap a first launch at K1 iterations, then race the unfinished
instances under several settings (the base continued + cold starts with mu_init = 1 and with the step lengths 1, 1/2, 1/4,
1/8), first to converge wins.  This tool measures what that would buy, on the CPU with the oracle (same algorithm as the
kernel) and the two-rate occupancy model of tools/sim_schedule.py:  python tools/portfolio_study.py > profiles/rNN_portfolio_study.txt
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p_)
import numpy as np  # noqa: E402
import oracle_lib  # noqa: E402
import sim_schedule as sim  # noqa: E402
from mpc_rl_for_avs_amd import synth  # noqa: E402
from mpc_rl_for_avs_amd.reference_path import reference_states  # noqa: E402

REF = reference_states(0.1)
lib = oracle_lib._load()
lib.oracle_set_experiment.argtypes = [ctypes.c_double, ctypes.c_double]
VARIANTS = {"base": (0.1, 0.25), "mu_init 1": (1.0, 0.25), "backtracking 1/2": (0.1, 0.5), "mu_init 0.01": (0.01, 0.25)}
sim.T1, sim.TS = 38.0, 27.0          # lone wave of the 201-register build; a SIMD shared by several waves
MAX_ITER, GUARD = 1000, 64


def solve(inp, sel, variant, max_iter=MAX_ITER):
    lib.oracle_set_experiment(*VARIANTS[variant])
    try:
        return oracle_lib.solve_batch(REF, inp["state"][sel], inp["ego_index"][sel], inp["weights"][sel], inp["is_collide"][sel],
                                      vref=inp["vref"][sel], others=inp["others"][sel], collision_cost=True, max_iter=max_iter,
                                      xy_bounds=False, nthreads=8, stall_window=GUARD)
    finally:
        lib.oracle_set_experiment(0.1, 0.25)


def main():
    conv = lambda o: (o["status"] == 0) | (o["status"] == 5)
    print(f"# BASELINE config 3 (B = 4096, 8 vehicles, collision cost), tol 1e-8, max_iter {MAX_ITER}, stall_window {GUARD}; "
          f"times from the occupancy model (lone wave {sim.T1:.0f} us per iteration, shared SIMD {sim.TS:.0f} us)")
    for seed in (0, 1, 2):
        inp = synth.solver_inputs(4096, 8, seed=seed)
        base = solve(inp, slice(None), "base")
        c0 = base["iters"].astype(float)
        ok0 = conv(base)
        print(f"\nseed {seed}: one launch: converged {ok0.mean():.4f}, iterations p99 {np.percentile(c0, 99):.0f} max {c0.max():.0f} -> "
              f"{sim.dispatch(c0, 3) / 1e3:.2f} ms; capped at 100: converged {(ok0 & (c0 <= 100)).mean():.4f} -> "
              f"{sim.dispatch(np.minimum(c0, 100), 3) / 1e3:.2f} ms; capped at 60: {(ok0 & (c0 <= 60)).mean():.4f} -> "
              f"{sim.dispatch(np.minimum(c0, 60), 3) / 1e3:.2f} ms")
        for K1 in (24, 32, 40):
            unf = np.nonzero(~(ok0 & (c0 <= K1)))[0]
            eff = {"base": np.where(ok0[unf], c0[unf], np.inf)}
            for name in list(VARIANTS)[1:]:
                o = solve(inp, unf, name)
                eff[name] = np.where(conv(o), o["iters"] + K1, np.inf)
            for names in (["base", "mu_init 1", "backtracking 1/2"], list(VARIANTS)):
                e = np.min(np.stack([eff[n] for n in names], axis=1), axis=1)
                lost = int(np.isinf(e).sum())
                rem = np.where(np.isinf(e), GUARD + 100, e) - K1            # a lost instance ends at its guard
                t1 = sim.dispatch(np.minimum(c0, K1), 3)
                t2 = sim.dispatch(list(np.repeat(rem, len(names))), 64) if rem.size else 0.0
                print(f"   first launch capped at {K1}: {len(unf)} unfinished ({len(unf) / 40.96:.1f} %), {t1 / 1e3:.2f} ms; race of "
                      f"{len(names)} settings: remaining chain p50 {np.median(rem):.0f} p90 {np.percentile(rem, 90):.0f} max {rem.max():.0f} "
                      f"iterations, {t2 / 1e3:.2f} ms; total {(t1 + t2) / 1e3 + 0.02:.2f} ms, converged {1 - lost / 4096:.4f}")


if __name__ == "__main__":
    main()
