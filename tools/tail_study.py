#!/usr/bin/env python3
"""Iteration-tail study on the CPU oracle (same algorithm as the kernel): the numbers VERDICT r4 item 1 is judged by.

config 3 (B = 4096, 8 vehicles, collision cost) seeds 0-7 at the default settings (cap 100, tol 1e-8) and seeds 0-2 at
the reference's (max_iter 1000, tol 1e-6); config 2 (bench.py's draw: B = 1024, V = 4, live objective); the live objective
at B = 4096; times from the two-rate occupancy model of tools/sim_schedule.py.
    python tools/tail_study.py [--quick] > profiles/rNN_tail_study.txt
Experimental knobs of the oracle are read from the environment by mpc_oracle.c itself (ORACLE_X_*), if any are compiled in.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p_)
import numpy as np  # noqa: E402
import oracle_lib  # noqa: E402
if os.environ.get('ORX'):          # an experimental build of the oracle
    oracle_lib._LIB_PATH = os.environ['ORX']
    oracle_lib.build = lambda force=False: oracle_lib._LIB_PATH
import sim_schedule as sim  # noqa: E402
from mpc_rl_for_avs_amd import synth  # noqa: E402
from mpc_rl_for_avs_amd.reference_path import reference_states  # noqa: E402

REF = reference_states(0.1)
T1 = float(os.environ.get("TAIL_T1", 32.3))     # lone-wave microseconds per iteration (profiles/r04_latency.txt)
TS = float(os.environ.get("TAIL_TS", 27.0))


def solve(inp, cc, max_iter, tol, stall=0):
    return oracle_lib.solve_batch(REF, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                                  others=inp["others"], collision_cost=cc, max_iter=max_iter, tol=tol, xy_bounds=False,
                                  nthreads=8, stall_window=stall)


def row(tag, o, cap):
    it = o["iters"].astype(float)
    st = o["status"]
    conv = (st == 0) | (st == 5) | (st == 6)
    sim.T1, sim.TS = T1, TS
    ms = sim.dispatch(np.minimum(it, cap), 2) / 1e3 if len(it) > 1024 else it.max() * T1 / 1e3
    w = oracle_lib.last_work()
    print(f"{tag:34s} conv {conv.mean():.4f} ({int((~conv).sum()):3d} not; st5 {int((st == 5).sum()):3d} st6 {int((st == 6).sum()):2d} st4 {int((st == 4).sum()):2d} "
          f"st2 {int((st == 2).sum())}) iters mean {it.mean():6.2f} p99 {np.percentile(it, 99):5.1f} p99.9 {np.percentile(it, 99.9):6.1f} "
          f"max {it.max():4.0f} at-cap {int((it >= cap).sum()):2d} >=60 {int((it >= 60).sum()):3d} >=40 {int((it >= 40).sum()):3d} "
          f"sweeps/it {w['sweeps'] / max(w['iterations'], 1):.3f} rolls/it {w['rollouts'] / max(w['iterations'], 1):.3f} model {ms:.2f} ms",
          flush=True)
    return it, conv


def main():
    quick = "--quick" in sys.argv
    t0 = time.time()
    seeds = range(3) if quick else range(8)
    tot = []
    for sd in seeds:
        inp = synth.solver_inputs(4096, 8, seed=sd)
        it, conv = row(f"config3 seed {sd} cap100 tol1e-8", solve(inp, True, 100, 1e-8), 100)
        tot.append((it, conv))
    allit = np.concatenate([t[0] for t in tot])
    allc = np.concatenate([t[1] for t in tot])
    print(f"   all seeds: conv {allc.mean():.5f} mean {allit.mean():.3f} p99 {np.percentile(allit, 99):.1f} p99.9 {np.percentile(allit, 99.9):.1f} "
          f"at-cap {int((allit >= 100).sum())}")
    for sd in range(3):
        inp = synth.solver_inputs(4096, 8, seed=sd)
        row(f"config3 seed {sd} max1000 tol1e-6", solve(inp, True, 1000, 1e-6), 1000)
    inp2 = synth.solver_inputs(1024, 4, seed=0)
    row("config2 (bench draw) cap100", solve(inp2, False, 100, 1e-8), 100)
    if not quick:
        for sd in range(3):
            inp = synth.solver_inputs(4096, 4, seed=sd)
            row(f"live objective B=4096 seed {sd}", solve(inp, False, 100, 1e-8), 100)
        for sd in range(1, 4):
            inp2 = synth.solver_inputs(1024, 4, seed=sd)
            row(f"config2 shape seed {sd}", solve(inp2, False, 100, 1e-8), 100)
    print(f"# {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
