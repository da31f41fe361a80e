#!/usr/bin/env python3
"""The instances that still run to the iteration cap (config 3, seeds 0 - 7, cap 100, tol 1e-8) handed to the independent
restatement of IPOPT's algorithm (oracle/ipopt_restated.py) at the REFERENCE's settings (tol 1e-6, max_iter 1000) from the
reference's cold start: does the reference's solver family do better on them?
    python tools/cap_runners_vs_ipopt.py > profiles/rNN_cap_runners_vs_ipopt.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p_)
import numpy as np  # noqa: E402
import ipopt_restated as ipr  # noqa: E402
import nlp_batch as nb  # noqa: E402
import oracle_lib  # noqa: E402
from mpc_rl_for_avs_amd import synth  # noqa: E402
from mpc_rl_for_avs_amd.reference_path import reference_states  # noqa: E402

REF = reference_states(0.1)
print("# seed instance | engine algorithm (oracle, cap 100, tol 1e-8): status iterations | the same at max_iter 1000, tol 1e-6 | "
      "IPOPT restatement (tol 1e-6, max_iter 1000): status iterations  [" + ", ".join(f"{k} = {v}" for k, v in sorted(ipr.STATUS.items())) + "]")
for seed in range(8):
    inp = synth.solver_inputs(4096, 8, seed=seed)
    kw = dict(vref=inp["vref"], others=inp["others"], collision_cost=True, xy_bounds=False, nthreads=8)
    o = oracle_lib.solve_batch(REF, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], max_iter=100, tol=1e-8, **kw)
    for i in np.nonzero(o["iters"] >= 100)[0]:
        sub = {k: (v[i:i + 1] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
        o6 = oracle_lib.solve_batch(REF, sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=sub["vref"],
                                    others=sub["others"], collision_cost=True, xy_bounds=False, max_iter=1000, tol=1e-6)
        p = nb.Batch.build(REF, sub["state"], sub["ego_index"], sub["weights"], sub["is_collide"], vref=sub["vref"], others=sub["others"],
                           collision_cost=True)
        t0 = time.time()
        r = ipr.solve(p, tol=1e-6, max_iter=1000, sf_min=1e-2)
        print(f"{seed} {int(i):5d} | {int(o['status'][i])} {int(o['iters'][i]):4d} | {int(o6['status'][0])} {int(o6['iters'][0]):4d} | "
              f"{int(np.ravel(r['status'])[0])} {int(np.ravel(r['iters'])[0]):4d}   ({time.time() - t0:.0f} s)", flush=True)
