"""Development aid: the GPU engine against the HOST EMULATION of the same kernel source (tests/cpu_wave_harness.cpp, compiled
with -ffp-contract=off) and against the C oracle, instance by instance: how many solutions are bit-identical, where the
others part ways, and which instances end in another minimiser.  Run with MPC_EXPERIMENT_LIB pointing at a candidate build
(e.g. one compiled with -ffp-contract=off) to see what the compiler's multiply-add contraction is responsible for."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import conftest, oracle_lib
from mpc_rl_for_avs_amd import engine, synth
from mpc_rl_for_avs_amd.reference_path import reference_states

ref = reference_states(0.1)
wave = conftest._host_solver("libcpu_wave.so", "cpu_wave_harness.cpp", "wave_solve_batch")
conv = lambda s: (s == 0) | ((s >= 5) & (s <= 7))
cases = [(1024, 8, True), (1024, 4, False), (4096, 8, True)]
seeds = [int(s) for s in sys.argv[1:]] or [0, 1, 2]
for seed in seeds:
    for (B, V, cc) in cases:
        inp = synth.solver_inputs(B, V, seed=seed)
        e = engine.MPCEngine(horizon=20, max_iter=100)
        t0 = time.time()
        g = e.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                          others=inp["others"], collision_cost=cc)
        tg = time.time() - t0
        e.close()
        sub = dict(inp)
        if not cc:
            sub["others"] = None
        split = B > 2048                       # the builds for deeper batches keep the linearised step in its own loop
        h = wave(ref, sub, collision_cost=cc, max_iter=100, split_linear=split)
        o = oracle_lib.solve_batch(ref, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                                   others=inp["others"], collision_cost=cc, max_iter=100, xy_bounds=False)
        same = np.array([np.array_equal(g["U"][b], h["U"][b]) and np.array_equal(g["X"][b], h["X"][b]) for b in range(B)])
        dU = np.abs(g["U"] - h["U"]).reshape(B, -1).max(axis=1)
        both = conv(g["status"]) & conv(o["status"])
        err = conftest.rel_u0_err(g["u0"], o["u0"])
        far = np.nonzero(both & (err > 1e-4))[0]
        bh = conv(g["status"]) & conv(h["status"])
        errh = conftest.rel_u0_err(g["u0"], h["u0"])
        farh = np.nonzero(bh & (errh > 1e-4))[0]
        print(f"seed {seed} B {B} V {V} cc {int(cc)}: gpu == host emulation bit for bit on {same.sum()} of {B}; iterations equal "
              f"{(g['iters'] == h['iters']).sum()}, statuses equal {(g['status'] == h['status']).sum()}; max |dU| median "
              f"{np.median(dU):.1e} p99 {np.percentile(dU, 99):.1e}; beyond 1e-4 of the host emulation: {farh.tolist()}, of the oracle: "
              f"{far.tolist()} (iters gpu {g['iters'][far].tolist()} oracle {o['iters'][far].tolist()}); call {tg*1e3:.1f} ms", flush=True)
