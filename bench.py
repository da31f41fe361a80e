#!/usr/bin/env python3
"""bench.py - MPC solves/sec of the MI355X engine on BASELINE.json's headline configuration.

A "step" is one pass of the hot path (`mpc_solve_batch`) over one batch of synthetic instances that is already
resident in HBM: configs[2] of BASELINE.json = batch 4096, horizon 20, 8 other vehicles, collision cost on.
With N GPUs every rank owns its own 4096 instances (weak scaling) and the step ends with the RCCL all-gather of
the actions, the path's only exchange.  Prints ONE JSON line on rank 0.

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (torch.distributed.run as a child
process, before this process has touched the GPU) and exits with the child's code; under torchrun WORLD_SIZE must
equal --gpus.  `n_gpus` in the output is always the number of ranks that ran.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096
HORIZON = 20
V = 8
# Iteration cap of the timed configuration = the engine's default (mpc_default_config: 100).  The reference allows IPOPT 1000
# iterations (agents/pure_mpc.py:294) and acts on the last iterate of a solve that fails (:303-305); an instance still running
# here at the cap returns its iterate with status 1 the same way.  A batch takes as long as its slowest instance, so the cap is
# the knob of this number: `solver_settings_sweep` in the same line reports 40 / 60, the reference's own settings (max_iter
# 1000 with tol 1e-8 and with its tol 1e-6) and max_iter 1000 with the progress guard (mpc_config.stall_window = 64).
# `value` counts CONVERGED instances only (a solve is an NLP solved to tolerance).
MAX_ITER = 100
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: FP64 vector
ALG_BYTES_PER_SOLVE = 756 + 32 * V   # inputs + u0/status/iters, FP64


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side measurements: profiling runs, so that every launch of the solve kernel in the trace "
                         "is one of the warm-up / timed steps")
    ap.add_argument("--streams", type=int, default=1,
                    help="batches in flight in the timed loop (default 1: one batch at a time, the headline definition)")
    return ap.parse_args()


def spawn_ranks(a) -> int:
    """--gpus N > 1 outside torchrun: one process per GPU via torch.distributed.run; this parent never touches a GPU."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def conv_mask(status):
    return (status == 0) | ((status >= 5) & (status <= 7))      # MPC_STATUS_IS_SOLVED


def cpu_baseline(inp):
    import numpy as np
    """The CPU oracle (a from-scratch port of the same NLP + algorithm, oracle/mpc_oracle.c) on the host cores: the
    whole batch with OpenMP over instances at the thread count with the best wall time, and a bounded sample on ONE
    thread.  Also returns the oracle's solutions (the parity check) and its work counters."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    # read by the OpenMP runtime the oracle's library brings in, when it is loaded: one thread per core, neighbours first
    os.environ.setdefault("OMP_PLACES", "cores")
    os.environ.setdefault("OMP_PROC_BIND", "close")
    import oracle_lib
    from mpc_rl_for_avs_amd.reference_path import reference_states
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))          # what this process may actually use
    except AttributeError:
        pass
    try:                                                         # ... and the container's CPU share (cgroup v2 quota)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            ncpu = max(1, min(ncpu, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    ref = reference_states()

    def run(n, threads):
        sl = slice(0, n)
        t0 = time.perf_counter()
        out = oracle_lib.solve_batch(ref, inp["state"][sl], inp["ego_index"][sl], inp["weights"][sl],
                                     inp["is_collide"][sl], vref=inp["vref"][sl], others=inp["others"][sl],
                                     collision_cost=True, max_iter=MAX_ITER, xy_bounds=False, nthreads=threads)
        return time.perf_counter() - t0, out
    # every thread count tried with the MEDIAN of its runs (the host is shared: a best-of figure does not reproduce);
    # `value` is the best median
    tried, best, out = [], None, None
    run(BATCH, ncpu)                                              # first touch: thread pool, page faults
    for cores in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
        ts = []
        for _ in range(5):
            dt, out = run(BATCH, cores)
            ts.append(dt)
        med = float(np.median(ts))
        tried.append({"threads": cores, "median_s": med, "min_s": float(min(ts)), "max_s": float(max(ts)),
                      "solves_per_s": BATCH / med})
        if best is None or med < best[0]:
            best = (med, cores)
    dt, cores = best
    work = oracle_lib.last_work()
    n1 = 1024
    t1s = [run(n1, 1)[0] for _ in range(3)]
    dt1 = float(np.median(t1s))
    # independent solvers on one thread: scipy SLSQP on the loop restatement of the NLP (analytic
    # reduced gradient, dense BFGS) and the dense restatement of IPOPT's algorithm - the closest thing to "what the
    # reference's solver does per instance" that can run here; CasADi's graph build + nlpsol construction, which the
    # reference pays on top at every step, is not included (tools/time_reference_casadi.py times it where CasADi exists)
    import warnings
    import ipopt_restated as ipr
    import nlp_batch as nb
    import nlp_spec as S
    import scipy_crosscheck as X
    n_ind = 16
    t0 = time.perf_counter()
    ok_s = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for b in range(n_ind):
            p = S.Problem.build(HORIZON, 0.1, inp["state"][b], inp["ego_index"][b], ref.copy(), inp["weights"][b],
                                inp["is_collide"][b], collision_cost=True, others=inp["others"][b])
            p.ref[:, 2] = inp["vref"][b]
            ok_s += bool(X.solve_slsqp(p)["success"])
    dt_s = time.perf_counter() - t0
    pb = nb.Batch.build(ref, inp["state"][:n_ind], inp["ego_index"][:n_ind], inp["weights"][:n_ind],
                        inp["is_collide"][:n_ind], vref=inp["vref"][:n_ind], others=inp["others"][:n_ind], collision_cost=True)
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):          # one BLAS thread: dense 208 x 208 factorisations do not want more
        t0 = time.perf_counter()
        ok_i = sum(int(ipr.solve(pb.take([b]), tol=1e-6, max_iter=1000, sf_min=1e-2)["status"] == 0) for b in range(n_ind))
        dt_i = time.perf_counter() - t0
    independent = {"scipy_slsqp": {"value": n_ind / dt_s, "unit": "solves/s", "cores": 1, "succeeded": ok_s,
                                   "sample": f"first {n_ind} instances, oracle/scipy_crosscheck.py, {dt_s:.1f} s"},
                   "ipopt_restated": {"value": n_ind / dt_i, "unit": "solves/s", "cores": 1, "succeeded": ok_i,
                                      "sample": f"first {n_ind} instances, oracle/ipopt_restated.py (dense numpy, tol 1e-6 "
                                                f"like the reference), {dt_i:.1f} s"}}
    return dict(value=BATCH / dt, unit="solves/s", cores=cores, kind="port", independent_solvers=independent,
                sample=f"all {BATCH} instances of the same batch, oracle/mpc_oracle.c, OpenMP over instances pinned to cores "
                       f"(OMP_PLACES=cores, proc_bind close), {cores} of {ncpu} usable hardware threads: the thread count "
                       f"with the best MEDIAN of 5 runs ({dt:.3f} s); mean {float(out['iters'].mean()):.1f} iterations",
                thread_counts_tried=tried,
                scaling_vs_one_thread=(BATCH / dt) / (n1 / dt1) / cores,
                one_thread={"value": n1 / dt1, "unit": "solves/s", "cores": 1,
                            "sample": f"first {n1} instances, median of 3 runs: {dt1:.2f} s"}), out, work


def pmc_summary():
    """The committed rocprofv3 PMC passes of this command (separate --pmc runs, tools/pmc_run.sh -> tools/pmc_summary.py
    -> profiles/rNN_pmc_summary.csv): {counter: mean per launch}, and the file they come from."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")))
    if not files:
        return {}, None
    vals = {r["counter"]: float(r["mean_per_launch"]) for r in csv.DictReader(open(files[-1]))}
    return vals, os.path.relpath(files[-1], ROOT)


def main():
    a = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and a.gpus > 1:
        sys.exit(spawn_ranks(a))
    world = int(world_env or "1")
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {a.gpus} "
                         f"or plainly as `python bench.py --gpus {a.gpus}`")
    # read when the HIP runtime starts: with the default of 4 hardware queues, streams carrying independent batches can
    # end up sharing a queue and serialise (the in_flight side measurement); multi-rank runs keep the runtime's default
    if world == 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import numpy as np
    import torch
    import torch.distributed as dist
    from mpc_rl_for_avs_amd import engine, synth, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the latter rehearses the RCCL path on 1 GPU
    saved_stdout = None
    if use_dist:
        # RCCL writes its version banner and kernel-command-line warnings to STDOUT when the communicator is created (at the
        # first collective): rank 0's stdout must carry ONE JSON line, so file descriptor 1 points at stderr until the timed
        # region is over
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    inp = synth.solver_inputs(BATCH, V, seed=rank, N=HORIZON)
    t = lambda x, dt_: torch.as_tensor(np.ascontiguousarray(x), dtype=dt_, device=dev)
    args = dict(state=t(inp["state"], torch.float64), ego_index=t(inp["ego_index"], torch.int32),
                weights=t(inp["weights"], torch.float64), is_collide=t(inp["is_collide"], torch.uint8),
                vref=t(inp["vref"], torch.float64), others=t(inp["others"], torch.float64), collision_cost=True)
    eng = engine.MPCEngine(horizon=HORIZON, max_iter=MAX_ITER, device=local_rank)
    out = dict(u0=torch.empty((BATCH, 2), dtype=torch.float64, device=dev),
               status=torch.empty(BATCH, dtype=torch.int32, device=dev),
               iters=torch.empty(BATCH, dtype=torch.int32, device=dev))
    gathered = gathered_st = None

    n_str = max(1, a.streams)
    side = [torch.cuda.Stream(dev) for _ in range(n_str)] if n_str > 1 else [torch.cuda.current_stream(dev)]
    outs_s = [out] + [dict((k, torch.empty_like(v)) for k, v in out.items()) for _ in range(n_str - 1)]

    def step(i, ev=None):
        nonlocal gathered, gathered_st
        with torch.cuda.stream(side[i % n_str]):
            if ev is not None:
                ev[0].record()
            # (several batches in flight on one handle: MPC_FLAG_THROUGHPUT, include/mpc_mi355x.h)
            eng.solve_batch_torch(**args, out=outs_s[i % n_str], throughput=n_str > 1)      # enqueued on torch's current stream = side[i]
            if ev is not None:
                ev[1].record()
            if use_dist:
                gathered, gathered_st = sharding.all_gather_results(outs_s[i % n_str]["u0"], outs_s[i % n_str]["status"])

    for i in range(max(a.warmup, n_str if n_str > 1 else 0)):
        step(i)
    # kernel-only HIP events, recorded on the stream the kernel is launched on (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, ev[i])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dist_info = None
    if use_dist:
        if tuple(gathered.shape) != (world * BATCH, 2) or tuple(gathered_st.shape) != (world * BATCH,):
            raise SystemExit(f"bench.py: rank {rank}: gathered actions {tuple(gathered.shape)} / status {tuple(gathered_st.shape)}, "
                             f"expected ({world * BATCH}, 2) / ({world * BATCH},) for {world} ranks")
        lo = rank * BATCH
        last = outs_s[(a.steps - 1) % n_str]
        # rank -> physical device: every rank must drive its own GPU (a 1-rank rehearsal has nothing to compare)
        ids = sharding.rank_devices(dev)
        try:
            n_distinct = sharding.assert_distinct_devices(ids)
        except RuntimeError as e:
            raise SystemExit(f"bench.py: {e}")
        dist_info = {"backend": dist.get_backend(), "world_size": world, "gathered_actions_shape": list(gathered.shape),
                     "gathered_status_shape": list(gathered_st.shape),
                     "own_block_equals_local": bool(torch.equal(gathered[lo:lo + BATCH], last["u0"]) and
                                                    torch.equal(gathered_st[lo:lo + BATCH], last["status"])),
                     "bytes_per_rank_per_step": BATCH * 24, "rank_devices": ids, "distinct_devices": n_distinct}
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern = np.array([s.elapsed_time(e) for s, e in ev])
    # with several batches in flight the events of one stream bracket a kernel that shares the GPU: not a kernel time
    kern_ms = float(kern.mean()) if n_str == 1 else None
    status = out["status"].cpu().numpy()
    iters = out["iters"].cpu().numpy()
    n_conv_all = int(conv_mask(status).sum())
    if use_dist:
        # every rank solves its own batch (seed = rank): the converged count is summed over the ranks, and each rank's
        # own kernel time / slowest instance is reported, because the step lasts as long as the rank whose batch holds
        # the longest chain of iterations (the time is bound by that chain, not by the gather)
        mine = torch.tensor([float(kern.mean()), float(iters.max()), float(n_conv_all)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu().numpy()
        n_conv_all = int(per_rank[:, 2].sum())
        dist_info.update({"per_rank_kernel_ms": [round(float(x), 4) for x in per_rank[:, 0]],
                          "per_rank_iters_max": [int(x) for x in per_rank[:, 1]],
                          "per_rank_converged": [int(x) for x in per_rank[:, 2]]})

    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        conv = conv_mask(status)
        value_all = world * BATCH * a.steps / elapsed
        value = value_all * n_conv_all / float(world * BATCH)     # solves to tolerance per second, counted on every rank
        pmc, pmc_file = pmc_summary()
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "mpc_solve_wave_kernel<CC=true, N=20, OCC=2, RELAX=7> (the latency build, used up to four waves "
                          "per SIMD of batch depth, mpc_engine.hip: dispatch_solve; kernel_ms brackets the call, i.e. it includes "
                          "the 16 us mpc_order_kernel in front of it)", "kernel_ms": kern_ms,
                "kernel_ms_median": float(np.median(kern)) if n_str == 1 else None,
                "algorithmic_bytes_per_solve": ALG_BYTES_PER_SOLVE,
                "note": "path is bound by the serial FP64 + LDS-latency chain of its slowest instance; the working set is "
                        "LDS-resident and HBM carries only inputs/outputs (DESIGN.md section 4)"}
        if kern_ms is not None:
            roof["achieved"] = BATCH * ALG_BYTES_PER_SOLVE / (kern_ms * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # KB units; the loads are 8-byte strided, outside the access widths MI355X_MICROARCH.md calibrates: raw counters
            roof["traffic"] = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            roof["traffic_source"] = f"{pmc_file}: separate rocprofv3 --pmc passes of this command, not measured in this run"
        res = {
            "metric": "MPC solves/sec (horizon=20, batch=4096)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: batch=4096 pure_mpc horizon=20, 8 other vehicles, collision cost "
                                   f"on, cold start, tol 1e-8, max_iter {MAX_ITER}; per-GPU batch fixed",
                       "batch_per_gpu": BATCH, "horizon": HORIZON, "n_vehicles": V, "seed": "rank", "max_iter": MAX_ITER,
                       "parallelism": f"instance-sharded x{world}, all-gather of actions" +
                                      (f", {n_str} batches in flight on {n_str} streams" if n_str > 1 else "")},
            "roofline": roof,
            "distributed": dist_info,
            "value_smooth_only": value_all * float((status == 0).mean()) if world == 1 else None,
            "solver": {"converged_frac": float(conv.mean()), "smooth_kkt_frac": float((status == 0).mean()),
                       "on_kink_frac": float(((status == 5) | (status == 7)).mean()),
                       "acceptable_level_frac": float((status >= 6).mean()), "iters_mean": float(iters.mean()),
                       "iters_p99": float(np.percentile(iters, 99)), "iters_max": int(iters.max()),
                       "value_all_instances": value_all,
                       "note": "value = solved instances per second (MPC_STATUS_IS_SOLVED: status 0 = KKT point of the smooth NLP to "
                               "tol; 5 = KKT point with a vehicle held at the d = 1 discontinuity of the collision cost, a notion "
                               "IPOPT does not have; 6 / 7 = IPOPT's acceptable level).  value_smooth_only (top level, rank 0's "
                               "batch) counts status 0 alone - what MPC_FLAG_STRICT_DISCONTINUITY would report as solved besides "
                               "status 6; value_all_instances also counts the ones that return their last iterate at the cap "
                               "(status 1) or stalled (4), as the reference does"},
        }
        # the resource this kernel actually consumes: vector-instruction issue slots (a wave64 FP64 instruction occupies
        # its SIMD for 4 cycles).  SIMD-cycles needed = wave-instructions (PMC) x 4; available = SIMDs x clock x kernel time.
        if "SQ_INSTS_VALU" in pmc and kern_ms is not None:
            prop = torch.cuda.get_device_properties(dev)
            simds = 4 * prop.multi_processor_count
            clock_hz = 1e3 * float(getattr(prop, "clock_rate", 2.4e6))
            res["valu_issue"] = {"wave_instructions_per_launch": pmc["SQ_INSTS_VALU"], "simds": simds,
                                 "clock_ghz": clock_hz / 1e9,
                                 "frac": pmc["SQ_INSTS_VALU"] * 4.0 / (simds * clock_hz * kern_ms * 1e-3),
                                 "source": f"{pmc_file} (instruction count of a separate --pmc pass) over this run's kernel time",
                                 "note": "fraction of the GPU's vector-instruction issue slots used during the kernel"}
        if world == 1 and not a.no_cpu_baseline:
            cb, oref, work = cpu_baseline(inp)
            res["cpu_baseline"] = cb
            # algorithmic FP64 rate: operations the algorithm needs (counted by the oracle, which runs the same iteration)
            # over the GPU's kernel time
            if kern_ms is not None:
                res["fp64"] = {"flops_per_solve": work["flops"] / BATCH, "transcendentals_per_solve": work["transcendentals"] / BATCH,
                               "flops_per_iteration": work["flops"] / max(work["iterations"], 1.0),
                               "achieved_tflops": work["flops"] / (kern_ms * 1e-3) / 1e12, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                               "frac": work["flops"] / (kern_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                               "note": "operation counts of oracle/mpc_oracle.c (oracle_last_work) for this batch over the kernel "
                                       "time: algorithmic flops, not issued instructions"}
            # parity: against the oracle instance by instance, and - independently of any solver - KKT certificates of
            # the reference NLP for every solution the engine calls converged (oracle/kkt_batch.py)
            import kkt_batch as kb
            import nlp_batch as nb
            from mpc_rl_for_avs_amd.reference_path import reference_states
            full = dict(u0=torch.empty((BATCH, 2), dtype=torch.float64, device=dev),
                        U=torch.empty((BATCH, HORIZON, 2), dtype=torch.float64, device=dev),
                        X=torch.empty((BATCH, HORIZON + 1, 4), dtype=torch.float64, device=dev),
                        status=torch.empty(BATCH, dtype=torch.int32, device=dev),
                        iters=torch.empty(BATCH, dtype=torch.int32, device=dev))
            eng.solve_batch_torch(**args, out=full, sync=True)
            u0 = full["u0"].cpu().numpy()
            both = conv & conv_mask(oref["status"])
            err = np.abs(u0 - oref["u0"]).max(axis=1) / np.maximum(1.0, np.abs(oref["u0"]).max(axis=1))
            sel = np.nonzero(conv)[0]
            p = nb.Batch.build(reference_states(), inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"],
                               vref=inp["vref"], others=inp["others"], collision_cost=True)
            psel = p.take(sel)
            sf_obj = kb.objective_scale(psel)        # IPOPT's objective scaling, from the NLP data alone
            Xs, Us = full["X"].cpu().numpy()[sel], full["U"].cpu().numpy()[sel]
            # the tolerance an instance was solved to: tol 1e-8, or IPOPT's acceptable level 1e-6 for status 6 / 7
            tol_i = np.where(status[sel] >= 6, 1e-6, 1e-8)
            cert = kb.certify(psel, Xs, Us, eps_c=tol_i / sf_obj, sf=sf_obj)
            plain = kb.certify(psel, Xs, Us, eps_c=tol_i)
            res["parity"] = {"both_converged": int(both.sum()), "u0_rel_linf_max": float(err[both].max()),
                             "u0_rel_linf_p99": float(np.percentile(err[both], 99)),
                             "frac_within_1e-4": float((err[both] <= 1e-4).mean()),
                             "status_equal_frac": float((status == oref["status"]).mean()),
                             "n_certified": int(((cert["stationarity"] <= tol_i) & (cert["stationarity_ipopt"] <= tol_i)).sum()),
                             "n_converged": int(sel.size),
                             "n_acceptable_level": int((status[sel] >= 6).sum()),
                             "kkt_stationarity_max": float(cert["stationarity"].max()),
                             "kkt_stationarity_ipopt_units_max_over_tol": float((cert["stationarity_ipopt"] / tol_i).max()),
                             "kkt_feasibility_max": float(cert["feasibility"].max()),
                             "kkt_bound_violation_max": float(cert["bound_violation"].max()),
                             "n_certified_unscaled_complementarity_1e-8": int((plain["stationarity"] <= tol_i).sum()),
                             "note": "certificates: relative stationarity with re-fitted non-negative multipliers complementary to 1e-8 in "
                                     "the units of IPOPT's criterion (objective scaled by sf, computed from the NLP data; 1e-8 / sf "
                                     "unscaled) - and to 1e-8 unscaled for the count beside it; oracle/kkt_batch.py; 'ref' = CPU "
                                     "oracle + certificates because CasADi/IPOPT cannot run here.  n_certified counts the instances whose "
                                     "stationarity is <= their tolerance (1e-8; 1e-6 at IPOPT's acceptable level) BOTH relative to "
                                     "max(1, |grad f|_inf) and in IPOPT's own units (scaled residual / s_d, Waechter & Biegler eq. (5), "
                                     "(6)); the gate is n_certified == n_converged, no allowance"}
        if world == 1 and not a.no_side:
            res.update(side_measurements(a, eng, args, inp, out, dev))
        print(json.dumps(res), flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()


def side_measurements(a, eng, args, inp, out, dev):
    """Not the metric: other iteration caps, batches in flight, the PCIe-inclusive call, the observation-level call,
    the iterative-linear QP.  Medians of event-timed repetitions."""
    import numpy as np
    import torch
    from mpc_rl_for_avs_amd import engine, synth
    res = {}

    def timed(fn, reps=7):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    # other solver settings: lower caps, and the reference's own (ipopt max_iter 1000, tol 1e-6: agents/pure_mpc.py:294-295)
    caps = {}
    for key, mi, tol, sw in (("max_iter 40", 40, 1e-8, 0), ("max_iter 60", 60, 1e-8, 0), ("max_iter 1000", 1000, 1e-8, 0),
                             ("max_iter 1000, tol 1e-6 (reference)", 1000, 1e-6, 0),
                             ("max_iter 1000, stall_window 64", 1000, 1e-8, 64),
                             ("max_iter 1000, tol 1e-6, stall_window 64", 1000, 1e-6, 64)):
        e2 = engine.MPCEngine(horizon=HORIZON, max_iter=mi, tol=tol, stall_window=sw, device=dev.index)
        o2 = e2.solve_batch_torch(**args, sync=True)
        ms = timed(lambda: e2.solve_batch_torch(**args, out=o2), reps=5 if mi <= 100 or sw else 3)
        st2, it2 = o2["status"].cpu().numpy(), o2["iters"].cpu().numpy()
        cf = float(conv_mask(st2).mean())
        caps[key] = {"ms": ms, "value": BATCH * cf / (ms * 1e-3), "value_all_instances": BATCH / (ms * 1e-3), "converged_frac": cf,
                     "at_cap": int((st2 == 1).sum()), "stalled": int((st2 == 4).sum()), "iters_mean": float(it2.mean()),
                     "iters_p99": float(np.percentile(it2, 99)), "iters_max": int(it2.max())}
        e2.close()
    res["solver_settings_sweep"] = caps
    # the reference's own solver settings (ipopt max_iter 1000, tol 1e-6: agents/pure_mpc.py:294-295), no stall_window, over
    # seeds 0-2: a first-class sibling of the headline
    e_ref = engine.MPCEngine(horizon=HORIZON, max_iter=1000, tol=1e-6, device=dev.index)
    rows = []
    for sd in (0, 1, 2):
        inp_s = synth.solver_inputs(BATCH, V, seed=sd, N=HORIZON)
        t_ = lambda x, dt_: torch.as_tensor(np.ascontiguousarray(x), dtype=dt_, device=dev)
        a_s = dict(state=t_(inp_s["state"], torch.float64), ego_index=t_(inp_s["ego_index"], torch.int32),
                   weights=t_(inp_s["weights"], torch.float64), is_collide=t_(inp_s["is_collide"], torch.uint8),
                   vref=t_(inp_s["vref"], torch.float64), others=t_(inp_s["others"], torch.float64), collision_cost=True)
        o_s = e_ref.solve_batch_torch(**a_s, sync=True)
        ms = timed(lambda: e_ref.solve_batch_torch(**a_s, out=o_s), reps=5)
        st_s, it_s = o_s["status"].cpu().numpy(), o_s["iters"].cpu().numpy()
        cf = float(conv_mask(st_s).mean())
        rows.append({"seed": sd, "ms": ms, "value": BATCH * cf / (ms * 1e-3), "converged_frac": cf, "smooth_kkt_frac": float((st_s == 0).mean()),
                     "iters_mean": float(it_s.mean()), "iters_max": int(it_s.max()), "stalled": int((st_s == 4).sum()),
                     "at_cap": int((st_s == 1).sum())})
    e_ref.close()
    mss = sorted(r["ms"] for r in rows)
    res["reference_settings"] = {"workload": "config 3 (B = 4096, 8 vehicles, collision cost) at the reference's IPOPT options: max_iter 1000, "
                                             "tol 1e-6, no stall_window", "per_seed": rows, "ms_min": mss[0], "ms_median": mss[1], "ms_max": mss[2],
                                 "value_median": sorted(r["value"] for r in rows)[1], "unit": "solves/s"}
    # MPC_FLAG_STRICT_DISCONTINUITY on the headline batch: the same iterates, an instance that ends on the d = 1 discontinuity
    # reported as not solved (status 8) with its last iterate - the reference-strict count
    o_st = eng.solve_batch_torch(**args, sync=True, strict_discontinuity=True)
    st_st = o_st["status"].cpu().numpy()
    res["strict_discontinuity"] = {"solved_frac": float(conv_mask(st_st).mean()), "kink_unsolved": int((st_st == 8).sum()),
                                   "note": "MPC_FLAG_STRICT_DISCONTINUITY: status 5 / 7 are reported as 8 (not solved, last iterate)"}
    # the headline over seeds 0-2 (the batch time is that of its slowest instance: it moves with the draw)
    per_seed = []
    for sd in (0, 1, 2):
        inp_s = synth.solver_inputs(BATCH, V, seed=sd, N=HORIZON)
        t_ = lambda x, dt_: torch.as_tensor(np.ascontiguousarray(x), dtype=dt_, device=dev)
        a_s = dict(state=t_(inp_s["state"], torch.float64), ego_index=t_(inp_s["ego_index"], torch.int32),
                   weights=t_(inp_s["weights"], torch.float64), is_collide=t_(inp_s["is_collide"], torch.uint8),
                   vref=t_(inp_s["vref"], torch.float64), others=t_(inp_s["others"], torch.float64), collision_cost=True)
        o_s = eng.solve_batch_torch(**a_s, sync=True)
        ms = timed(lambda: eng.solve_batch_torch(**a_s, out=o_s), reps=9)
        st_s, it_s = o_s["status"].cpu().numpy(), o_s["iters"].cpu().numpy()
        per_seed.append({"seed": sd, "ms": ms, "converged_frac": float(conv_mask(st_s).mean()), "iters_max": int(it_s.max()),
                         "value": BATCH * float(conv_mask(st_s).mean()) / (ms * 1e-3)})
    mss = sorted(r["ms"] for r in per_seed)
    res["headline_over_seeds"] = {"per_seed": per_seed, "ms_min": mss[0], "ms_median": mss[1], "ms_max": mss[2],
                                  "value_min": min(r["value"] for r in per_seed), "value_median": sorted(r["value"] for r in per_seed)[1],
                                  "value_max": max(r["value"] for r in per_seed), "unit": "solves/s",
                                  "note": "config 3 at cap 100, median of 9 event-timed launches per seed; `value` above is seed 0"}
    # BASELINE configs[1]: batch=1024, horizon 20, 4 other vehicles, the LIVE objective (agents/pure_mpc.py:204-212: others
    # matter only through the detector), one GPU
    inp2 = synth.solver_inputs(1024, 4, seed=0, N=HORIZON)
    t_ = lambda x, dt_: torch.as_tensor(np.ascontiguousarray(x), dtype=dt_, device=dev)
    a2 = dict(state=t_(inp2["state"], torch.float64), ego_index=t_(inp2["ego_index"], torch.int32),
              weights=t_(inp2["weights"], torch.float64), is_collide=t_(inp2["is_collide"], torch.uint8),
              vref=t_(inp2["vref"], torch.float64), others=None, collision_cost=False)
    o2 = eng.solve_batch_torch(**a2, sync=True)
    ms = timed(lambda: eng.solve_batch_torch(**a2, out=o2), reps=15)
    st2, it2 = o2["status"].cpu().numpy(), o2["iters"].cpu().numpy()
    res["config2"] = {"workload": "BASELINE configs[1]: batch=1024 pure_mpc horizon=20, 4 other vehicles, live objective, cold "
                                  f"start, tol 1e-8, max_iter {MAX_ITER}", "ms": ms, "value": 1024 * float(conv_mask(st2).mean()) / (ms * 1e-3),
                      "unit": "solves/s", "converged_frac": float(conv_mask(st2).mean()), "iters_mean": float(it2.mean()),
                      "iters_max": int(it2.max())}
    # BASELINE configs[3]: 256 parallel intersection environments, MPC-in-the-loop rollout (policy -> clip -> mpc_predict_batch
    # -> environment step -> buffer row; agents/ppo_mpc.py:385-469), v0 = the RL action is the reference speed
    from mpc_rl_for_avs_amd import rollout
    torch.manual_seed(1234)
    pol = rollout.ActorCritic(1).to(dev)
    e4 = engine.MPCEngine(horizon=HORIZON, max_iter=MAX_ITER, device=dev.index)
    env4 = rollout.SyntheticIntersectionEnv(256, device=dev, seed=0, n_others=4)
    col4 = rollout.BatchedCollector(env4, pol, e4, version="v0", algorithm="ppo", n_steps=64, collision_cost=False, seed=0)
    col4.collect_rollouts()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    st4 = col4.collect_rollouts()
    torch.cuda.synchronize()
    dt4 = time.perf_counter() - t1
    s4 = col4.last_mpc["status"].cpu().numpy()
    res["config4"] = {"workload": "BASELINE configs[3]: ppo_mpc 256 parallel intersection envs (synthetic stand-in for highway-env), "
                                  "MPC-in-the-loop rollout of 64 steps, 4 other vehicles, v0, untrained seeded policy, hipGraph step",
                      "value": 256 * 64 / dt4, "unit": "env-steps/s", "ms_per_step": dt4 / 64 * 1e3,
                      "converged_frac_last_step": float(conv_mask(s4).mean()), "episodes": st4["episodes"],
                      "mpc_unconverged_in_rollout": int(st4.get("mpc_unconverged", -1))}
    # the same rollout with the v1 input domain: the RL action = the three cost weights from [-1, 1]^3 (agents/ppo_mpc.py:407-420),
    # i.e. negative cost weights most of the time
    torch.manual_seed(1234)
    pol1 = rollout.ActorCritic(3).to(dev)
    env41 = rollout.SyntheticIntersectionEnv(256, device=dev, seed=0, n_others=4)
    col41 = rollout.BatchedCollector(env41, pol1, e4, version="v1", algorithm="ppo", n_steps=64, collision_cost=False, seed=0)
    col41.collect_rollouts()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    st41 = col41.collect_rollouts()
    torch.cuda.synchronize()
    dt41 = time.perf_counter() - t1
    res["config4_v1"] = {"workload": "the config-4 rollout with v1 actions (cost weights from [-1, 1]^3, agents/ppo_mpc.py:407-420)",
                         "value": 256 * 64 / dt41, "unit": "env-steps/s", "ms_per_step": dt41 / 64 * 1e3,
                         "converged_frac_rollout": 1.0 - float(st41.get("mpc_unconverged", 0)) / (256 * 64),
                         "mpc_unconverged_in_rollout": int(st41.get("mpc_unconverged", -1))}
    e4.close()
    # BASELINE configs[0] / the reference's own caller shape (main/run_pure_mpc.py:27): ONE environment, predict() per step,
    # closed loop on the synthetic environment; wall time of the call as Python sees it (H2D of the observation, preamble and
    # solve kernels, D2H of the action, read-back of the detector record)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import run_pure_mpc
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the agent prints the reference's "NOTICE: Not found solution" (agents/
        outcome, log, lat_ms = run_pure_mpc.run(steps=150, n_others=1, seed=0, verbose=False)   # pure_mpc.py:303-305): stdout carries ONE JSON line
    res["predict_b1"] = {"workload": "BASELINE configs[0]: single ego, horizon 20, 1 other vehicle, closed loop (tools/run_pure_mpc.py), "
                                     "PureMPC_Agent.predict() per step", "ms_median": float(np.median(lat_ms)),
                         "ms_p95": float(np.percentile(lat_ms, 95)), "steps": len(log), "outcome": outcome,
                         "converged_frac": float(np.mean([r[6] in (0, 5, 6, 7) for r in log])), "unit": "ms per predict() call",
                         "calls_per_s": 1e3 / float(np.median(lat_ms))}
    # batches in flight: the straggler tail of one batch overlaps with the bulk of the next ones (same kernel, same inputs,
    # identical outputs) - what a serving loop with several independent environment groups would run
    n_fl = 8       # = GPU_MAX_HW_QUEUES above: streams that share a hardware queue serialise (tools/gpu_inflight.py, round 5),
    # and which of torch's pool streams share one is the runtime's choice: each is probed (round 6, mpc_streams_overlap)
    try:
        streams, probed = engine.concurrent_streams(n_fl, dev), True
    except engine.EngineError:          # fewer than eight hardware queues in this process (GPU_MAX_HW_QUEUES set from outside)
        streams, probed = [torch.cuda.Stream(dev) for _ in range(n_fl)], False
    outs = []
    for sq in streams:
        with torch.cuda.stream(sq):
            outs.append(eng.solve_batch_torch(**args, throughput=True))
    torch.cuda.synchronize()
    k_fl = max(3 * a.steps, 12 * n_fl)
    t1 = time.perf_counter()
    for i in range(k_fl):
        with torch.cuda.stream(streams[i % n_fl]):
            eng.solve_batch_torch(**args, out=outs[i % n_fl], throughput=True)      # MPC_FLAG_THROUGHPUT
    torch.cuda.synchronize()
    el = time.perf_counter() - t1
    res["in_flight"] = {"streams": n_fl, "streams_probed_to_overlap": probed, "steps": k_fl, "value": BATCH * k_fl / el, "unit": "solves/s",
                        "ms_per_batch": el / k_fl * 1e3, "counts": "all instances (converged fraction as in `solver`)",
                        "identical_outputs_across_streams": bool(all(torch.equal(o["u0"], outs[0]["u0"]) for o in outs)),
                        "frac_u0_within_1e-6_of_the_timed_run": float(((outs[0]["u0"] - out["u0"]).abs().amax(dim=1) <= 1e-6).float().mean()),
                        "note": f"throughput with {n_fl} batches of 4096 in flight (MPC_FLAG_THROUGHPUT: the 3-waves-per-SIMD build, 12 "
                                "instances per CU by LDS); `value` above is one batch at a time"}
    # ONE launch of 65 536 instances of the same workload (the bulk regime: every SIMD holds three waves until the launch drains):
    # what the tail costs when nothing else is in flight to cover it (launch-order tiers, mpc_engine.hip mpc_order_kernel)
    inp_b = synth.solver_inputs(65536, V, seed=0, N=HORIZON)
    t_ = lambda x, dt_: torch.as_tensor(np.ascontiguousarray(x), dtype=dt_, device=dev)
    a_b = dict(state=t_(inp_b["state"], torch.float64), ego_index=t_(inp_b["ego_index"], torch.int32),
               weights=t_(inp_b["weights"], torch.float64), is_collide=t_(inp_b["is_collide"], torch.uint8),
               vref=t_(inp_b["vref"], torch.float64), others=t_(inp_b["others"], torch.float64), collision_cost=True)
    o_b = eng.solve_batch_torch(**a_b, sync=True)
    ms_b = timed(lambda: eng.solve_batch_torch(**a_b, out=o_b), reps=3)
    st_b = o_b["status"].cpu().numpy()
    res["bulk_launch"] = {"workload": "one launch of 65 536 instances of config 3 (16 x the headline batch), cap 100", "ms": ms_b,
                          "value": 65536 / (ms_b * 1e-3), "unit": "solves/s", "converged_frac": float(conv_mask(st_b).mean()),
                          "counts": "all instances"}
    del a_b, o_b
    # BASELINE configs[4]'s total (2048 environments, 8 GPUs x 256 there) on ONE GPU: two groups of 1024 environments stepped on two
    # streams that run side by side (rollout.PipelinedCollector, engine.concurrent_streams).  In a child process with the
    # runtime's default of 4 hardware queues: with more than 4 (this process runs with 8 for the in_flight measurement above) the
    # same two groups are 20 % slower - 1.80 against 2.22 M env-steps/s for 5, 6, 7, 8 or 16 queues against 2, 3 or 4
    # (profiles/r06_hw_queues_scan.txt) - although the streams pass the overlap probe
    import subprocess
    env5 = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    try:      # a side measurement in another process must not cost the run its line
        r5 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_rollout.py"), "--envs", "2048", "--groups", "2", "--steps", "64"],
                            env=env5, capture_output=True, text=True, timeout=600)
        if r5.returncode != 0:
            raise RuntimeError(f"exit code {r5.returncode}: " + r5.stderr[-1500:])
        d5 = json.loads(r5.stdout.strip().splitlines()[-1])
        res["config5_one_gpu"] = {"workload": "2048 parallel intersection envs (BASELINE configs[4]'s total) on one GPU, two groups of 1024 on "
                                              "two streams, MPC-in-the-loop rollout of 64 steps, v0, hipGraph steps (child process: "
                                              "tools/bench_rollout.py --envs 2048 --groups 2)",
                                  "value": d5["env_steps_per_s"], "unit": "env-steps/s", "ms_per_step": d5["ms_per_step"],
                                  "converged_frac_rollout": d5["converged_frac_rollout"], "graph": d5["graph"], "groups": d5["groups"]}
    except Exception as exc:  # noqa: BLE001
        res["config5_one_gpu"] = {"error": f"tools/bench_rollout.py --envs 2048 --groups 2 did not produce a line: {exc}"[:2000]}
    # the same call with HOST pointers (numpy in, numpy out): H2D of the inputs, solve, D2H of u0/status/iters
    ts = []
    for _ in range(6):
        t1 = time.perf_counter()
        eng.solve_batch(inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                        others=inp["others"], collision_cost=True, want_trajectories=False)
        ts.append(time.perf_counter() - t1)
    ms = float(np.median(ts[1:])) * 1e3
    res["pcie_inclusive"] = {"ms": ms, "value": BATCH / (ms * 1e-3), "unit": "solves/s",
                             "note": "mpc_solve_batch with host pointers, wall time of the synchronous call"}
    # observation-level call (device preamble + solve, per-environment detector state): fresh handle, first step of an episode
    obs = torch.as_tensor(synth.make_obs_batch(BATCH, V, seed=0), device=dev)
    w = args["weights"]
    e3 = engine.MPCEngine(horizon=HORIZON, max_iter=MAX_ITER, device=dev.index)
    o3 = e3.predict_batch_torch(obs, w, collision_cost=True, sync=True)

    def pred():
        e3.reset_env_state()
        e3.predict_batch_torch(obs, w, collision_cost=True, out=o3)
    ms = timed(pred, reps=5)
    st3 = o3["status"].cpu().numpy()
    res["predict_batch"] = {"ms": ms, "value": BATCH / (ms * 1e-3), "unit": "env-steps/s",
                            "converged_frac": float(conv_mask(st3).mean()),
                            "note": "mpc_predict_batch (observation in, action out) incl. the reset of the detector state"}
    e3.close()
    # the iterative-linear agent's QP (agents/pure_mpc_linear.py) on the same ego states, first call of an episode
    st_l = args["state"][:, [0, 1, 3, 2]].contiguous()
    U_l = torch.zeros((BATCH, HORIZON, 2), dtype=torch.float64, device=dev)
    o_l = eng.ltv_solve_batch_torch(st_l, U_l, sync=True)

    def ltv():
        U_l.zero_()
        eng.ltv_solve_batch_torch(st_l, U_l, out=o_l)
    ms = timed(ltv, reps=5)
    st_q = o_l["status"].cpu().numpy()
    solved = st_q == 0
    res["ltv_qp"] = {"value": float(solved.sum()) / (ms * 1e-3), "unit": "solves/s", "batch": BATCH, "ms": ms,
                     "value_all_instances": BATCH / (ms * 1e-3),
                     "solved_frac": float(solved.mean()), "speed_out_of_bounds_frac": float((st_q == 3).mean()),
                     "iters_mean": float(o_l["iters"].cpu().numpy()[solved].mean()),
                     "note": "mpc_ltv_solve_batch, reference agents/pure_mpc_linear.py; value counts the QPs that were solved - "
                             "the synthetic ego speed U(0, 12) exceeds that agent's MAX_SPEED 40/3.6 in 7 % of the instances, whose "
                             "QP has no feasible point (status 3, as in the reference); DESIGN.md section 4.5"}
    return res


if __name__ == "__main__":
    main()
