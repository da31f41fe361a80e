#!/usr/bin/env python3
"""bench.py - MPC solves/sec of the MI355X engine on BASELINE.json's headline configuration.

A "step" is one pass of the hot path (`mpc_solve_batch`) over one batch of synthetic instances that is already
resident in HBM: configs[2] of BASELINE.json = batch 4096, horizon 20, 8 other vehicles, collision cost on.
With N GPUs every rank owns its own 4096 instances (weak scaling) and the step ends with the RCCL all-gather of
the actions, the path's only exchange.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os

# read when the HIP runtime starts: with the default of 4 hardware queues, streams carrying independent batches can
# end up sharing a queue and serialise ("batches in flight"); plain kernel launches only - replayed
# hipGraphs (tools/bench_rollout.py --graph) measured slower with 8
if int(os.environ.get("WORLD_SIZE", "1")) == 1:      # multi-rank runs (RCCL) keep the runtime's default
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096
HORIZON = 20
V = 8
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_SOLVE = 756 + 32 * V   # inputs + u0/status/iters, FP64


def cpu_baseline(inp, sample: int):
    """The CPU oracle (a from-scratch port of the same NLP + algorithm, oracle/mpc_oracle.c) on the host cores.
    OpenMP over instances; the thread count with the best wall time is reported (more threads than physical
    cores only adds scheduling noise on this short job)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    from mpc_rl_for_avs_amd.reference_path import reference_states
    ncpu = os.cpu_count() or 1
    sl = slice(0, sample)
    ref = reference_states()
    best = None
    for cores in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 32)}, reverse=True):
        for _ in range(2):
            t0 = time.perf_counter()
            out = oracle_lib.solve_batch(ref, inp["state"][sl], inp["ego_index"][sl], inp["weights"][sl],
                                         inp["is_collide"][sl], vref=inp["vref"][sl], others=inp["others"][sl],
                                         collision_cost=True, max_iter=100, xy_bounds=False, nthreads=cores)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, cores, out)
    dt, cores, out = best
    return dict(value=sample / dt, unit="solves/s", cores=cores, kind="port",
                sample=f"first {sample} instances of the same batch, oracle/mpc_oracle.c (OpenMP over instances, "
                       f"{cores} of {ncpu} hardware threads, best wall time {dt:.2f} s), "
                       f"mean {float(out['iters'].mean()):.1f} iterations"), out


def pmc_counter(name):
    """Mean per launch of one counter from the committed PMC passes (profiles/r*_pmc_summary.csv), or None."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")))
    if not files:
        return None
    vals = {r["counter"]: float(r["mean_per_launch"]) for r in csv.DictReader(open(files[-1]))}
    return vals.get(name)


def pmc_traffic_bytes():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs of this
    same command, profiles/r*_pmc_summary.csv, KB units). The loads are 8-byte strided, outside the access widths
    MI355X_MICROARCH.md calibrates, so the raw counters are used (no x2 correction)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")))
    if not files:
        return None
    vals = {r["counter"]: float(r["mean_per_launch"]) for r in csv.DictReader(open(files[-1]))}
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side measurements (in_flight, ltv_qp): profiling runs, so that every launch of the solve "
                         "kernel in the trace is one of the warm-up / timed steps")
    ap.add_argument("--streams", type=int, default=1,
                    help="batches in flight in the timed loop (default 1: one batch at a time, the headline definition)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from mpc_rl_for_avs_amd import engine, synth, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the latter rehearses the RCCL path on 1 GPU
    if use_dist:
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
            os.environ["NCCL_DEBUG"] = "WARN"        # keep RCCL's version banner off stdout: rank 0 prints ONE line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    inp = synth.solver_inputs(BATCH, V, seed=rank, N=HORIZON)
    t = lambda x, dt_: torch.as_tensor(np.ascontiguousarray(x), dtype=dt_, device=dev)
    args = dict(state=t(inp["state"], torch.float64), ego_index=t(inp["ego_index"], torch.int32),
                weights=t(inp["weights"], torch.float64), is_collide=t(inp["is_collide"], torch.uint8),
                vref=t(inp["vref"], torch.float64), others=t(inp["others"], torch.float64), collision_cost=True)
    eng = engine.MPCEngine(horizon=HORIZON, max_iter=100, device=local_rank)
    out = dict(u0=torch.empty((BATCH, 2), dtype=torch.float64, device=dev),
               status=torch.empty(BATCH, dtype=torch.int32, device=dev),
               iters=torch.empty(BATCH, dtype=torch.int32, device=dev))
    gathered = None

    n_str = max(1, a.streams)
    side = [torch.cuda.Stream(dev) for _ in range(n_str)] if n_str > 1 else [torch.cuda.current_stream(dev)]
    outs_s = [out] + [dict((k, torch.empty_like(v)) for k, v in out.items()) for _ in range(n_str - 1)]

    def step(i=0):
        nonlocal gathered
        with torch.cuda.stream(side[i % n_str]):
            eng.solve_batch_torch(**args, out=outs_s[i % n_str])      # enqueued on that stream
            if use_dist:
                gathered = sharding.all_gather_actions(outs_s[i % n_str]["u0"])

    for i in range(max(a.warmup, n_str if n_str > 1 else 0)):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    # kernel-only events (same stream the kernel is launched on = torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        with torch.cuda.stream(side[i % n_str]):
            ev[i][0].record()
            eng.solve_batch_torch(**args, out=outs_s[i % n_str])
            ev[i][1].record()
            if use_dist:
                gathered = sharding.all_gather_actions(outs_s[i % n_str]["u0"])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        assert gathered.shape == (world * BATCH, 2)
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
    status = out["status"].cpu().numpy()
    iters = out["iters"].cpu().numpy()

    if rank == 0:
        value = world * BATCH * a.steps / elapsed
        achieved = BATCH * ALG_BYTES_PER_SOLVE / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": "MPC solves/sec (horizon=20, batch=4096)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: batch=4096 pure_mpc horizon=20, 8 other vehicles, "
                                   "collision cost on, cold start, tol 1e-8, max_iter 100; per-GPU batch fixed",
                       "batch_per_gpu": BATCH, "horizon": HORIZON, "n_vehicles": V, "seed": "rank",
                       "parallelism": f"instance-sharded x{world}, all-gather of actions" +
                                      (f", {n_str} batches in flight on {n_str} streams" if n_str > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(),
                         "kernel": "mpc_solve_wave_kernel<CC=1,N=20>", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_solve": ALG_BYTES_PER_SOLVE,
                         "note": "path is bound by the serial FP64 + LDS-latency chain of its slowest instance; the working "
                                 "set is LDS-resident and HBM carries only inputs/outputs (DESIGN.md section 4)"},
            "solver": {"converged_frac": float(((status == 0) | (status == 5)).mean()),
                       "smooth_kkt_frac": float((status == 0).mean()), "on_kink_frac": float((status == 5).mean()),
                       "iters_mean": float(iters.mean()), "iters_p99": float(np.percentile(iters, 99)),
                       "iters_max": int(iters.max())},
        }
        # the resource this kernel actually consumes: vector-instruction issue slots (a wave64 FP64 instruction
        # occupies its SIMD for 4 cycles).  SIMD-cycles needed = wave-instructions (PMC) x 4; available = SIMDs x clock
        # x kernel time.  Small because the batch ends with a few lone waves.
        valu = pmc_counter("SQ_INSTS_VALU")
        if valu is not None:
            prop = torch.cuda.get_device_properties(dev)
            simds = 4 * prop.multi_processor_count
            clock_hz = 1e3 * float(getattr(prop, "clock_rate", 2.4e6))
            res["valu_issue"] = {"wave_instructions_per_launch": valu, "simds": simds, "clock_ghz": clock_hz / 1e9,
                                 "frac": valu * 4.0 / (simds * clock_hz * kern_ms * 1e-3),
                                 "note": "fraction of the GPU's vector-instruction issue slots used during the kernel"}
        if world == 1 and not a.no_cpu_baseline:
            cb, oref = cpu_baseline(inp, sample=BATCH)
            res["cpu_baseline"] = cb
            u0 = out["u0"].cpu().numpy()
            both = ((status == 0) | (status == 5)) & ((oref["status"] == 0) | (oref["status"] == 5))
            err = np.abs(u0 - oref["u0"]).max(axis=1) / np.maximum(1.0, np.abs(oref["u0"]).max(axis=1))
            res["parity"] = {"both_converged": int(both.sum()), "u0_rel_linf_max": float(err[both].max()),
                             "u0_rel_linf_p99": float(np.percentile(err[both], 99)),
                             "frac_within_1e-4": float((err[both] <= 1e-4).mean())}
        if world == 1 and not a.no_side:
            # side measurement (not the metric): the same batch solves with 6 of them in flight, round-robin on 6 HIP
            # streams - the straggler tail of one batch (a few lone waves, GPU mostly idle) overlaps with the bulk of the
            # next ones.  Same kernel, same inputs, identical outputs; what a serving loop with several independent
            # environment groups would run.
            n_fl = 6        # HIP spreads streams over 4 hardware queues: 6 streams keep all of them busy whatever the mapping
            streams = [torch.cuda.Stream(dev) for _ in range(n_fl)]
            outs = []
            for sq in streams:
                with torch.cuda.stream(sq):
                    outs.append(eng.solve_batch_torch(**args))
            torch.cuda.synchronize()
            k_fl = max(3 * a.steps, 12 * n_fl)      # long enough that the unoverlapped tail of the last batches is small
            t1 = time.perf_counter()
            for i in range(k_fl):
                with torch.cuda.stream(streams[i % n_fl]):
                    eng.solve_batch_torch(**args, out=outs[i % n_fl])
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            res["in_flight"] = {"streams": n_fl, "steps": k_fl, "value": BATCH * k_fl / el, "unit": "solves/s",
                                "ms_per_batch": el / k_fl * 1e3,
                                "identical_outputs": bool(all(torch.equal(o["u0"], out["u0"]) for o in outs)),
                                "note": "throughput with 6 batches of 4096 in flight; `value` above is one batch at a time"}
            # side measurement (not the metric): the iterative-linear agent's QP (agents/pure_mpc_linear.py) on the same
            # ego states, first call of an episode (zero stored profile), device-resident inputs
            st_l = args["state"][:, [0, 1, 3, 2]].contiguous()
            U_l = torch.zeros((BATCH, HORIZON, 2), dtype=torch.float64, device=dev)
            o_l = eng.ltv_solve_batch_torch(st_l, U_l, sync=True)
            ts = []
            for _ in range(5):
                U_l.zero_()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.ltv_solve_batch_torch(st_l, U_l, out=o_l)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            st_q = o_l["status"].cpu().numpy()
            res["ltv_qp"] = {"value": BATCH / (float(np.median(ts)) * 1e-3), "unit": "solves/s", "batch": BATCH,
                             "ms": float(np.median(ts)), "solved_frac": float((st_q == 0).mean()),
                             "speed_out_of_bounds_frac": float((st_q == 3).mean()),
                             "iters_mean": float(o_l["iters"].cpu().numpy()[st_q == 0].mean()),
                             "note": "mpc_ltv_solve_batch, reference agents/pure_mpc_linear.py; DESIGN.md section 4.5"}
        print(json.dumps(res), flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
