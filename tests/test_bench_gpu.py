"""bench.py on the GPU box: the single-process line and the launch the driver uses for N > 1 (torch.distributed.run, one
rank per GPU, RCCL) rehearsed with one rank - process-group init on the nccl backend, the barrier / max-over-ranks timing
around the engine's own stream, the rank-0 JSON line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _line(cmd, env=None, only_line=False):
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    if only_line:       # the driver reads rank 0's stdout: nothing but the JSON line may be on it (RCCL's banner is not)
        assert [l for l in res.stdout.splitlines() if l.strip()] == lines, res.stdout[-2000:]
    return json.loads(lines[0])


def _check(d, steps):
    assert d["metric"] == "MPC solves/sec (horizon=20, batch=4096)" and d["unit"] == "solves/s"
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["higher_is_better"] is True and d["dtype"] == "f64"
    assert d["value"] > 1e5 and d["ms_per_step"] > 0 and d["vs_baseline"] is None
    assert d["config"]["workload"] and d["data"] == "synthetic"
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["solver"]["converged_frac"] >= 0.995 and d["solver"]["iters_max"] <= d["config"]["max_iter"] == 100
    # value = converged instances per second; every instance of the batch is in value_all_instances
    assert d["ms_per_step"] * 1e-3 * d["solver"]["value_all_instances"] == pytest.approx(d["config"]["batch_per_gpu"], rel=1e-9)
    assert d["value"] == pytest.approx(d["solver"]["value_all_instances"] * d["solver"]["converged_frac"], rel=1e-9)


def test_single_process_line():
    d = _line([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-side"], only_line=True)
    _check(d, 3)


def test_single_process_line_has_no_process_group():
    """Without torchrun / BENCH_FORCE_DIST the line says so: `distributed` is null (what test_one_rank_over_rccl's
    assertions would catch if that test silently ran this path)."""
    d = _line([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-side"])
    assert d["distributed"] is None


def test_one_rank_over_rccl():
    """The launch the driver uses for N > 1 with one rank: init_process_group("nccl") = RCCL, barrier, the all-gather of
    actions + status after every solve, max-over-ranks timing.  BENCH_FORCE_DIST makes a world of 1 take that path."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_FORCE_DIST="1")
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", "29641", "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--no-side"], env=env, only_line=True)
    _check(d, 3)
    assert "x1" in d["config"]["parallelism"]
    g = d["distributed"]                         # only the process-group path fills this in
    assert g["backend"] == "nccl" and g["world_size"] == 1
    assert g["gathered_actions_shape"] == [4096, 2] and g["gathered_status_shape"] == [4096]
    assert g["own_block_equals_local"] is True
    # every rank's own kernel time, slowest instance and converged count (the step lasts as long as the slowest rank)
    assert len(g["per_rank_kernel_ms"]) == 1 and 0.5 < g["per_rank_kernel_ms"][0] < 50.0
    assert g["per_rank_iters_max"] == [d["solver"]["iters_max"]]
    assert abs(g["per_rank_converged"][0] / 4096 - d["solver"]["converged_frac"]) < 1e-9


def test_config5_shape_rollout_over_rccl():
    """BASELINE config 5 at its per-step workload on the one GPU there is: 2048 environments, MPC in the loop
    (agents/a2c_mpc.py:138-153 is the call being sharded), a 1-rank RCCL group, actions AND status all-gathered every
    step and equal to the local ones.  (8 ranks x 256 environments is the driver's to launch.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_FORCE_DIST="1")
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", "29643", "tools/bench_rollout.py", "--envs", "2048", "--steps", "16",
               "--algorithm", "a2c"], env=env)
    assert d["envs"] == 2048 and d["n_gpus"] == 1 and d["steps_per_env"] == 16
    g = d["distributed"]
    assert g["backend"] == "nccl" and g["gathered_actions_shape"] == [2048, 2] and g["gathered_status_shape"] == [2048]
    assert g["own_block_equals_local"] is True
    assert d["env_steps_per_s"] > 1e5 and d["converged_frac"] > 0.99
    # the RCCL gather is captured INSIDE the step's hipGraph: sharded runs keep the one-launch step
    assert d["graph"] is True and d["graph_fallback_reason"] is None, d["graph_fallback_reason"]
    assert d["fused_glue"] is True


def test_default_line_is_the_only_stdout_and_carries_every_single_gpu_config():
    """The line the driver records (no --no-side, CPU baseline on): nothing else on stdout, `roofline` and `cpu_baseline`
    present, and the driver-visible side objects for the other BASELINE configurations one GPU can produce: config 2
    (B = 1024, V = 4, live objective), config 4 (256-environment rollout), B = 1 predict() latency, the headline over
    seeds 0-2 and the reference's solver settings."""
    d = _line([sys.executable, "bench.py", "--steps", "5", "--warmup", "2"], only_line=True)
    _check(d, 5)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 1e3 and d["cpu_baseline"]["cores"] >= 1
    assert d["roofline"]["kernel_ms"] > 0 and d["parity"]["n_certified"] == d["parity"]["n_converged"]
    s = d["headline_over_seeds"]
    assert [r["seed"] for r in s["per_seed"]] == [0, 1, 2] and s["ms_min"] <= s["ms_median"] <= s["ms_max"] < 20.0
    c2, c4, b1 = d["config2"], d["config4"], d["predict_b1"]
    assert "batch=1024" in c2["workload"] and c2["converged_frac"] > 0.99 and 0.3 < c2["ms"] < 5.0
    assert "256 parallel" in c4["workload"] and c4["unit"] == "env-steps/s" and c4["value"] > 1e5
    assert b1["steps"] >= 50 and 0.1 < b1["ms_median"] < 5.0 and b1["converged_frac"] > 0.9
    ref = d["solver_settings_sweep"]["max_iter 1000, tol 1e-6 (reference)"]
    assert ref["converged_frac"] > 0.995 and ref["iters_max"] <= 1000
    # round 5: what the headline counts, and the reference's settings as a first-class sibling
    assert 0.9 * d["value"] < d["value_smooth_only"] <= d["value"]
    rs = d["reference_settings"]
    assert [r["seed"] for r in rs["per_seed"]] == [0, 1, 2] and rs["ms_max"] < 8.0 and all(r["converged_frac"] > 0.995 for r in rs["per_seed"])
    sd = d["strict_discontinuity"]
    assert sd["kink_unsolved"] > 0 and abs(sd["solved_frac"] - (d["solver"]["smooth_kkt_frac"] + d["solver"]["acceptable_level_frac"])) < 2e-3
    v1 = d["config4_v1"]
    assert v1["unit"] == "env-steps/s" and v1["value"] > 5e4 and v1["converged_frac_rollout"] > 0.95
    # round 6: the regimes beside the headline - batches in flight on streams probed to overlap, one bulk launch, config 5's total
    fl, bl, c5 = d["in_flight"], d["bulk_launch"], d["config5_one_gpu"]
    assert fl["streams"] == 8 and fl["streams_probed_to_overlap"] and fl["identical_outputs_across_streams"] and fl["value"] > 2.0 * d["value"]
    assert bl["unit"] == "solves/s" and bl["value"] > 2.0 * d["value"] and bl["converged_frac"] > 0.995
    assert c5["unit"] == "env-steps/s" and c5["value"] > 1e6 and c5["groups"] == 2 and c5["converged_frac_rollout"] > 0.99
