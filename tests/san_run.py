"""Child process of tests/test_sanitizers.py (run with libasan preloaded and MPC_TEST_SANITIZE=1): the CPU oracle and the
host builds of the three kernel sources (solve, iterative-linear QP, observation preamble) compiled with AddressSanitizer
and UndefinedBehaviorSanitizer, driven over the shapes the interface allows.  The host builds run the KERNEL SOURCE
(mpc_wave.hpp, mpc_ltv.hpp, mpc_preamble.hpp) with every LDS access bounds-checked against lds_doubles() in
tests/host_wave_ctx.hpp, so an indexing error of the kernel for some horizon / vehicle count shows up here, on the CPU.
Any sanitizer report aborts the process; the parent asserts on the exit code."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
assert os.environ.get("MPC_TEST_SANITIZE") == "1"

import conftest  # noqa: E402
import oracle_lib  # noqa: E402
from mpc_rl_for_avs_amd import synth  # noqa: E402
from mpc_rl_for_avs_amd.reference_path import reference_states  # noqa: E402

ref = reference_states(0.1)
wave = conftest._host_solver("libcpu_wave.so", "cpu_wave_harness.cpp", "wave_solve_batch")
n = 0
for N, V, cc, B in ((20, 8, True, 24), (20, 4, False, 24), (16, 3, True, 8), (1, 2, True, 4), (2, 0, False, 4), (33, 9, True, 4),
                    (64, 16, True, 3), (64, 0, False, 3)):
    inp = synth.solver_inputs(B, min(max(V, 1), 9), seed=100 + N)
    if V == 0:
        inp["others"] = None
    elif V > 9:
        inp["others"] = np.ascontiguousarray(np.concatenate([inp["others"], inp["others"] + 5.0], axis=1)[:, :V])
    vr = inp["vref"]
    inp["vref"] = np.ascontiguousarray(np.concatenate([vr, np.repeat(vr[:, -1:], max(N - 20, 0), axis=1)], axis=1)[:, :N + 1])
    got = wave(ref, inp, N=N, collision_cost=cc, max_iter=60)
    want = oracle_lib.solve_batch(ref, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                                  others=inp["others"], collision_cost=cc, N=N, max_iter=60, xy_bounds=False)
    both = conftest.converged(got["status"]) & conftest.converged(want["status"])
    assert both.sum() >= 1 and conftest.rel_u0_err(got["u0"], want["u0"])[both].max() < 1e-4, (N, V, cc)
    n += B
# warm start path
inp = synth.solver_inputs(8, 4, seed=3)
cold = wave(ref, inp, collision_cost=True, max_iter=60)
wave(ref, inp, collision_cost=True, max_iter=60, u_init=np.roll(cold["U"], -1, axis=1))
# iterative-linear QP kernel source
ltv = conftest._ltv_solver()
for N in (5, 20, 64):
    st = conftest.ltv_states(6, seed=N)
    out = ltv(ref, st, np.zeros((6, N, 2)), N=N)
    assert out["u0"].shape == (6, 2)
# observation preamble: episodes with detector memory, 1..17 rows
import test_preamble_cpu as tp  # noqa: E402
lib = tp.load_pre()
for rows in (1, 2, 10, 17):
    dp = tp.DevicePreamble(lib, ref)
    for t in range(4):
        obs = np.zeros((5, rows, 8), np.float32)
        full = synth.make_obs_batch(5, min(rows - 1, 9), seed=40 * rows + t)
        obs[:, :min(rows, 10)] = full[:, :min(rows, 10)]
        if rows > 10:
            obs[:, 10:] = full[:, 1:rows - 9]
        o = dp(obs)
        assert o["state"].shape == (5, 4)
# fused environment step (mpc_synth_env.hpp): every vehicle count, resets and respawns
import ctypes  # noqa: E402
import subprocess  # noqa: E402
import test_synth_env_cpu as te  # noqa: E402
out_env = os.path.join(conftest.BUILD_DIR, "libcpu_synth_env.so")
subprocess.run(["g++"] + conftest.HOST_CXXFLAGS + ["-o", out_env, os.path.join(HERE, "cpu_synth_env_harness.cpp")], check=True)
envlib = ctypes.CDLL(out_env)
rng = np.random.default_rng(0)
for K in (0, 1, 4, 9):
    he = te.HostEnv(envlib, 33, K, seed=K)
    he.reset()
    for _ in range(220):
        he.step(np.stack([rng.uniform(-6, 6, 33), rng.uniform(-0.1, 0.1, 33)], axis=1))
# the wave-per-environment form of the preamble (mpc_preamble_wave.hpp): every LDS word bounds-checked by HostCtx
for rows in (1, 2, 10, 17):
    dp = tp.DevicePreamble(lib, ref, wave=True)
    for t in range(4):
        obs = np.zeros((5, rows, 8), np.float32)
        full = synth.make_obs_batch(5, min(rows - 1, 9), seed=40 * rows + t)
        obs[:, :min(rows, 10)] = full[:, :min(rows, 10)]
        if rows > 10:
            obs[:, 10:] = full[:, 1:rows - 9]
        o = dp(obs)
        assert o["state"].shape == (5, 4)
# rollout glue (mpc_rollout_glue.hpp): policy forward / sample / MPC inputs and the buffer row
import torch  # noqa: E402
import glue_host  # noqa: E402
from mpc_rl_for_avs_amd import rollout  # noqa: E402
for A, version in ((1, "v0"), (3, "v1"), (8, "v1")):
    pol = rollout.ActorCritic(A)
    g = glue_host.policy_act(pol, np.random.default_rng(A).normal(size=(7, 10, 8)).astype(np.float32),
                             np.zeros((7, A), np.float32), version, True)
    assert np.isfinite(g["actions"]).all()
import ctypes  # noqa: E402
for keep in (True, False):            # end of a rollout: truncation bootstrap + GAE over a buffer of odd sizes
    T, B, A = 9, 5, 3
    buf = rollout.RolloutBuffer(T, B, A, "cpu", keep_terminal=keep)
    buf._row.copy_(torch.randn(buf._row.shape))
    lv, d8, tv = torch.randn(B), torch.zeros(B, dtype=torch.uint8), torch.randn(T, B)
    q = lambda t: ctypes.c_void_p(t.data_ptr())
    assert glue_host.load().glue_rollout_finish(T, B, A, buf._cols, 1 if keep else 0, q(buf._row), q(lv), q(d8),
                                                q(tv) if keep else None, 0.99, 0.95, q(buf.advantages), q(buf.returns)) == 0
    assert torch.isfinite(buf.advantages).all()
print(f"sanitized run ok: {n} wave solves, LTV, preamble (both forms), environment and rollout-glue harnesses clean")
