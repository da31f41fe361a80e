"""TEST INFRASTRUCTURE - the rollout-glue kernels' per-thread code (csrc/mpc_rollout_glue.hpp) compiled for the host
(tests/cpu_rollout_glue_harness.cpp) behind numpy / torch-CPU wrappers."""
import ctypes
import os
import subprocess

import numpy as np

import conftest

_lib = None


def load():
    global _lib
    if _lib is None:
        out = os.path.join(conftest.BUILD_DIR, "libcpu_rollout_glue.so")
        src = os.path.join(conftest.ROOT, "tests", "cpu_rollout_glue_harness.cpp")
        deps = [os.path.join(conftest.ROOT, "mpc-rl_for_avs_amd", "csrc", f) for f in ("mpc_rollout_glue.hpp", "mpc_synth_env.hpp",
                                                                                      "mpc_core.hpp")]
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in [src] + deps):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            flags = [f for f in conftest.HOST_CXXFLAGS if f != "-ffp-contract=off"]      # fmaf is explicit in this source
            subprocess.run(["g++"] + flags + ["-o", out, src], check=True)
        _lib = ctypes.CDLL(out)
        _lib.glue_policy_act.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 10 + [ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p] + \
            [ctypes.c_int] * 2 + [ctypes.c_void_p] * 5
        _lib.glue_rollout_record.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 20
        _lib.glue_rollout_finish.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4 + [ctypes.c_double] * 2 + \
            [ctypes.c_void_p] * 2
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def policy_act(pol, obs, noise, version="v0", clip=True, draw=None):
    """ActorCritic.act through the kernel's code: obs [B, 10, 8] / noise [B, A] float32 numpy -> dict.
    draw = (seed, env_offset, step): the kernel's own counter-based draws instead (returned as o["noise"])."""
    pol.refresh_fused()
    f = {k: np.ascontiguousarray(v.detach().cpu().numpy(), np.float32) for k, v in pol._fz.items()}
    f["c0"] = f["c0"].reshape(1)
    B, A, H2 = obs.shape[0], pol.action_dim, f["b1"].size
    obs = np.ascontiguousarray(obs.reshape(B, -1), np.float32)
    noise = np.ascontiguousarray(noise, np.float32)
    o = dict(actions=np.zeros((B, A), np.float32), values=np.zeros(B, np.float32), log_probs=np.zeros(B, np.float32),
             weights=np.full((B, 3), np.nan), ref_speed=np.full(B, np.nan))
    v1 = version == "v1"
    rc = load().glue_policy_act(B, A, H2, _p(obs), _p(f["w1"]), _p(f["b1"]), _p(f["w2"]), _p(f["b2"]), _p(f["wh"]), _p(f["bh"]),
                                _p(f["std"]), _p(f["c0"]), _p(noise), 0 if draw is None else int(draw[0]),
                                0 if draw is None else int(draw[1]), None if draw is None else _p(np.array([draw[2]], np.int64)),
                                1 if v1 else 0, 1 if clip else 0, _p(o["actions"]),
                                _p(o["values"]), _p(o["log_probs"]), _p(o["weights"]) if v1 else None,
                                None if v1 else _p(o["ref_speed"]))
    assert rc == 0
    o["noise"] = noise
    return o
