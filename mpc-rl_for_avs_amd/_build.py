"""Build the gfx950 shared library of the engine in-tree with hipcc (no JIT cache, no pip install)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpc_mi355x.so")
_SOURCES = [os.path.join(_HERE, "csrc", "mpc_engine.hip")]
_DEPS = _SOURCES + [os.path.join(_HERE, "csrc", f) for f in ("mpc_core.hpp", "mpc_wave.hpp", "mpc_ltv.hpp",
                                                              "mpc_preamble.hpp", "mpc_preamble_wave.hpp", "mpc_wave_dev.hpp", "mpc_synth_env.hpp", "mpc_rollout_glue.hpp")] + \
    [os.path.join(os.path.dirname(_HERE), "include", "mpc_mi355x.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libmpc_mi355x.so next to this file; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB_PATH + ".tmp"] + _SOURCES
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    if verbose:
        print(res.stderr)
    return LIB_PATH
