"""The C-ABI shared library: it loads, exports every symbol include/mpc_mi355x.h declares, and refuses to
run without a GPU (CPU only - no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mpc_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mpc_rl_for_avs_amd import engine
    lib = engine.load_library()
    names = _declared_symbols()
    assert {"mpc_create", "mpc_destroy", "mpc_set_reference", "mpc_solve_batch", "mpc_last_error",
            "mpc_version", "mpc_default_config", "mpc_workspace_bytes"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mpc_mi355x.h but not exported"
    import __graft_entry__ as g
    assert lib.mpc_version() == g.abi_version_of_header() == 8
    assert sorted(engine._EXPORTS) == names


def test_default_config_and_argument_checks():
    from mpc_rl_for_avs_amd import engine
    lib = engine.load_library()
    cfg = engine._Config()
    lib.mpc_default_config(ctypes.byref(cfg))
    assert (cfg.horizon, cfg.max_iter, cfg.device) == (20, 100, 0)
    assert cfg.dt == 0.1 and cfg.tol == 1e-8 and cfg.w_distance == 10.0 and cfg.w_collision == 1.0
    assert cfg.struct_size == ctypes.sizeof(engine._Config)
    h = ctypes.c_void_p()
    cfg.horizon = 0
    assert lib.mpc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1          # MPC_ERR_INVALID_ARG
    assert b"horizon" in lib.mpc_last_error()
    cfg.horizon = 20
    cfg.struct_size = 4
    assert lib.mpc_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert lib.mpc_solve_batch(None, 1, None, None, None, None, None, None, 0, 0, None, None, None, None, None,
                               None) == -1
    # the entry points of round 4 refuse bad arguments before they touch a device
    assert lib.mpc_eval_nlp(None, 1, None, None, None, None, None, 0, 0, None, None, None, None) == -1
    buf = (ctypes.c_float * 256)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mpc_rollout_finish(0, 0, 4, 1, 85, 0, p, p, p, None, 0.99, 0.95, p, p, None) == -1      # T = 0
    assert b"1 <= T" in lib.mpc_last_error()
    assert lib.mpc_rollout_finish(0, 9000, 4, 1, 85, 0, p, p, p, None, 0.99, 0.95, p, p, None) == -1   # T > 8192
    assert lib.mpc_rollout_finish(0, 8, 4, 1, 84, 0, p, p, p, None, 0.99, 0.95, p, p, None) == -1      # row layout
    assert lib.mpc_rollout_finish(0, 8, 4, 1, 85, 0, None, p, p, None, 0.99, 0.95, p, p, None) == -1   # null row
    assert lib.mpc_policy_act(0, 4, 0, 128, *([p] * 10), 0, 0, None, 0, 1, p, p, p, None, p, None) == -1   # action_dim 0
    assert lib.mpc_policy_act(0, 4, 3, 128, *([p] * 10), 0, 0, None, 1, 1, p, p, p, None, None, None) == -1  # v1 without weights out
    assert lib.mpc_streams_overlap(0, None, None, None) == -1                 # nowhere to put the answer


def test_sized_default_config_refuses_a_short_struct():
    """A binding that still declares an older (shorter) mpc_config is refused and NOT written to."""
    from mpc_rl_for_avs_amd import engine
    lib = engine.load_library()

    class Old(ctypes.Structure):                       # the 48-byte layout of ABI 1
        _fields_ = engine._Config._fields_[:-2]
    buf = (ctypes.c_uint8 * 64)(*([0xAB] * 64))        # the object plus what lies behind it
    old = Old.from_buffer(buf)
    rc = lib.mpc_default_config_sized(ctypes.cast(ctypes.byref(old), ctypes.POINTER(engine._Config)), ctypes.sizeof(Old))
    assert rc == -1 and b"48 bytes" in lib.mpc_last_error()
    assert bytes(buf) == b"\xab" * 64                  # untouched, in particular behind the 48 bytes
    cfg = engine._Config()
    assert lib.mpc_default_config_sized(ctypes.byref(cfg), ctypes.sizeof(cfg)) == 0
    assert cfg.struct_size == ctypes.sizeof(cfg) == 56 and cfg.ltv_passes == 1 and cfg.stall_window == 0


def test_graft_entry_build_returns():
    """The documented build command (README / INTEGRATION.md: `import __graft_entry__ as g; g.build()`) must not raise."""
    import __graft_entry__ as g
    g.build()


def test_integration_md_binding_matches_the_library():
    """The ctypes stub INTEGRATION.md tells a maintainer to paste (Option B): its `_Cfg` must have the library's
    mpc_config layout, field for field like engine._Config, and its default-config call must succeed."""
    from mpc_rl_for_avs_amd import engine, _build
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"^class _Cfg\(ctypes\.Structure\):.*?\n(?=\ndef )", text, re.S | re.M)
    assert m, "INTEGRATION.md no longer holds the _Cfg snippet"
    ns = {"ctypes": ctypes}
    exec(m.group(0), ns)
    Cfg = ns["_Cfg"]
    assert [(n, t) for n, t in Cfg._fields_] == [(n, t) for n, t in engine._Config._fields_]
    lib = engine.load_library()
    cfg = Cfg()
    call = re.search(r"_lib\.(mpc_default_config\w*)\(ctypes\.byref\(cfg\)(, ctypes\.sizeof\(cfg\))?\)", text)
    assert call and call.group(1) == "mpc_default_config_sized" and call.group(2), "the stub must use the sized call"
    fn = getattr(lib, call.group(1))
    fn.argtypes = None
    assert fn(ctypes.byref(cfg), ctypes.sizeof(cfg)) == 0
    assert cfg.struct_size == ctypes.sizeof(Cfg) and cfg.horizon == 20 and cfg.ltv_passes == 1
    assert f'ctypes.CDLL("{os.path.basename(_build.LIB_PATH)}")' in text


def test_no_gpu_fails_loudly():
    """Without a HIP device the engine must raise, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mpc_rl_for_avs_amd import engine
    with pytest.raises(engine.EngineError, match="no HIP device"):
        engine.MPCEngine()
    from mpc_rl_for_avs_amd.pure_mpc import PureMPC_Agent

    class Env:
        config = {"simulation_frequency": 30, "policy_frequency": 10, "observation": {"vehicles_count": 10}}
    with pytest.raises(engine.EngineError):
        PureMPC_Agent(Env(), dict(horizon=20, render=False, weight_speed=1, weight_control=1, weight_input_diff=1))


def test_product_never_imports_oracle():
    """The shipped package must not reference the oracle (test infrastructure) anywhere."""
    pkg = os.path.join(ROOT, "mpc-rl_for_avs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not any(w in text for w in ("oracle_lib", "nlp_spec", "mpc_oracle", "ltv_oracle")), f


def _c_client():
    """tests/abi_client.c compiled with gcc: the ABI used from plain C (dlopen, no Python objects, no torch)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "_build", "abi_client")
    src = os.path.join(ROOT, "tests", "abi_client.c")
    hdr = os.path.join(ROOT, "include", "mpc_mi355x.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-o", exe, src, "-ldl", "-lm"], check=True)
    return exe


def test_c_client_without_gpu():
    """The header compiles as C, every entry point the client binds resolves, argument errors come back as codes, and
    without a device mpc_create says MPC_ERR_NO_DEVICE (with one it succeeds)."""
    import subprocess
    from mpc_rl_for_avs_amd import _build
    _build.build()
    r = subprocess.run([_c_client(), _build.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "create rc=" in r.stdout


@pytest.mark.gpu
def test_c_client_solves_on_gpu():
    """Known answers through the ABI from plain C: NLP on the straight part of the path, state outside its bounds,
    the iterative-linear QP."""
    import subprocess
    from mpc_rl_for_avs_amd import _build
    _build.build()
    r = subprocess.run([_c_client(), _build.LIB_PATH, "solve"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "create rc=0" in r.stdout and "solve ok" in r.stdout
