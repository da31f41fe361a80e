"""TEST INFRASTRUCTURE - ctypes loader for the CPU oracle (oracle/mpc_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("MPC_TEST_SANITIZE") == "1"      # ASan + UBSan build (tests/test_sanitizers.py)
_LIB_PATH = os.path.join(_HERE, "_build", "san" if _SAN else "", "libmpc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mpc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["san"] if _SAN else []) + (["-B"] if force else []), check=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int32)
        bp = ctypes.POINTER(ctypes.c_uint8)
        _lib.oracle_solve_batch.restype = ctypes.c_int
        _lib.oracle_solve_batch.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, ctypes.c_int, dp, ip, dp, dp, bp, dp,
            ctypes.c_int, ctypes.c_uint32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
            dp, dp, dp, dp, ip, ip, dp, ctypes.c_int]
        _lib.oracle_solve_batch_warm.restype = ctypes.c_int
        _lib.oracle_solve_batch_warm.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, ctypes.c_int, dp, ip, dp, dp, bp, dp,
            ctypes.c_int, ctypes.c_uint32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
            dp, dp, dp, dp, dp, ip, ip, dp, ctypes.c_int]
        _lib.oracle_set_stall_window.restype = None
        _lib.oracle_set_stall_window.argtypes = [ctypes.c_int]
        _lib.oracle_last_work.restype = None
        _lib.oracle_last_work.argtypes = [dp]
    return _lib


def last_work():
    """Work of the last solve_batch call, counted by the oracle: dict(iterations, sweeps, rollouts, solves, flops,
    transcendentals) - flops / transcendentals by the per-stage operation counts documented in mpc_oracle.c."""
    out = np.zeros(6)
    _load().oracle_last_work(out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return dict(zip(("iterations", "sweeps", "rollouts", "solves", "flops", "transcendentals"), out.tolist()))


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ty))


def solve_batch(ref_table, state, ego_index, weights, is_collide, vref=None, others=None, N=20, dt=0.1,
                collision_cost=False, w_distance=10.0, w_collision=1.0, tol=1e-8, max_iter=200, nthreads=0,
                xy_bounds=True, u_init=None, stall_window=0):
    """Solve B instances on the CPU. Returns dict(u0, U, X, lam, status, iters, kkt).

    xy_bounds=False drops the |x|,|y| <= 500 bounds of the reference NLP (agents/pure_mpc.py:272-274), which
    can never be active for a horizon of N*dt seconds starting inside the intersection; the GPU kernel does
    the same, tests/ check that both settings give the same controls."""
    lib = _load()
    lib.oracle_set_stall_window(int(stall_window))      # the engine's optional progress guard (mpc_config.stall_window)
    ref_table = np.ascontiguousarray(ref_table, dtype=np.float64)
    state = np.ascontiguousarray(state, dtype=np.float64)
    B = state.shape[0]
    ego_index = np.ascontiguousarray(ego_index, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    is_collide = np.ascontiguousarray(is_collide, dtype=np.uint8)
    if vref is not None:
        vref = np.ascontiguousarray(vref, dtype=np.float64)
        assert vref.shape == (B, N + 1)
    V = 0
    if others is not None:
        others = np.ascontiguousarray(others, dtype=np.float64)
        V = others.shape[1]
    u0 = np.zeros((B, 2)); U = np.zeros((B, N, 2)); X = np.zeros((B, N + 1, 4)); lam = np.zeros((B, N + 1, 4))
    status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32); kkt = np.zeros(B)
    if u_init is not None:           # warm start (not in the reference): initial controls [B, N, 2]
        u_init = np.ascontiguousarray(u_init, dtype=np.float64)
        assert u_init.shape == (B, N, 2)
    rc = lib.oracle_solve_batch_warm(
        B, N, dt, _p(ref_table, ctypes.c_double), ref_table.shape[0], _p(state, ctypes.c_double),
        _p(ego_index, ctypes.c_int32), _p(vref, ctypes.c_double), _p(weights, ctypes.c_double),
        _p(is_collide, ctypes.c_uint8), _p(others, ctypes.c_double), V,
        (1 if collision_cost else 0) | (0 if xy_bounds else 2),
        w_distance, w_collision, tol, max_iter, _p(u_init, ctypes.c_double), _p(u0, ctypes.c_double), _p(U, ctypes.c_double),
        _p(X, ctypes.c_double), _p(lam, ctypes.c_double), _p(status, ctypes.c_int32),
        _p(iters, ctypes.c_int32), _p(kkt, ctypes.c_double), nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle_solve_batch failed rc={rc}")
    return dict(u0=u0, U=U, X=X, lam=lam, status=status, iters=iters, kkt=kkt)
