"""Fixtures for the iterative-linear (LTV-QP) path from a solver that shares nothing with the kernel or with
oracle/ltv_oracle.py's interior-point method: the Goldfarb-Idnani dual active-set method (oracle/qp_active_set.py) on
the QP assembled from the plain-loop transcription of the cvxpy statements of agents/pure_mpc_linear.py:205-257
(`ltv_oracle.objective_loops` / `constraint_loops`, evaluated at unit vectors).  The only shared pieces are the
reference-pinned helpers that define the QP's data: nearest reference index, reference window, predict_motion
(tests/golden/ltv_reference_numpy.npz pins them to the reference's own functions).

    python tests/golden/make_ltv_independent.py      ->  tests/golden/ltv_independent_solutions.npz

Per horizon T in (20, 12): states from the synthetic generator (speeds inside [0, 40/3.6]); a third of the instances
with a zero stored profile (first call of an agent), a third with a random one, a third with the active-set solution of
the first call as the stored profile (what a second call sees)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mpc_rl_for_avs_amd  # noqa: E402,F401
from mpc_rl_for_avs_amd.reference_path import reference_states  # noqa: E402
import ltv_oracle as L  # noqa: E402
import qp_active_set as Q  # noqa: E402
from conftest import ltv_states  # noqa: E402


def solve_one(ref, state, nominal, T, dt=0.1):
    tgt = L.nearest_index(state[None, 0], state[None, 1], ref)
    xref = L.reference_window(ref, tgt, T)[0]
    xbar = L.nominal_rollout(state[None], nominal[None, :, 0], nominal[None, :, 1], dt)[0]
    H, g, A, b, f0 = Q.build_from_loops(L, state, xref, xbar, dt, T)
    x, mult, act = Q.solve(H, g, A, b)
    fval = L.objective_loops(x.reshape(T, 2), state, xref, xbar, dt)
    assert L.constraint_loops(x.reshape(T, 2), state, xbar, dt).min() >= -1e-9
    # KKT of the exact solution, from the loop-built data
    grad = H @ x + g
    assert np.abs(grad - A.T @ mult).max() <= 1e-8 * max(1.0, np.abs(grad).max()) and mult.min() >= 0.0
    pos = mult[act] if act else np.array([np.inf])
    return x.reshape(T, 2), fval, len(act), float(pos.min()), int(tgt[0])


def main():
    ref = reference_states(0.1)
    out = {}
    for T, n_each, seed in ((20, 56, 101), (12, 16, 202)):
        st = ltv_states(4 * n_each, seed=seed)
        st = st[(st[:, 2] >= 0.0) & (st[:, 2] <= L.MAX_SPEED)][:n_each]
        assert len(st) == n_each
        rng = np.random.default_rng(seed)
        rand = rng.uniform(-1.0, 1.0, (n_each, T, 2)) * np.array([1.5, 0.3])
        states, noms, sols, fvals, nact, mmin, tgts = [], [], [], [], [], [], []
        for b in range(n_each):
            first = solve_one(ref, st[b], np.zeros((T, 2)), T)
            for nominal in (np.zeros((T, 2)), rand[b], first[0]):
                u, fv, na, mm, tg = first if nominal is not rand[b] and not nominal.any() else solve_one(ref, st[b], nominal, T)
                states.append(st[b]); noms.append(nominal); sols.append(u); fvals.append(fv); nact.append(na)
                mmin.append(mm); tgts.append(tg)
            print(f"T={T} instance {b + 1}/{n_each}", flush=True)
        out[f"state_T{T}"] = np.array(states)
        out[f"nominal_T{T}"] = np.array(noms)
        out[f"U_T{T}"] = np.array(sols)
        out[f"objective_T{T}"] = np.array(fvals)
        out[f"n_active_T{T}"] = np.array(nact, dtype=np.int32)
        out[f"min_multiplier_T{T}"] = np.array(mmin)
        out[f"target_index_T{T}"] = np.array(tgts, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "ltv_independent_solutions.npz"), **out)
    for k, v in out.items():
        print(k, v.shape)


if __name__ == "__main__":
    main()
