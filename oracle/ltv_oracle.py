"""TEST INFRASTRUCTURE - CPU oracle for the iterative-linear (LTV-QP) MPC agent of the reference.

Only tests/, __graft_entry__.smoke() and tools that time a CPU baseline may import this; the product never does.

Restates reference agents/pure_mpc_linear.py:
  constants                         :23-36   (R, Rd, Q_v_yaw, Qf, MAX_STEER, MAX_DSTEER, MAX_ACCEL, MAX_DECEL, MAX_SPEED)
  nearest_index                     :38-60   calc_nearest_index_in_direction (start_idx = 0, first minimum wins)
  linear_model                      :62-82   linear_model_matrix (steer_ref = 0, C = 0: the model has NO affine term)
  nominal_rollout                   :84-110  predict_motion (speed clamped, yaw uses the updated speed, x/y the updated yaw)
  reference window                  :178-187
  QP (cost, dynamics, bounds)       :205-257 _linear_mpc_control
  failure -> action (0, 0), profile kept   :193-196, 261-263
The reference hands the QP to cvxpy -> ECOS (:259), neither of which exists offline, and has no
test or recorded output for this path.  What pins this file: the reference's own numpy helpers (linear_model_matrix,
predict_motion, calc_nearest_index_in_direction) imported from /root/reference by tests/golden/make_golden.py ->
tests/golden/ltv_reference_numpy.npz; since round 4 the reference's own QP STATEMENTS (`_linear_mpc_control`,
:205-257) executed with a numeric cvxpy stand-in (tests/golden/standins.py) -> tests/golden/ltv_reference_random.npz:
objective to 1e-10, constraint slacks to 1e-9 (tests/test_reference_vectors.py::test_ltv_oracle_equals_the_references_
helpers_and_qp).  The SOLVER (ECOS) is PARITY UNPINNED by the reference; instead: an independent objective/constraint
evaluator written as plain loops over the cvxpy statements (`objective_loops`, `constraint_loops`) driving scipy
SLSQP, and KKT certificates of the returned solutions (tests/test_ltv_oracle.py).  The QP is strictly convex in the
controls (R > 0), so its minimiser is unique and any correct solver returns the same controls up to its tolerance.

Solver: the states are eliminated through the (linear) dynamics and the dense QP in the 2T controls is solved by an
infeasible-start primal-dual interior-point method with Mehrotra's predictor-corrector (one common step length) -
the same iteration the HIP kernel runs, with dense LU solves of the augmented system in (du, dz) where the kernel
runs a Riccati sweep.  Stopping rule (same in the kernel): |c - s|_inf <= 1e-9, |grad L|_inf <= 1e-9 max(1e3, start),
s.z / m <= 1e-10, all at one iterate.
"""
from __future__ import annotations

import math

import numpy as np

NX, NU = 4, 2                       # state (x, y, v, yaw), input (acceleration, steer)    pure_mpc_linear.py:23-24
R_DIAG = (0.01, 0.01)               # :27
RD_DIAG = (0.01, 1.0)               # :28
Q_V, Q_YAW = 20.0, 0.5              # :29
QF_DIAG = (1.0, 1.0, 0.0, 0.5)      # :30, scaled by the horizon (:134)
MAX_STEER = math.radians(30.0)      # :33
MAX_DSTEER = math.radians(30.0)     # :34
MAX_ACCEL = 2.0                     # :35
MAX_DECEL = -5.0                    # :36
MAX_SPEED = 40 / 3.6                # :37
WHEELBASE = 2.5                     # :131 default

STATUS_CONVERGED, STATUS_MAX_ITER, STATUS_FACTORIZATION, STATUS_INFEASIBLE = 0, 1, 2, 3
S_INIT_MIN, Z_INIT = 1.0, 100.0    # initial slacks max(c, S_INIT_MIN), initial multipliers
TOL_P, TOL_D_REL, TOL_MU = 1e-9, 1e-9, 1e-10   # |c - s|_inf, |grad L|_inf / max(1e3, its value at the start), s.z / m


def nearest_index(px, py, ref):
    """pure_mpc_linear.py:38-60 with start_idx = 0: first index of the smallest squared distance."""
    dx = ref[:, 0][None, :] - np.asarray(px, dtype=np.float64)[:, None]
    dy = ref[:, 1][None, :] - np.asarray(py, dtype=np.float64)[:, None]
    return np.argmin(dx * dx + dy * dy, axis=1).astype(np.int32)


def linear_model(v_bar, yaw_bar, dt, wheelbase=WHEELBASE):
    """pure_mpc_linear.py:62-82 for arrays of operating points; steer_ref = 0 (:224).  Returns A [...,4,4], B [...,4,2]."""
    v_bar = np.asarray(v_bar, dtype=np.float64)
    yaw_bar = np.asarray(yaw_bar, dtype=np.float64)
    A = np.zeros(v_bar.shape + (4, 4))
    A[..., 0, 0] = A[..., 1, 1] = A[..., 2, 2] = A[..., 3, 3] = 1.0
    A[..., 0, 2] = dt * np.cos(yaw_bar)
    A[..., 0, 3] = -dt * v_bar * np.sin(yaw_bar)
    A[..., 1, 2] = dt * np.sin(yaw_bar)
    A[..., 1, 3] = dt * v_bar * np.cos(yaw_bar)
    A[..., 3, 2] = dt * math.tan(0.0) / wheelbase
    Bm = np.zeros(v_bar.shape + (4, 2))
    Bm[..., 2, 0] = dt
    Bm[..., 3, 1] = dt * v_bar / (wheelbase * math.cos(0.0) ** 2)
    return A, Bm


def nominal_rollout(x0, oa, od, dt, wheelbase=WHEELBASE):
    """pure_mpc_linear.py:84-110 for a batch: x0 [B,4], oa/od [B,T] -> xbar [B,T+1,4]."""
    x0 = np.asarray(x0, dtype=np.float64)
    B, T = oa.shape
    xbar = np.zeros((B, T + 1, 4))
    xbar[:, 0] = x0
    x, y, v, yaw = (x0[:, i].copy() for i in range(4))
    for i in range(T):
        v = v + oa[:, i] * dt
        v = np.maximum(0.0, np.minimum(v, MAX_SPEED))
        yaw = yaw + (v / wheelbase) * np.tan(od[:, i]) * dt
        x = x + v * np.cos(yaw) * dt
        y = y + v * np.sin(yaw) * dt
        xbar[:, i + 1] = np.stack([x, y, v, yaw], axis=1)
    return xbar


def reference_window(ref, target, T):
    """pure_mpc_linear.py:178-187: rows min(target + i, M - 1), columns reordered to (x, y, v, heading)."""
    idx = np.minimum(target[:, None] + np.arange(T + 1)[None, :], ref.shape[0] - 1)
    w = ref[idx]                      # [B, T+1, 4] columns x, y, v, heading already (base_agent.py:118-154)
    return w


def condense(x0, A, Bm):
    """x_t = xfree_t + M_t u for the model x_{t+1} = A_t x_t + B_t u_t (no affine term): xfree [B,T+1,4], M [B,T+1,4,2T]."""
    Bn, T = A.shape[0], A.shape[1]
    xfree = np.zeros((Bn, T + 1, 4))
    M = np.zeros((Bn, T + 1, 4, 2 * T))
    xfree[:, 0] = x0
    for t in range(T):
        xfree[:, t + 1] = np.einsum("bij,bj->bi", A[:, t], xfree[:, t])
        M[:, t + 1] = np.einsum("bij,bjk->bik", A[:, t], M[:, t])
        M[:, t + 1, :, 2 * t:2 * t + 2] += Bm[:, t]
    return xfree, M


def build_qp(x0, xref, xbar, dt, wheelbase=WHEELBASE):
    """Condensed form of pure_mpc_linear.py:205-257: minimise 1/2 u'Hu + g'u subject to c0 + C u >= 0.

    Constraint rows of stage t (8 per stage, the two rate rows of stage 0 do not exist and are left out):
    a_t - MAX_DECEL, MAX_ACCEL - a_t, d_t + MAX_STEER, MAX_STEER - d_t            (:246-251)
    r - (d_t - d_{t-1}), r + (d_t - d_{t-1}), r = MAX_DSTEER dt                    (:235-238)
    v_{t+1}, MAX_SPEED - v_{t+1}                                                   (:252-256; node 0 is a constant)
    """
    Bn, T = xbar.shape[0], xbar.shape[1] - 1
    A, Bm = linear_model(xbar[:, :T, 2], xbar[:, :T, 3], dt, wheelbase)
    xfree, M = condense(x0, A, Bm)
    n = 2 * T
    H = np.zeros((Bn, n, n))
    g = np.zeros((Bn, n))
    for t in range(T + 1):
        q = np.array([0.0, 0.0, Q_V, Q_YAW]) if t < T else T * np.array(QF_DIAG)
        Mt = M[:, t]
        H += 2.0 * np.einsum("bik,i,bil->bkl", Mt, q, Mt)
        g += 2.0 * np.einsum("bik,i,bi->bk", Mt, q, xfree[:, t] - xref[:, t])
    for t in range(T):
        for j in range(2):
            H[:, 2 * t + j, 2 * t + j] += 2.0 * R_DIAG[j]
            if t < T - 1:
                a, b = 2 * t + j, 2 * t + 2 + j
                H[:, a, a] += 2.0 * RD_DIAG[j]
                H[:, b, b] += 2.0 * RD_DIAG[j]
                H[:, a, b] -= 2.0 * RD_DIAG[j]
                H[:, b, a] -= 2.0 * RD_DIAG[j]
    rows_C, rows_c0 = [], []
    rate = MAX_DSTEER * dt
    for t in range(T):
        e = np.zeros((Bn, 8, n))
        c0 = np.zeros((Bn, 8))
        e[:, 0, 2 * t] = 1.0; c0[:, 0] = -MAX_DECEL
        e[:, 1, 2 * t] = -1.0; c0[:, 1] = MAX_ACCEL
        e[:, 2, 2 * t + 1] = 1.0; c0[:, 2] = MAX_STEER
        e[:, 3, 2 * t + 1] = -1.0; c0[:, 3] = MAX_STEER
        if t >= 1:
            e[:, 4, 2 * t + 1] = -1.0; e[:, 4, 2 * t - 1] = 1.0; c0[:, 4] = rate
            e[:, 5, 2 * t + 1] = 1.0; e[:, 5, 2 * t - 1] = -1.0; c0[:, 5] = rate
        e[:, 6] = M[:, t + 1, 2]; c0[:, 6] = xfree[:, t + 1, 2]
        e[:, 7] = -M[:, t + 1, 2]; c0[:, 7] = MAX_SPEED - xfree[:, t + 1, 2]
        keep = slice(0, 8) if t >= 1 else [0, 1, 2, 3, 6, 7]
        rows_C.append(e[:, keep])
        rows_c0.append(c0[:, keep])
    C = np.concatenate(rows_C, axis=1)
    c0 = np.concatenate(rows_c0, axis=1)
    return dict(H=H, g=g, C=C, c0=c0, xfree=xfree, M=M, A=A, B=Bm)


def _max_step(v, dv):
    """largest alpha in [0, 1] with v + alpha dv >= 0, per instance"""
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(dv < 0.0, -v / dv, np.inf)
    return np.minimum(1.0, r.min(axis=1))


def solve_qp(H, g, C, c0, u_start, max_iter=50):
    """Mehrotra predictor-corrector on the condensed QP, batched.  Returns u, z, status, iters."""
    Bn, n = g.shape
    m = c0.shape[1]
    u = u_start.copy()
    c = c0 + np.einsum("bmn,bn->bm", C, u)
    s = np.maximum(c, S_INIT_MIN)
    z = np.full((Bn, m), Z_INIT)
    status = np.full(Bn, STATUS_MAX_ITER, dtype=np.int32)
    iters = np.zeros(Bn, dtype=np.int32)
    active = np.ones(Bn, dtype=bool)
    for it in range(max_iter + 1):
        c = c0 + np.einsum("bmn,bn->bm", C, u)
        r_p = c - s
        grad = np.einsum("bkl,bl->bk", H, u) + g
        r_d = grad - np.einsum("bmn,bm->bn", C, z)
        mu = (s * z).sum(axis=1) / m
        if it == 0:
            tol_d = TOL_D_REL * np.maximum(1e3, np.abs(r_d).max(axis=1))
        dual_ok = np.abs(r_d).max(axis=1) <= tol_d
        done = (np.abs(r_p).max(axis=1) <= TOL_P) & dual_ok & (mu <= TOL_MU)
        newly = active & done
        status[newly] = STATUS_CONVERGED
        iters[newly] = it
        active &= ~done
        if it == max_iter or not active.any():
            break
        # Newton step from the augmented system in (du, dz),
        #   H du - C' dz = -r_d,   C du + (s/z) dz = -r_p + r_c / z     (ds eliminated through z ds + s dz = r_c),
        # rows scaled so that no entry exceeds 1.  Unlike the normal equations H + C' (z/s) C its conditioning does
        # not degrade as s -> 0 on the active rows (s/z -> 0 there), so dz does not pick up z/s times a rounding error.
        rho = 1.0 / np.maximum(1.0, s / z)
        K = np.zeros((Bn, n + m, n + m))
        K[:, :n, :n] = H
        K[:, :n, n:] = -np.swapaxes(C, 1, 2)
        K[:, n:, :n] = C * rho[:, :, None]
        K[:, n:, n:] = np.einsum("bm,mk->bmk", rho * s / z, np.eye(m))
        bad = active & ~np.isfinite(K).all(axis=(1, 2))
        status[bad] = STATUS_FACTORIZATION
        iters[bad] = it
        active &= ~bad
        K[~active] = np.eye(n + m)             # finished instances ride along with a zero step

        def newton(r_c):
            rhs = np.concatenate([-r_d, rho * (-r_p + r_c / z)], axis=1)
            sol = np.linalg.solve(K, rhs[..., None])[..., 0]
            du_, dz_ = sol[:, :n], sol[:, n:]
            ds_lin = np.einsum("bmn,bn->bm", C, du_) + r_p          # accurate where the slack is large
            ds_cmp = (r_c - s * dz_) / z                              # accurate where the multiplier is large
            return du_, np.where(s > z, ds_lin, ds_cmp), dz_

        du, ds, dz = newton(-s * z)
        a_aff = np.minimum(_max_step(s, ds), _max_step(z, dz))
        mu_aff = ((s + a_aff[:, None] * ds) * (z + a_aff[:, None] * dz)).sum(axis=1) / m
        sigma = np.maximum((mu_aff / mu) ** 3, 0.1 * TOL_MU / mu)     # never aim below the stopping threshold
        du, ds2, dz2 = newton(sigma[:, None] * mu[:, None] - ds * dz - s * z)
        alpha = np.minimum(1.0, 0.99 * np.minimum(_max_step_unbounded(s, ds2), _max_step_unbounded(z, dz2)))
        alpha = np.where(active, alpha, 0.0)[:, None]
        u = u + alpha * du
        s = s + alpha * ds2
        z = z + alpha * dz2
    iters[active & (status == STATUS_MAX_ITER)] = max_iter
    return u, z, status, iters


def _max_step_unbounded(v, dv):
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(dv < 0.0, -v / dv, np.inf)
    return r.min(axis=1)


def solve_batch(ref_table, state, u_nominal, dt=0.1, max_iter=50, wheelbase=WHEELBASE, passes=1):
    """One call of IterativeLinearMPC_Agent._solve (pure_mpc_linear.py:153-203) for B instances.

    state [B,4] = (x, y, v, yaw) of the ego vehicle; u_nominal [B,T,2] = the stored profile (oa, od) (zeros on the
    first call, :190-192); passes = trip count of the loop at :189 (1 in the reference): every pass re-simulates the
    profile the previous one stored and solves the QP linearised about it, the first failing pass ends the call with
    the action (0, 0) and the profile stored so far (:193-196).  Returns dict(u0 [B,2], U [B,T,2] new profile, X
    [B,T+1,4], status, iters (summed over the passes), target_index, z multipliers, u_raw, xref, and xbar of the last
    pass each instance ran)."""
    ref_table = np.asarray(ref_table, dtype=np.float64)
    state = np.asarray(state, dtype=np.float64)
    U = np.array(u_nominal, dtype=np.float64, copy=True)
    Bn, T = U.shape[0], U.shape[1]
    target = nearest_index(state[:, 0], state[:, 1], ref_table)
    xref = reference_window(ref_table, target, T)[:, :, [0, 1, 2, 3]]
    infeasible = (state[:, 2] < 0.0) | (state[:, 2] > MAX_SPEED)        # x[2, 0] == v0 against :252-256
    x0 = state.copy()
    x0[infeasible, 2] = np.clip(x0[infeasible, 2], 0.0, MAX_SPEED)    # keeps the batch finite; their result is discarded
    status = np.full(Bn, STATUS_CONVERGED, dtype=np.int32)
    iters = np.zeros(Bn, dtype=np.int32)
    live = np.ones(Bn, dtype=bool)                                      # no pass has failed yet
    X = np.zeros((Bn, T + 1, 4))
    xbar = np.zeros((Bn, T + 1, 4))
    z = u_raw = None
    for _ in range(passes):
        xbar_p = nominal_rollout(state, U[:, :, 0], U[:, :, 1], dt, wheelbase)
        qp = build_qp(x0, xref, xbar_p, dt, wheelbase)
        u_start = np.zeros((Bn, 2 * T))  # the nominal profile fixes the model only; the minimiser does not depend on the start
        u, z_p, st_p, it_p = solve_qp(qp["H"], qp["g"], qp["C"], qp["c0"], u_start, max_iter=max_iter)
        st_p[infeasible] = STATUS_INFEASIBLE
        it_p[infeasible] = 0
        if z is None:
            z, u_raw = z_p.copy(), u.reshape(Bn, T, 2).copy()
        z[live], u_raw[live], xbar[live] = z_p[live], u.reshape(Bn, T, 2)[live], xbar_p[live]
        X[live] = (qp["xfree"] + np.einsum("btik,bk->bti", qp["M"], u))[live]
        status[live] = st_p[live]
        iters[live] += it_p[live]
        ok = live & (st_p == STATUS_CONVERGED)
        U[ok] = u.reshape(Bn, T, 2)[ok]
        live = ok
    u0 = np.where(live[:, None], U[:, 0], 0.0)
    return dict(u0=u0, U=U, X=X, status=status, iters=iters, target_index=target, z=z, u_raw=u_raw, xref=xref, xbar=xbar)


# ---------------------------------------------------------------------------------------------------------------
# Independent evaluator: the cvxpy statements of _linear_mpc_control as plain loops (no matrices shared with build_qp)
# ---------------------------------------------------------------------------------------------------------------
def simulate_linear(x0, u, xbar, dt, wheelbase=WHEELBASE):
    T = u.shape[0]
    x = np.zeros((T + 1, 4))
    x[0] = x0
    for t in range(T):
        v_bar, yaw_bar = xbar[t, 2], xbar[t, 3]
        A = np.eye(4)
        A[0, 2] = dt * math.cos(yaw_bar)
        A[0, 3] = -dt * v_bar * math.sin(yaw_bar)
        A[1, 2] = dt * math.sin(yaw_bar)
        A[1, 3] = dt * v_bar * math.cos(yaw_bar)
        Bm = np.zeros((4, 2))
        Bm[2, 0] = dt
        Bm[3, 1] = dt * v_bar / wheelbase
        x[t + 1] = A @ x[t] + Bm @ u[t]
    return x


def objective_loops(u, x0, xref, xbar, dt):
    T = u.shape[0]
    x = simulate_linear(x0, u, xbar, dt)
    cost = 0.0
    for t in range(T):
        cost += R_DIAG[0] * u[t, 0] ** 2 + R_DIAG[1] * u[t, 1] ** 2
        if t < T - 1:
            cost += RD_DIAG[0] * (u[t + 1, 0] - u[t, 0]) ** 2 + RD_DIAG[1] * (u[t + 1, 1] - u[t, 1]) ** 2
        cost += Q_V * (x[t, 2] - xref[t, 2]) ** 2 + Q_YAW * (x[t, 3] - xref[t, 3]) ** 2
    for i in range(4):
        cost += T * QF_DIAG[i] * (x[T, i] - xref[T, i]) ** 2
    return cost


def constraint_loops(u, x0, xbar, dt):
    """all inequality constraints as a vector that must be >= 0"""
    T = u.shape[0]
    x = simulate_linear(x0, u, xbar, dt)
    out = []
    for t in range(T):
        out += [MAX_ACCEL - u[t, 0], u[t, 0] - MAX_DECEL, MAX_STEER - u[t, 1], MAX_STEER + u[t, 1]]
        if t < T - 1:
            d = u[t + 1, 1] - u[t, 1]
            out += [MAX_DSTEER * dt - d, MAX_DSTEER * dt + d]
    for t in range(T + 1):
        out += [x[t, 2], MAX_SPEED - x[t, 2]]
    return np.array(out)
