"""TEST INFRASTRUCTURE - first-order optimality certificates for whole batches of candidate solutions.

Only tests/, __graft_entry__.smoke() and bench.py's parity side object may import this; never the product path.

Independent of every solver in this repository: given only the primal point (X, U) an engine returned, it decides
whether multipliers EXIST that make it a KKT point of the reference NLP (agents/pure_mpc.py:128-280, restated in
oracle/nlp_batch.py), to stated tolerances.  Nothing a solver computed besides (X, U) is used.

How: the state rows of the stationarity system determine the equality multipliers by the adjoint recursion
    lam_k = (I + dt A_k)' lam_{k+1} - grad_{x_k} f + zL_k - zU_k                      (L = f + lam'c - zL'(x-lo) - zU'(hi-x))
so the control rows  r_k = grad_{u_k} f - dt B_k' lam_{k+1} - zL^u_k + zU^u_k  are affine in the bound multipliers,
r = r0 + G z.  A certificate is a z with 0 <= z_i <= eps_c * scale / slack_i  (non-negative, complementary to
relative eps_c) that makes |r|_inf small: a box-constrained least-squares problem with 2N = 40 rows per instance
(scipy lsq_linear / BVLS), over the bounds whose slack is small enough for their multiplier to matter.

With the collision cost on, the objective is discontinuous at d = 1 (archive/pure_mpc.py:189-196: it jumps UP by
900 w_distance / d^2 when a vehicle comes nearer than 1 m).  A point with a vehicle exactly at d = 1 is a local minimiser
of that function iff it is a KKT point of the outer branch with the constraint |p_k - o_jk|^2 >= 1 added (moving
inwards only increases the cost): pairs with | |p-o|^2 - 1 | <= wall_tol enter the certificate as such constraints, and
`n_wall` reports how many a point needed.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import lsq_linear, nnls

import nlp_batch as nb


def certify(p: nb.Batch, X, U, eps_c=1e-6, wall_tol=1e-6, relax=1e-8, slack_max=10.0, chunk=512, sf=None):
    """Certificates for all instances of `p` at (X[B,N+1,4], U[B,N,2]).

    eps_c: complementarity allowed to the multipliers, z_i * slack_i <= eps_c * scale; a number or one value per instance
    (`objective_scale` below gives the per-instance allowance that IPOPT's own criterion implies).  The default 1e-6 is what a
    solver stopping on IPOPT's scaled error at tol 1e-8 guarantees here: the objective is scaled by up to 1e-2
    (nlp_scaling_max_gradient: 100 / |grad f(start)|_inf), so z s <= 1e-8 in scaled units is 1e-6 in the units of
    `scale` = |grad f(solution)|_inf when the gradient has fallen to the order of 1 (the reference's own tol 1e-6 would
    be 1e-4 in these units).  slack_max: bounds with more slack get no multiplier candidate.

    Returns dict of arrays [B]:
      stationarity   |r|_inf / scale of the best multipliers, scale = max(1, |grad f|_inf)  (relative, like IPOPT's s_d)
      feasibility    |c|_inf  (initial condition + dynamics defects)
      bound_violation   beyond the bounds relaxed by `relax` (IPOPT's bound_relax_factor)
      n_active       bounds / walls that received a multiplier candidate;  n_wall of them walls
      scale
      s_c            (with sf) IPOPT's complementarity scaling of the fitted multipliers, see below
      stationarity_ipopt, s_d   (with sf) the dual infeasibility in IPOPT's own units: sf |r|_inf / s_d, see the end of this function
      residual, lamsum, zsum    |r|_inf unscaled, |lambda|_1 and |z|_1 of the fitted multipliers

    sf (optional, one value per instance: `objective_scale`): IPOPT's criterion is complementarity <= tol * s_c with
    s_c = max(s_max, |z|_1 / n) / s_max, s_max = 100 (Waechter & Biegler 2006, eq. (6)) in the units of the SCALED objective -
    an instance whose bound multipliers average more than 100 (a handful per batch of 4096: long saturated stretches) is
    allowed proportionally more.  With sf the fit is repeated with eps_c * s_c where the fitted multipliers give s_c > 1
    (their sum in scaled units is sf * sum z); without it s_c = 1 throughout, the stricter reading.
    """
    B, N = p.B, p.N
    eps_c = np.broadcast_to(np.asarray(eps_c, dtype=np.float64), (B,))
    out = {k: np.zeros(B) for k in ("stationarity", "feasibility", "bound_violation", "scale")}
    out["n_active"] = np.zeros(B, dtype=np.int64)
    out["n_wall"] = np.zeros(B, dtype=np.int64)
    out["zsum"] = np.zeros(B)
    out["lamsum"] = np.zeros(B)
    out["residual"] = np.zeros(B)
    for s in range(0, B, chunk):
        sel = np.arange(s, min(B, s + chunk))
        r = _certify_chunk(p.take(sel), X[sel], U[sel], eps_c[sel], wall_tol, relax, slack_max)
        for k in out:
            out[k][sel] = r[k]
    out["s_c"] = np.ones(B)
    if sf is not None:
        sfv = np.broadcast_to(np.asarray(sf, dtype=np.float64), (B,))
        nvar = 6 * N
        for _ in range(3):                                        # the allowance and the multipliers it admits, to a fixed point
            s_c = np.maximum(100.0, sfv * out["zsum"] / nvar) / 100.0
            redo = np.nonzero(s_c > out["s_c"] * (1.0 + 1e-9))[0]
            if redo.size == 0:
                break
            out["s_c"][redo] = s_c[redo]
            r = _certify_chunk(p.take(redo), X[redo], U[redo], eps_c[redo] * s_c[redo], wall_tol, relax, slack_max)
            better = r["stationarity"] <= out["stationarity"][redo]
            for k in ("stationarity", "n_active", "n_wall", "zsum", "lamsum", "residual"):
                out[k][redo] = np.where(better, r[k], out[k][redo])
        # IPOPT's own optimality measure of the dual infeasibility (Waechter & Biegler 2006, eq. (5), (6)): the residual of the
        # SCALED problem divided by s_d = max(s_max, (|lambda|_1 + |z|_1) / (m + n)) / s_max, s_max = 100, with n = 6 N + 4
        # variables and m = 4 N + 4 equality rows of the reference's transcription (agents/pure_mpc.py:249-264) and the scaled
        # multipliers sf * lambda, sf * z.  This is the number IPOPT compares with `tol`; `stationarity` above is the same
        # residual relative to max(1, |grad f|_inf) instead, the stricter reading wherever the multipliers are large.
        out["s_d"] = np.maximum(100.0, sfv * (out["lamsum"] + out["zsum"]) / (10 * N + 8)) / 100.0
        out["stationarity_ipopt"] = sfv * out["residual"] / out["s_d"]
    return out


def objective_scale(p: nb.Batch):
    """IPOPT's gradient-based objective scaling (nlp_scaling_max_gradient 100) as the engines apply it, from the NLP data
    alone: sf = 100 / clamp(|grad f|_inf along the cold-start rollout, 100, 1e4), in [0.01, 1].  A solver that stops on
    IPOPT's scaled error at `tol` guarantees z s <= tol in units of the SCALED objective, i.e. tol / sf in the units of
    this module: `certify(..., eps_c=tol / objective_scale(p))` is that criterion instance by instance."""
    B, N, dt = p.B, p.N, p.dt
    U0 = np.zeros((B, N, 2))
    slow = p.state[:, 3] < 0.01                       # a standing vehicle starts with a_0 > 0 (strict interior of v >= 0)
    U0[slow, 0, 0] = (0.01 - p.state[slow, 3]) / dt
    X0 = np.zeros((B, N + 1, 4))
    X0[:, 0] = p.state
    for k in range(N):
        f, _ = dyn(X0[:, k:k + 1], U0[:, k:k + 1])
        X0[:, k + 1] = X0[:, k] + dt * f[:, 0]
    gX, gU = nb.cost_grad(p, X0, U0)
    gmax = np.maximum(np.abs(gX[:, 1:N]).max(axis=(1, 2)), np.abs(gU[:, 0, 0]))
    return 100.0 / np.clip(gmax, 100.0, 1e4)


def dyn(X, U):
    return nb.dyn(X, U)


def _certify_chunk(p, X, U, eps_c, wall_tol, relax, slack_max):
    B, N, dt = p.B, p.N, p.dt
    gX, gU = nb.cost_grad(p, X, U)
    _, d = nb.dyn(X, U)
    A, Bm = nb.dyn_jac(d)
    Phi = np.eye(4)[None, None] + dt * A                       # [B, N, 4, 4]   d x_{k+1} / d x_k
    scale = np.maximum(1.0, np.maximum(np.abs(gX).max(axis=(1, 2)), np.abs(gU).max(axis=(1, 2))))
    # ---- base multipliers (all bound multipliers zero) and control residual r0
    lam = np.zeros((B, N + 2, 4))
    lam[:, N] = -gX[:, N]
    for k in range(N - 1, 0, -1):
        lam[:, k] = np.einsum("bji,bj->bi", Phi[:, k], lam[:, k + 1]) - gX[:, k]
    r0 = gU - dt * np.einsum("bkji,bkj->bki", Bm, lam[:, 1:N + 1])          # [B, N, 2]
    # ---- sensitivity of r_k to a unit shift of lam_j along each state direction: Sens[b, j, k] (2 x 4), k + 1 <= j
    Sens = np.zeros((B, N + 1, N, 2, 4))
    for j in range(1, N + 1):
        Mj = np.tile(np.eye(4), (B, 1, 1))                                    # d lam_{m} / d lam_j, m = j
        for k in range(j - 1, -1, -1):                                        # row k uses lam_{k+1}
            Sens[:, j, k] = -dt * np.einsum("bji,bjl->bil", Bm[:, k], Mj)
            if k >= 1:
                Mj = np.einsum("bji,bjl->bil", Phi[:, k], Mj)                # lam_k shift = Phi_k' (lam_{k+1} shift)
    # ---- slacks
    xlo, xhi = nb.X_LO - relax * np.maximum(1, np.abs(nb.X_LO)), nb.X_HI + relax * np.maximum(1, np.abs(nb.X_HI))
    ulo, uhi = nb.U_LO - relax * np.maximum(1, np.abs(nb.U_LO)), nb.U_HI + relax * np.maximum(1, np.abs(nb.U_HI))
    sxl, sxu = X - xlo, xhi - X
    sul, suu = U - ulo, uhi - U
    viol = np.maximum(0.0, -np.minimum(np.minimum(sxl.min(axis=(1, 2)), sxu.min(axis=(1, 2))),
                                       np.minimum(sul.min(axis=(1, 2)), suu.min(axis=(1, 2)))))
    feas = np.abs(nb.constraints(p, X, U)).max(axis=(1, 2))
    walls = None
    if p.cc and p.others is not None and p.others.shape[1]:
        dp = X[:, :N, None, :2] - p.other_pos()                               # [B, N, V, 2]
        g = np.sum(dp * dp, axis=-1) - 1.0
        walls = (np.abs(g) <= wall_tol)
        walls[:, 0] = False                                                   # X_0 is pinned
    stat = np.zeros(B)
    nact = np.zeros(B, dtype=np.int64)
    nwall = np.zeros(B, dtype=np.int64)
    zsum = np.zeros(B)
    lamsum = np.zeros(B)
    resabs = np.zeros(B)
    for b in range(B):
        cols, ub, shifts = [], [], []      # shifts: (node j, direction in state space) of the multiplier's push on lam_j
        zmax = eps_c[b] * scale[b]
        thr = slack_max      # a bound with more slack could hold a multiplier of at most eps_c / slack_max, relative

        def add(col, slack, shift=None):
            cols.append(col.ravel())
            ub.append(zmax / max(slack, 1e-300))
            shifts.append(shift)
        for k, i in zip(*np.nonzero(sul[b] < thr)):
            c = np.zeros((N, 2)); c[k, i] = -1.0
            add(c, sul[b, k, i])
        for k, i in zip(*np.nonzero(suu[b] < thr)):
            c = np.zeros((N, 2)); c[k, i] = 1.0
            add(c, suu[b, k, i])
        for j, i in zip(*np.nonzero(sxl[b, 1:] < thr)):
            add(Sens[b, j + 1, :, :, i], sxl[b, j + 1, i], (j + 1, np.eye(4)[i]))      # zL shifts lam_j by +e_i
        for j, i in zip(*np.nonzero(sxu[b, 1:] < thr)):
            add(-Sens[b, j + 1, :, :, i], sxu[b, j + 1, i], (j + 1, -np.eye(4)[i]))
        if walls is not None:
            for j, v in zip(*np.nonzero(walls[b])):
                col = Sens[b, j, :, :, 0] * (2.0 * dp[b, j, v, 0]) + Sens[b, j, :, :, 1] * (2.0 * dp[b, j, v, 1])
                cols.append(col.ravel())
                ub.append(np.inf)                                             # an active wall: multiplier free in sign +
                shifts.append((j, np.array([2.0 * dp[b, j, v, 0], 2.0 * dp[b, j, v, 1], 0.0, 0.0])))
                nwall[b] += 1
        rb = r0[b].ravel()
        zfit = np.zeros(0)
        if cols:
            # scaled unknowns: a boxed multiplier as a fraction t in [0, 1] of its box, a free one (box beyond anything
            # the residual could ask for) in units of `scale` - BVLS is only reliable on well-scaled columns
            G = np.stack(cols, axis=1)
            ubv = np.array(ub)
            cn = np.maximum(np.abs(G).max(axis=0), 1e-200)
            free = ubv * cn > 1e3 * scale[b]
            unit = np.where(free, scale[b] / cn, np.minimum(ubv, 1e300))
            hi = np.where(free, np.inf, 1.0)
            Gs = G * unit[None, :] / scale[b]
            sol = lsq_linear(Gs, -rb / scale[b], bounds=(np.zeros(len(ub)), hi), method="bvls", tol=1e-15, max_iter=1000)
            xs = sol.x
            if np.abs(Gs @ xs + rb / scale[b]).max() > 1e-10:
                # BVLS can stop early on these badly conditioned 40 x ~120 systems (cond ~ 1e11) with a set of variables parked at
                # their upper bounds (round 5: instance 2791 of config 3, seed 0, ended at 1.4e-8 where the optimum is 2e-16 - the
                # reason two gates had been loosened).  Two other exact methods for the same convex problem, best of the three:
                # non-negative least squares (Lawson-Hanson; optimal for the boxed problem whenever its solution respects the
                # boxes, which it does unless a complementarity box really binds) and the trust-region reflective method.
                cand = [xs]
                xn, _ = nnls(Gs, -rb / scale[b])
                if np.all(xn <= hi):
                    cand.append(xn)
                trf = lsq_linear(Gs, -rb / scale[b], bounds=(np.zeros(len(ub)), hi), method="trf", tol=1e-15, max_iter=5000,
                                 lsq_solver="exact")
                cand.append(np.clip(trf.x, 0.0, hi))
                xs = min(cand, key=lambda x_: np.abs(Gs @ x_ + rb / scale[b]).max())
            res = rb + G @ (xs * unit)
            zfit = xs * unit
            zsum[b] = float(np.sum(zfit[np.isfinite(ubv)]))      # bound multipliers (the walls' are not part of s_c's z)
        else:
            res = rb
        stat[b] = np.abs(res).max() / scale[b]
        resabs[b] = np.abs(res).max()
        nact[b] = len(cols)
        # |lambda|_1 of the equality multipliers that go with the fitted bound multipliers: the adjoint recursion again, with
        # every state-bound / wall multiplier's push on its node (lam_0 = the multiplier of the reference's X[0] = state row)
        push = np.zeros((N + 1, 4))
        for zv, sh in zip(zfit, shifts):
            if sh is not None:
                push[sh[0]] += zv * sh[1]
        lb = np.zeros((N + 2, 4))
        lb[N] = -gX[b, N] + push[N]
        for k in range(N - 1, -1, -1):
            lb[k] = Phi[b, k].T @ lb[k + 1] - gX[b, k] + push[k]
        lamsum[b] = np.abs(lb[:N + 1]).sum()
    return dict(stationarity=stat, feasibility=feas, bound_violation=viol, scale=scale, n_active=nact, n_wall=nwall, zsum=zsum,
                lamsum=lamsum, residual=resabs)
