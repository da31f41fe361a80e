"""TEST INFRASTRUCTURE - dense full-space restatement of IPOPT's algorithm for the reference NLP.

Only tests/, tests/golden/make_*.py and tools/ may import this; never the product path.

The reference solves its NLP with `casadi.nlpsol('solver', 'ipopt', ...)`, options `max_iter 1000, tol 1e-6`, all else
default (reference agents/pure_mpc.py:285-300).  casadi==3.6.6 (requirements.txt:4, bundling IPOPT 3.14 + MUMPS) is not
installed in this image, so this module restates the published algorithm - A. Waechter, L. T. Biegler, "On the
implementation of an interior-point filter line-search algorithm for large-scale nonlinear programming", Math. Prog.
106 (2006), with the default option values of IPOPT 3.14 - on the *identical* problem in the *identical* full-space form
(124 variables, 84 equalities at N = 20, cold start of pure_mpc.py:240-246), with dense numpy linear algebra:

  * bounds relaxed by bound_relax_factor 1e-8, start pushed inside by bound_push = bound_frac = 1e-2 (sec. 3.6);
  * gradient-based objective scaling, nlp_scaling_max_gradient 100 (sec. 3.8); the constraint rows need none here;
  * bound multipliers 1, equality multipliers by least squares, dropped above constr_mult_init_max 1e3 (sec. 3.6);
  * monotone barrier update (mu_init 0.1, kappa_eps 10, kappa_mu 0.2, theta_mu 1.5), error E_mu with s_d, s_c (sec. 2.1);
  * primal-dual step from the augmented system with the inertia-correction heuristic IC (sec. 3.1);
  * fraction-to-the-boundary rule, filter line search with switching and Armijo conditions, second-order correction
    (max_soc 4, kappa_soc 0.99), filter reset at barrier updates (sec. 2.3-2.4, 3.2); alpha_for_y = primal;
  * multiplier safeguard kappa_Sigma 1e10 (eq. 16).

Not restated: the feasibility restoration phase (sec. 3.3: when the step falls below alpha_min this solver stops with
status "restoration"), the watchdog, the acceptable-level termination and the tiny-step logic.  It shares no solver code
with oracle/mpc_oracle.c or the HIP kernel (those factorise stage by stage; this one factorises the dense 208x208 KKT
matrix), only the NLP functions of oracle/nlp_batch.py, which tests check against finite differences.  Its role: the
independent solver behind tests/golden/independent_solutions.npz.
"""
from __future__ import annotations

import numpy as np

import nlp_batch as nb

STATUS = {0: "converged", 1: "max_iter", 2: "inertia correction failed", 5: "restoration"}


def _inertia_ok(K, n, m):
    """Inertia of the symmetric matrix K from its Bunch-Kaufman factorisation (Sylvester's law of inertia)."""
    from scipy.linalg import ldl
    _, D, _ = ldl(K, lower=True, check_finite=False)
    d0 = np.diag(D)
    d1 = np.diag(D, -1)
    npos = nneg = nzero = 0
    i = 0
    N = D.shape[0]
    while i < N:
        if i + 1 < N and d1[i] != 0.0:      # 2x2 pivot: one positive and one negative eigenvalue (det < 0) or same sign
            a, b, c = d0[i], d1[i], d0[i + 1]
            det = a * c - b * b
            tr = a + c
            if det < 0:
                npos += 1
                nneg += 1
            elif det > 0:
                if tr > 0:
                    npos += 2
                else:
                    nneg += 2
            else:
                nzero += 1
                if tr > 0:
                    npos += 1
                elif tr < 0:
                    nneg += 1
                else:
                    nzero += 1
            i += 2
        else:
            if d0[i] > 0:
                npos += 1
            elif d0[i] < 0:
                nneg += 1
            else:
                nzero += 1
            i += 1
    return npos == n and nneg == m, nzero > 0


def solve(p: nb.Batch, tol=1e-6, max_iter=1000, mu_init=0.1, trace=False, sf_min=1e-8, z_init=None):
    """Solve the single instance `p` (p.B == 1).  Returns dict(X, U, lam, zL, zU, status, iters, kkt, n_ic, n_soc)."""
    assert p.B == 1
    N = p.N
    n, m = 6 * N + 4, 4 * (N + 1)
    lo0, hi0 = nb.bounds_vec(N)
    lo = lo0 - 1e-8 * np.maximum(1.0, np.abs(lo0))
    hi = hi0 + 1e-8 * np.maximum(1.0, np.abs(hi0))

    def fun(z):
        X, U = nb.unpack(N, z[None])
        return float(nb.cost(p, X, U)[0])

    def grad(z):
        X, U = nb.unpack(N, z[None])
        gX, gU = nb.cost_grad(p, X, U)
        return nb.pack(gX, gU)[0]

    def con(z):
        X, U = nb.unpack(N, z[None])
        return nb.constraints(p, X, U)[0].ravel()

    def jac(z):
        X, U = nb.unpack(N, z[None])
        return nb.jac_dense(p, X, U)[0]

    def hess(z, lam, sf):
        X, U = nb.unpack(N, z[None])
        return nb.hess_dense(p, X, U, lam.reshape(1, N + 1, 4), sf)[0]

    # ---- starting point (pure_mpc.py:240-246) pushed into the interior
    if z_init is None:
        z = nb.pack(np.tile(p.state[:, None, :], (1, N + 1, 1)), np.zeros((1, N, 2)))[0]
    else:
        z = np.array(z_init, dtype=np.float64)
    pL = np.minimum(1e-2 * np.maximum(1.0, np.abs(lo)), 1e-2 * (hi - lo))
    pU = np.minimum(1e-2 * np.maximum(1.0, np.abs(hi)), 1e-2 * (hi - lo))
    z = np.minimum(np.maximum(z, lo + pL), hi - pU)
    sf = min(1.0, max(sf_min, 100.0 / max(1e-300, float(np.max(np.abs(grad(z)))))))
    zL = np.ones(n)
    zU = np.ones(n)
    J = jac(z)
    g = sf * grad(z)
    KK = np.block([[np.eye(n), J.T], [J, np.zeros((m, m))]])
    sol = np.linalg.solve(KK, -np.concatenate([g - zL + zU, np.zeros(m)]))
    lam = sol[n:]
    if np.max(np.abs(lam)) > 1e3:
        lam = np.zeros(m)

    mu = mu_init
    tau = max(0.99, 1.0 - mu)
    kap_eps, kap_mu, th_mu = 10.0, 0.2, 1.5
    gam_th, gam_phi, eta_phi, delta_sw, s_th, s_phi, gam_alpha = 1e-5, 1e-8, 1e-8, 1.0, 1.1, 2.3, 0.05
    KSIG = 1e10
    smax = 100.0
    dw_last = 0.0
    theta0 = float(np.sum(np.abs(con(z))))
    th_max, th_min = 1e4 * max(1.0, theta0), 1e-4 * max(1.0, theta0)
    filt = []      # list of (theta, phi) corners; th_max handled separately
    n_ic = n_soc = 0
    status, it, E0 = 1, 0, np.inf

    def barrier(zz, mu_):
        return sf * fun(zz) - mu_ * (np.sum(np.log(zz - lo)) + np.sum(np.log(hi - zz)))

    def in_filter(th, ph):
        return th >= th_max or any(th >= t and ph >= f for (t, f) in filt)

    for it in range(max_iter + 1):
        c = con(z)
        J = jac(z)
        g = sf * grad(z)
        sL, sU = z - lo, hi - z
        rd = g + J.T @ lam - zL + zU
        s_d = max(smax, (np.sum(np.abs(lam)) + np.sum(zL) + np.sum(zU)) / (m + 2 * n)) / smax
        s_c = max(smax, (np.sum(zL) + np.sum(zU)) / (2 * n)) / smax

        def E(mu_):
            return max(np.max(np.abs(rd)) / s_d, np.max(np.abs(c)),
                       max(np.max(np.abs(sL * zL - mu_)), np.max(np.abs(sU * zU - mu_))) / s_c)
        E0 = E(0.0)
        if E0 <= tol:
            status = 0
            break
        if it == max_iter:
            break
        changed = False
        while E(mu) <= kap_eps * mu and mu > tol / 10.0:
            mu = max(tol / 10.0, min(kap_mu * mu, mu ** th_mu))
            tau = max(0.99, 1.0 - mu)
            changed = True
        if changed:
            filt = []

        # ---- search direction, inertia correction (algorithm IC)
        W = hess(z, lam, sf)
        Sig = zL / sL + zU / sU
        gphi = g - mu / sL + mu / sU
        rhs = -np.concatenate([gphi + J.T @ lam, c])
        dw, dc = 0.0, 0.0
        d = None
        for attempt in range(200):
            K = np.block([[W + np.diag(Sig + dw), J.T], [J, -dc * np.eye(m)]])
            ok, singular = _inertia_ok(K, n, m)
            if ok:
                d = np.linalg.solve(K, rhs)
                break
            n_ic += 1
            if singular and dc == 0.0:
                dc = 1e-8 * mu ** 0.25
            if dw == 0.0:
                dw = 1e-4 if dw_last == 0.0 else max(1e-20, dw_last / 3.0)
            else:
                dw = dw * (100.0 if dw_last == 0.0 else 8.0)
            if dw > 1e40:
                break
        if d is None:
            status = 2
            break
        if dw > 0.0:
            dw_last = dw
        dx, dlam = d[:n], d[n:]
        dzL = mu / sL - zL - zL / sL * dx
        dzU = mu / sU - zU + zU / sU * dx

        def max_step(v, dv):
            neg = dv < 0
            return min(1.0, float(np.min(-tau * v[neg] / dv[neg]))) if np.any(neg) else 1.0
        a_max = min(max_step(sL, dx), max_step(sU, -dx))
        a_z = min(max_step(zL, dzL), max_step(zU, dzU))

        # ---- filter line search
        theta = float(np.sum(np.abs(c)))
        phi = barrier(z, mu)
        dphi = float(gphi @ dx)
        if dphi < 0:
            a_min = min(gam_th, gam_phi * theta / (-dphi))
            if theta <= th_min:
                a_min = min(a_min, delta_sw * theta ** s_th / (-dphi) ** s_phi)
            a_min *= gam_alpha
        else:
            a_min = gam_alpha * gam_th
        alpha = a_max
        accepted = False
        first = True
        dx_used = dx
        while alpha >= a_min:
            zt = z + alpha * dx
            th_t = float(np.sum(np.abs(con(zt))))
            ph_t = barrier(zt, mu)
            switching = dphi < 0 and alpha * (-dphi) ** s_phi > delta_sw * theta ** s_th

            def acceptable(th_t, ph_t):
                if in_filter(th_t, ph_t):
                    return False
                if theta <= th_min and switching:
                    return ph_t <= phi + eta_phi * alpha * dphi
                return th_t <= (1 - gam_th) * theta or ph_t <= phi - gam_phi * theta
            if acceptable(th_t, ph_t):
                accepted = True
                break
            # ---- second-order correction (only at the first trial and only if infeasibility did not decrease)
            if first and th_t >= theta:
                c_soc = alpha * c + con(zt)
                th_old = theta
                for _ in range(4):
                    n_soc += 1
                    ds = np.linalg.solve(K, -np.concatenate([gphi + J.T @ lam, c_soc]))
                    dxs = ds[:n]
                    a_s = min(max_step(sL, dxs), max_step(sU, -dxs))
                    zs = z + a_s * dxs
                    cs = con(zs)
                    th_s = float(np.sum(np.abs(cs)))
                    ph_s = barrier(zs, mu)
                    if acceptable(th_s, ph_s):
                        accepted = True
                        zt, th_t, ph_t, dx_used, alpha = zs, th_s, ph_s, dxs, a_s
                        dlam = ds[n:]
                        break
                    if th_s > 0.99 * th_old:
                        break
                    th_old = th_s
                    c_soc = a_s * c_soc + cs
                if accepted:
                    break
            first = False
            alpha *= 0.5
        if trace:
            print(f"it {it:3d} mu {mu:.1e} E0 {E0:.3e} th {theta:.3e} phi {phi:.6e} dw {dw:.1e} a_max {a_max:.3e} "
                  f"alpha {alpha:.3e} a_z {a_z:.3e} acc {accepted}")
        if not accepted:
            status = 5
            break
        if not (theta <= th_min and switching and ph_t <= phi + eta_phi * alpha * dphi):
            filt.append(((1 - gam_th) * theta, phi - gam_phi * theta))
        z = zt
        lam = lam + alpha * dlam
        zL = zL + a_z * dzL
        zU = zU + a_z * dzU
        sL, sU = z - lo, hi - z
        zL = np.maximum(np.minimum(zL, KSIG * mu / sL), mu / (KSIG * sL))
        zU = np.maximum(np.minimum(zU, KSIG * mu / sU), mu / (KSIG * sU))

    X, U = nb.unpack(N, z[None])
    return dict(X=X[0], U=U[0], lam=lam.reshape(N + 1, 4) / sf, zL=zL / sf, zU=zU / sf, status=status, iters=it,
                kkt=E0, n_ic=n_ic, n_soc=n_soc, sf=sf)
