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
  * multiplier safeguard kappa_Sigma 1e10 (eq. 16);
  * (round 4) the feasibility RESTORATION PHASE of sec. 3.3, entered when the line search falls below alpha_min or the
    inertia correction fails: the current point's (theta, phi) pair joins the filter; the same interior-point method is run
    on   min  rho |p + n|_1 + zeta / 2 |D_R (x - x_R)|^2   s.t.  c(x) - p + n = 0,  p, n >= 0,  x_L <= x <= x_U
    with rho = 1000, zeta = sqrt(mu), D_R = diag(min(1, 1 / |x_R|)), started at x_R with mu_bar = max(mu, |c(x_R)|_inf), p
    and n from eq. (30), lambda = 0, z_p = mu_bar / p, z_n = mu_bar / n and the x-bound multipliers capped at rho; it
    ends at the first iterate that is acceptable to the ORIGINAL filter and has theta <= kappa_resto theta(x_R)
    (kappa_resto 0.9).  Back in the regular method the bound multipliers take a Newton step for complementarity with the
    restoration's total x-step (all reset to 1 if one exceeds bound_mult_reset_threshold 1000), the equality multipliers are
    zero (constr_mult_reset_threshold 0), mu is unchanged.  A restoration that cannot reduce the infeasibility ends the
    solve (status 5, IPOPT's "Restoration Failed" / "Converged to a point of local infeasibility").

  * (round 5) the ACCEPTABLE-LEVEL TERMINATION (IpOptErrorConvCheck: acceptable_tol 1e-6, acceptable_iter 15,
    acceptable_dual_inf_tol 1e10, acceptable_constr_viol_tol 1e-2, acceptable_compl_inf_tol 1e-2, acceptable_obj_change_tol
    1e20 - all defaults, the reference sets none of them, agents/pure_mpc.py:291-296): the 15th consecutive iterate whose
    scaled error is at most acceptable_tol ends the solve with status 3 ("Solved To Acceptable Level", which casadi reports
    as success like "Solve Succeeded"); an iterate acceptable in that sense also turns a restoration failure or a tiny-step
    exit into status 3, as IPOPT's `STOP_AT_ACCEPTABLE_POINT` does.  It cannot fire when tol >= acceptable_tol (the
    reference's tol 1e-6: the regular test is met first);
  * (round 5) the TINY-STEP logic (IpBacktrackingLineSearch::DetectTinyStep, tiny_step_tol 10 eps, tiny_step_y_tol 1e-2): a
    search direction with max_i |dx_i| / (1 + |x_i|) below 10 eps at an iterate with constraint violation below 1e-4 is
    taken without a line search (full fraction-to-the-boundary step); two such iterations in a row with |d lambda|_inf <
    tiny_step_y_tol raise the tiny-step flag, which forces the next barrier update (MonotoneMuUpdate) even if the
    barrier problem's error test is not met, and ends the solve with status 4 ("Search Direction Becomes Too Small",
    a failure exit of nlpsol unless the point is acceptable) when mu cannot be lowered any more.

Not restated: the watchdog (measured in round 5 on the engine's own globalisation, where it bought nothing: DESIGN.md
section 2.1).  It shares no solver code
with oracle/mpc_oracle.c or the HIP kernel (those factorise stage by stage; this one factorises the dense 208x208 KKT
matrix), only the NLP functions of oracle/nlp_batch.py, which tests check against finite differences and - since round 4 -
against the reference's own objective / constraint statements evaluated numerically (tests/test_reference_vectors.py).
Its role: the independent solver behind tests/golden/independent_solutions.npz and closed_loop_ipopt.npz.
"""
from __future__ import annotations

import numpy as np

import nlp_batch as nb

STATUS = {0: "converged", 1: "max_iter", 2: "inertia correction failed", 3: "solved to acceptable level",
          4: "search direction becomes too small (tiny step)", 5: "restoration failed / not restated",
          6: "restoration converged to a point of local infeasibility"}
ACCEPTABLE_TOL, ACCEPTABLE_ITER = 1e-6, 15
ACCEPTABLE_DUAL_INF, ACCEPTABLE_CONSTR_VIOL, ACCEPTABLE_COMPL_INF = 1e10, 1e-2, 1e-2
TINY_STEP_TOL, TINY_STEP_Y_TOL = 10.0 * np.finfo(np.float64).eps, 1e-2
RHO_RESTO, KAPPA_RESTO, BOUND_MULT_RESET = 1000.0, 0.9, 1000.0


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


class _OrigNLP:
    """The reference NLP in its full-space form, objective scaled by sf (sec. 3.8), bounds relaxed (sec. 3.5)."""

    def __init__(self, p: nb.Batch, sf: float):
        self.p, self.N, self.sf = p, p.N, sf
        self.n, self.m = 6 * p.N + 4, 4 * (p.N + 1)
        lo0, hi0 = nb.bounds_vec(p.N)
        self.lo = lo0 - 1e-8 * np.maximum(1.0, np.abs(lo0))
        self.hi = hi0 + 1e-8 * np.maximum(1.0, np.abs(hi0))

    def _xu(self, z):
        return nb.unpack(self.N, z[None])

    def fun(self, z):
        X, U = self._xu(z)
        return self.sf * float(nb.cost(self.p, X, U)[0])

    def grad(self, z):
        X, U = self._xu(z)
        gX, gU = nb.cost_grad(self.p, X, U)
        return self.sf * nb.pack(gX, gU)[0]

    def con(self, z):
        X, U = self._xu(z)
        return nb.constraints(self.p, X, U)[0].ravel()

    def jac(self, z):
        X, U = self._xu(z)
        return nb.jac_dense(self.p, X, U)[0]

    def hess(self, z, lam, obj=True):
        X, U = self._xu(z)
        return nb.hess_dense(self.p, X, U, lam.reshape(1, self.N + 1, 4), self.sf if obj else 0.0)[0]


class _RestoNLP:
    """Waechter & Biegler (2006) eq. (29): variables w = (x, p, n)."""

    def __init__(self, orig: _OrigNLP, xR, mu):
        self.o, self.xR = orig, xR.copy()
        self.nx, self.m = orig.n, orig.m
        self.n = orig.n + 2 * orig.m
        self.zeta = float(np.sqrt(mu))
        self.DR2 = (1.0 / np.maximum(1.0, np.abs(xR))) ** 2
        self.lo = np.concatenate([orig.lo, np.zeros(2 * orig.m)])
        self.hi = np.concatenate([orig.hi, np.full(2 * orig.m, np.inf)])

    def split(self, w):
        return w[:self.nx], w[self.nx:self.nx + self.m], w[self.nx + self.m:]

    def fun(self, w):
        x, pp, nn = self.split(w)
        return RHO_RESTO * float(np.sum(pp) + np.sum(nn)) + 0.5 * self.zeta * float(np.sum(self.DR2 * (x - self.xR) ** 2))

    def grad(self, w):
        x, _, _ = self.split(w)
        return np.concatenate([self.zeta * self.DR2 * (x - self.xR), np.full(2 * self.m, RHO_RESTO)])

    def con(self, w):
        x, pp, nn = self.split(w)
        return self.o.con(x) - pp + nn

    def jac(self, w):
        x, _, _ = self.split(w)
        return np.hstack([self.o.jac(x), -np.eye(self.m), np.eye(self.m)])

    def hess(self, w, lam, obj=True):
        x, _, _ = self.split(w)
        W = np.zeros((self.n, self.n))
        W[:self.nx, :self.nx] = self.o.hess(x, lam, obj=False) + np.diag(self.zeta * self.DR2)
        return W


def _ipm(P, z, zL, zU, lam, mu, tol, max_iter, trace=False, exit_test=None, allow_resto=True, counters=None, tag=""):
    """The interior-point filter line-search method on problem P from the interior point z.  exit_test(z_trial) -> True ends
    the run at that iterate (the restoration phase's return condition).  Returns dict(z, zL, zU, lam, mu, status, iters,
    kkt); status 7 = left through exit_test."""
    n, m = P.n, P.m
    lo, hi = P.lo, P.hi
    fl, fu = np.isfinite(lo), np.isfinite(hi)
    counters = counters if counters is not None else {}
    for k in ("n_ic", "n_soc", "n_resto", "resto_iters"):
        counters.setdefault(k, 0)
    tau = max(0.99, 1.0 - mu)
    kap_eps, kap_mu, th_mu = 10.0, 0.2, 1.5
    gam_th, gam_phi, eta_phi, delta_sw, s_th, s_phi, gam_alpha = 1e-5, 1e-8, 1e-8, 1.0, 1.1, 2.3, 0.05
    KSIG = 1e10
    smax = 100.0
    dw_last = 0.0
    theta0 = float(np.sum(np.abs(P.con(z))))
    th_max, th_min = 1e4 * max(1.0, theta0), 1e-4 * max(1.0, theta0)
    filt = []
    status, it, E0 = 1, 0, np.inf
    nb_cnt = int(fl.sum() + fu.sum())
    n_acceptable = 0                  # consecutive acceptable iterates (IpOptErrorConvCheck::acceptable_counter_)
    is_acceptable = False
    tiny_last = False                 # the previous iteration's direction was tiny
    tiny_flag = False                 # IpData().tiny_step_flag(): two tiny directions in a row
    orig = isinstance(P, _OrigNLP)    # both rules belong to the regular method, not to the restoration subproblem

    def slack(zz):
        return np.where(fl, zz - lo, 1.0), np.where(fu, hi - zz, 1.0)

    def barrier(zz, mu_):
        sL_, sU_ = slack(zz)
        return P.fun(zz) - mu_ * (np.sum(np.log(sL_[fl])) + np.sum(np.log(sU_[fu])))

    def in_filter(th, ph):
        return th >= th_max or any(th >= t and ph >= f for (t, f) in filt)

    def max_step(v, dv, mask):
        neg = mask & (dv < 0)
        return min(1.0, float(np.min(-tau * v[neg] / dv[neg]))) if np.any(neg) else 1.0

    while True:
        c = P.con(z)
        J = P.jac(z)
        g = P.grad(z)
        sL, sU = slack(z)
        zL = np.where(fl, zL, 0.0)
        zU = np.where(fu, zU, 0.0)
        rd = g + J.T @ lam - zL + zU
        s_d = max(smax, (np.sum(np.abs(lam)) + np.sum(zL) + np.sum(zU)) / (m + nb_cnt)) / smax
        s_c = max(smax, (np.sum(zL) + np.sum(zU)) / nb_cnt) / smax

        def E(mu_):
            comp = max(np.max(np.abs((sL * zL - mu_)[fl]), initial=0.0), np.max(np.abs((sU * zU - mu_)[fu]), initial=0.0))
            return max(np.max(np.abs(rd)) / s_d, np.max(np.abs(c)), comp / s_c)
        E0 = E(0.0)
        if E0 <= tol:
            status = 0
            break
        if orig:
            # acceptable level: the scaled error and the three unscaled parts (objective scaling sf <= 1 makes the scaled
            # dual infeasibility / complementarity the smaller ones; their bounds 1e10 / 1e-2 are met a fortiori by E0 <= 1e-6)
            comp0 = max(np.max((sL * zL)[fl], initial=0.0), np.max((sU * zU)[fu], initial=0.0))
            is_acceptable = (E0 <= ACCEPTABLE_TOL and np.max(np.abs(rd)) / P.sf <= ACCEPTABLE_DUAL_INF
                             and np.max(np.abs(c)) <= ACCEPTABLE_CONSTR_VIOL and comp0 / P.sf <= ACCEPTABLE_COMPL_INF)
            n_acceptable = n_acceptable + 1 if is_acceptable else 0
            if n_acceptable >= ACCEPTABLE_ITER:
                status = 3
                break
        if it >= max_iter:
            break
        changed = False
        force = tiny_flag                 # MonotoneMuUpdate: a tiny step asks for the next barrier problem regardless
        tiny_flag = False
        stop_tiny = False
        while (E(mu) <= kap_eps * mu or force) and mu > tol / 10.0:
            mu = max(tol / 10.0, min(kap_mu * mu, mu ** th_mu))
            tau = max(0.99, 1.0 - mu)
            changed = True
            force = False
        if force and not changed:
            stop_tiny = True              # "Problem solved to best possible numerical accuracy": mu cannot be lowered
        if stop_tiny:
            status = 3 if is_acceptable else 4
            break
        if changed:
            filt = []

        # ---- search direction, inertia correction (algorithm IC)
        W = P.hess(z, lam)
        Sig = np.where(fl, zL / sL, 0.0) + np.where(fu, zU / sU, 0.0)
        gphi = g - np.where(fl, mu / sL, 0.0) + np.where(fu, mu / sU, 0.0)
        rhs = -np.concatenate([gphi + J.T @ lam, c])
        dw, dc = 0.0, 0.0
        d = None
        K = None
        for attempt in range(200):
            K = np.block([[W + np.diag(Sig + dw), J.T], [J, -dc * np.eye(m)]])
            ok, singular = _inertia_ok(K, n, m)
            if ok:
                d = np.linalg.solve(K, rhs)
                break
            counters["n_ic"] += 1
            if singular and dc == 0.0:
                dc = 1e-8 * mu ** 0.25
            if dw == 0.0:
                dw = 1e-4 if dw_last == 0.0 else max(1e-20, dw_last / 3.0)
            else:
                dw = dw * (100.0 if dw_last == 0.0 else 8.0)
            if dw > 1e40:
                break
        theta = float(np.sum(np.abs(c)))
        phi = barrier(z, mu)
        accepted = False
        need_resto = d is None
        if d is not None:
            if dw > 0.0:
                dw_last = dw
            dx, dlam = d[:n], d[n:]
            dzL = np.where(fl, mu / sL - zL - zL / sL * dx, 0.0)
            dzU = np.where(fu, mu / sU - zU + zU / sU * dx, 0.0)
            a_max = min(max_step(sL, dx, fl), max_step(sU, -dx, fu))
            a_z = min(max_step(zL, dzL, fl), max_step(zU, dzU, fu))

            # ---- a tiny search direction is taken as it is (DetectTinyStep)
            tiny = bool(orig and np.max(np.abs(dx) / (1.0 + np.abs(z))) <= TINY_STEP_TOL and np.max(np.abs(c), initial=0.0) <= 1e-4)
            if tiny:
                if tiny_last and np.max(np.abs(dlam), initial=0.0) < TINY_STEP_Y_TOL:
                    tiny_flag = True
                tiny_last = True
                z = z + a_max * dx
                lam = lam + a_max * dlam
                zL = zL + a_z * dzL
                zU = zU + a_z * dzU
                sL, sU = slack(z)
                zL = np.where(fl, np.maximum(np.minimum(zL, KSIG * mu / sL), mu / (KSIG * sL)), 0.0)
                zU = np.where(fu, np.maximum(np.minimum(zU, KSIG * mu / sU), mu / (KSIG * sU)), 0.0)
                it += 1
                if trace:
                    print(f"{tag}it {it:3d} mu {mu:.1e} E0 {E0:.3e} tiny step taken without line search")
                continue
            tiny_last = False

            # ---- filter line search
            dphi = float(gphi @ dx)
            if dphi < 0:
                a_min = min(gam_th, gam_phi * theta / (-dphi))
                if theta <= th_min:
                    a_min = min(a_min, delta_sw * theta ** s_th / (-dphi) ** s_phi)
                a_min *= gam_alpha
            else:
                a_min = gam_alpha * gam_th
            alpha = a_max
            first = True
            switching = False
            zt = z
            th_t = ph_t = 0.0
            while alpha >= a_min:
                zt = z + alpha * dx
                th_t = float(np.sum(np.abs(P.con(zt))))
                ph_t = barrier(zt, mu)
                switching = dphi < 0 and alpha * (-dphi) ** s_phi > delta_sw * theta ** s_th

                def acceptable(th_t, ph_t):
                    if not np.isfinite(ph_t) or in_filter(th_t, ph_t):
                        return False
                    if theta <= th_min and switching:
                        return ph_t <= phi + eta_phi * alpha * dphi
                    return th_t <= (1 - gam_th) * theta or ph_t <= phi - gam_phi * theta
                if acceptable(th_t, ph_t):
                    accepted = True
                    break
                # ---- second-order correction (only at the first trial and only if infeasibility did not decrease)
                if first and th_t >= theta:
                    c_soc = alpha * c + P.con(zt)
                    th_old = theta
                    for _ in range(4):
                        counters["n_soc"] += 1
                        ds = np.linalg.solve(K, -np.concatenate([gphi + J.T @ lam, c_soc]))
                        dxs = ds[:n]
                        a_s = min(max_step(sL, dxs, fl), max_step(sU, -dxs, fu))
                        zs = z + a_s * dxs
                        cs = P.con(zs)
                        th_s = float(np.sum(np.abs(cs)))
                        ph_s = barrier(zs, mu)
                        if acceptable(th_s, ph_s):
                            accepted = True
                            zt, th_t, ph_t, alpha = zs, th_s, ph_s, a_s
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
                print(f"{tag}it {it:3d} mu {mu:.1e} E0 {E0:.3e} th {theta:.3e} phi {phi:.6e} dw {dw:.1e} a_max {a_max:.3e} "
                      f"alpha {alpha:.3e} a_z {a_z:.3e} acc {accepted}")
            need_resto = not accepted
        if need_resto:
            if not allow_resto or not isinstance(P, _OrigNLP):
                status = 5 if d is not None else 2
                break
            if is_acceptable:                # IPOPT: "Restoration phase is called at acceptable point" -> STOP_AT_ACCEPTABLE_POINT
                status = 3
                break
            # ---- feasibility restoration phase (sec. 3.3)
            filt.append(((1 - gam_th) * theta, phi - gam_phi * theta))
            if theta <= tol * 1e-2:          # nothing to restore: IPOPT's "Restoration phase is called at almost feasible point"
                status = 5
                break
            R = _RestoNLP(P, z, mu)
            mu_bar = max(mu, float(np.max(np.abs(c))))
            a_ = (mu_bar - RHO_RESTO * c) / (2.0 * RHO_RESTO)
            nn0 = a_ + np.sqrt(a_ * a_ + mu_bar * c / (2.0 * RHO_RESTO))
            nn0 = np.maximum(nn0, 1e-300)
            pp0 = np.maximum(c + nn0, 1e-300)
            w0 = np.concatenate([z, pp0, nn0])
            wL = np.concatenate([np.minimum(RHO_RESTO, zL), mu_bar / pp0, mu_bar / nn0])
            wU = np.concatenate([np.minimum(RHO_RESTO, zU), np.zeros(2 * m)])
            th_R, mu_now, filt_now, zR = theta, mu, list(filt), z

            def back(w):
                x = w[:n]
                th_x = float(np.sum(np.abs(P.con(x))))
                if th_x > KAPPA_RESTO * th_R:
                    return False
                ph_x = barrier(x, mu_now)
                if not np.isfinite(ph_x):
                    return False
                return not (th_x >= th_max or any(th_x >= t and ph_x >= f for (t, f) in filt_now))
            counters["n_resto"] += 1
            r = _ipm(R, w0, wL, wU, np.zeros(m), mu_bar, tol, max(0, max_iter - it), trace=trace, exit_test=back,
                     allow_resto=False, counters=counters, tag=tag + "  R ")
            it += r["iters"]
            counters["resto_iters"] += r["iters"]
            if r["status"] != 7:
                # the restoration problem was solved (or gave up) without reaching a point the filter accepts
                # (IPOPT: RESTORATION_FAILED / LOCALLY_INFEASIBLE / "restoration converged to a feasible point that is
                # unacceptable to the filter" - all of them failure exits of nlpsol; the iterate the caller gets, and the
                # reference then ACTS on (agents/pure_mpc.py:303-311), is the regular method's last one, x_R)
                status = 6 if r["status"] == 0 else (1 if r["status"] == 1 else 5)
                break
            x_new = r["z"][:n]
            dxr = x_new - zR
            sL, sU = slack(zR)
            dzL = np.where(fl, mu / sL - zL - zL / sL * dxr, 0.0)
            dzU = np.where(fu, mu / sU - zU + zU / sU * dxr, 0.0)
            a_z = min(max_step(zL, dzL, fl), max_step(zU, dzU, fu))
            zL, zU = zL + a_z * dzL, zU + a_z * dzU
            if max(np.max(zL), np.max(zU)) > BOUND_MULT_RESET:
                zL, zU = np.where(fl, 1.0, 0.0), np.where(fu, 1.0, 0.0)
            lam = np.zeros(m)
            z = x_new
            continue
        if not (theta <= th_min and switching and ph_t <= phi + eta_phi * alpha * dphi):
            filt.append(((1 - gam_th) * theta, phi - gam_phi * theta))
        z = zt
        lam = lam + alpha * dlam
        zL = zL + a_z * dzL
        zU = zU + a_z * dzU
        sL, sU = slack(z)
        zL = np.where(fl, np.maximum(np.minimum(zL, KSIG * mu / sL), mu / (KSIG * sL)), 0.0)
        zU = np.where(fu, np.maximum(np.minimum(zU, KSIG * mu / sU), mu / (KSIG * sU)), 0.0)
        it += 1
        if exit_test is not None and exit_test(z):
            status = 7
            break
    return dict(z=z, zL=zL, zU=zU, lam=lam, mu=mu, status=status, iters=it, kkt=E0)


def solve(p: nb.Batch, tol=1e-6, max_iter=1000, mu_init=0.1, trace=False, sf_min=1e-8, z_init=None, restoration=True):
    """Solve the single instance `p` (p.B == 1).  Returns dict(X, U, lam, zL, zU, status, iters, kkt, n_ic, n_soc, n_resto,
    resto_iters).  restoration=False stops where IPOPT would enter its restoration phase (status 5), as rounds 2-3 did."""
    assert p.B == 1
    N = p.N
    probe = _OrigNLP(p, 1.0)
    n, m, lo, hi = probe.n, probe.m, probe.lo, probe.hi
    # ---- starting point (pure_mpc.py:240-246) pushed into the interior
    if z_init is None:
        z = nb.pack(np.tile(p.state[:, None, :], (1, N + 1, 1)), np.zeros((1, N, 2)))[0]
    else:
        z = np.array(z_init, dtype=np.float64)
    pL = np.minimum(1e-2 * np.maximum(1.0, np.abs(lo)), 1e-2 * (hi - lo))
    pU = np.minimum(1e-2 * np.maximum(1.0, np.abs(hi)), 1e-2 * (hi - lo))
    z = np.minimum(np.maximum(z, lo + pL), hi - pU)
    sf = min(1.0, max(sf_min, 100.0 / max(1e-300, float(np.max(np.abs(probe.grad(z)))))))
    P = _OrigNLP(p, sf)
    zL = np.ones(n)
    zU = np.ones(n)
    J = P.jac(z)
    g = P.grad(z)
    KK = np.block([[np.eye(n), J.T], [J, np.zeros((m, m))]])
    sol = np.linalg.solve(KK, -np.concatenate([g - zL + zU, np.zeros(m)]))
    lam = sol[n:]
    if np.max(np.abs(lam)) > 1e3:
        lam = np.zeros(m)
    cnt = {}
    r = _ipm(P, z, zL, zU, lam, mu_init, tol, max_iter, trace=trace, allow_resto=restoration, counters=cnt)
    X, U = nb.unpack(N, r["z"][None])
    return dict(X=X[0], U=U[0], lam=r["lam"].reshape(N + 1, 4) / sf, zL=r["zL"] / sf, zU=r["zU"] / sf, status=r["status"],
                iters=r["iters"], kkt=r["kkt"], n_ic=cnt["n_ic"], n_soc=cnt["n_soc"], n_resto=cnt["n_resto"],
                resto_iters=cnt["resto_iters"], sf=sf)
