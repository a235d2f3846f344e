"""Laplace approximation over the random-effect coefficients.

In the reference, `random = "coeff_re"` (/root/reference/R/sde.R:522, 656-658) makes TMB integrate the
random effects out of the joint negative log-likelihood g(theta, u) by the Laplace approximation

    f(theta) = g(theta, u_hat) + 1/2 log det H_uu(theta, u_hat) - n_u/2 log(2 pi),   u_hat = argmin_u g(theta, u)

with an inner Newton solve and AD-exact derivatives.  TMB is not part of this engine; this module gives the
same objective on top of ANY joint (value, gradient) evaluator -- on the GPU that is `Engine.eval`, whose cost
(0.1-1 ms per evaluation) makes derivative-free outer layers affordable:

  * inner problem: Newton iterations on u, Hessian H_uu by central finite differences of the GPU gradient
    (2 n_u evaluations), warm-started at the previous u_hat;
  * outer gradient: central finite differences of f(theta) (documented approximation of TMB's implicit-function
    gradient; tolerance set by `fd_step`).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np


class LaplaceObjective:
    """fn / gr of the marginal negative log-likelihood over the OUTER parameters.

    joint(par_full) -> (value, grad_full)   joint penalised nllk and its gradient (full parameter vector)
    par_full        initial full vector (fixed entries stay at these values)
    idx_outer       indices optimised by the caller (fixed effects, log_sigma_obs, log_lambda)
    idx_random      indices integrated out (coeff_re)
    """

    def __init__(self, joint: Callable[[np.ndarray], Tuple[float, np.ndarray]], par_full: np.ndarray,
                 idx_outer: Sequence[int], idx_random: Sequence[int], fd_step: float = 1e-4, hess_step: float = 1e-4,
                 newton_tol: float = 1e-8, max_newton: int = 30):
        self.joint = joint
        self.par_full = np.array(par_full, dtype=np.float64)
        self.io = np.asarray(idx_outer, dtype=int)
        self.ir = np.asarray(idx_random, dtype=int)
        self.fd_step, self.hess_step, self.newton_tol, self.max_newton = fd_step, hess_step, newton_tol, max_newton
        self.u_hat = self.par_full[self.ir].copy()
        self.par = self.par_full[self.io].copy()
        self.n_joint_eval = 0
        self._cache_x = None
        self._cache_f = None
        self.last_hessian = None

    # -- joint pieces ------------------------------------------------------------------------------
    def _full(self, theta, u):
        p = self.par_full.copy()
        p[self.io] = theta
        p[self.ir] = u
        return p

    def _g(self, theta, u):
        self.n_joint_eval += 1
        v, g = self.joint(self._full(theta, u))
        return v, g[self.ir]

    def _hess_uu(self, theta, u):
        n = len(u)
        H = np.zeros((n, n))
        h = self.hess_step
        for k in range(n):
            e = np.zeros(n)
            e[k] = h * max(1.0, abs(u[k]))
            gp = self._g(theta, u + e)[1]
            gm = self._g(theta, u - e)[1]
            H[:, k] = (gp - gm) / (2 * e[k])
        return 0.5 * (H + H.T)

    def inner(self, theta, u0=None):
        """Newton solve for u_hat(theta); returns (u_hat, g value, H_uu)."""
        u = self.u_hat.copy() if u0 is None else np.array(u0, dtype=np.float64)
        val, gu = self._g(theta, u)
        H = None
        for _ in range(self.max_newton):
            H = self._hess_uu(theta, u)
            # Away from the inner optimum the joint nllk need not be convex in u (a penalty with a null space leaves
            # those directions to the data): shift an indefinite Hessian to positive definite for the STEP
            # (Levenberg), so that the iteration keeps descending towards a minimum, where H itself is positive
            w = np.linalg.eigvalsh(H)
            Hs = H if w[0] > 1e-8 * max(1.0, abs(w[-1])) else H + (abs(w[0]) + 1e-3 * max(1.0, abs(w[-1]))) * np.eye(len(u))
            try:
                step = np.linalg.solve(Hs, gu)
            except np.linalg.LinAlgError:
                step = np.linalg.lstsq(Hs, gu, rcond=None)[0]
            # backtracking: the joint is close to quadratic in u, a full step almost always passes
            t = 1.0
            for _ in range(20):
                v_new, g_new = self._g(theta, u - t * step)
                if np.isfinite(v_new) and v_new <= val + 1e-12 * abs(val):
                    break
                t *= 0.5
            u = u - t * step
            done = np.max(np.abs(t * step)) <= self.newton_tol * max(1.0, np.max(np.abs(u)))
            val, gu = v_new, g_new
            if done:
                break
        if H is None:
            H = self._hess_uu(theta, u)
        return u, val, H

    # -- marginal objective ----------------------------------------------------------------------------
    def fn(self, theta=None, update_warm_start=True):
        theta = self.par if theta is None else np.asarray(theta, dtype=np.float64)
        if self._cache_x is not None and np.array_equal(theta, self._cache_x):
            return self._cache_f
        u, val, H = self.inner(theta)
        sign, logdet = np.linalg.slogdet(H)
        f = val + 0.5 * logdet - 0.5 * len(u) * np.log(2 * np.pi) if sign > 0 else np.inf
        if update_warm_start:
            self.u_hat = u
            self.last_hessian = H
            self._cache_x, self._cache_f = theta.copy(), f
        return f

    def gr(self, theta=None):
        theta = self.par if theta is None else np.asarray(theta, dtype=np.float64)
        self.fn(theta)                       # centres the warm start on theta
        u_keep, H_keep, cx, cf = self.u_hat.copy(), self.last_hessian, self._cache_x, self._cache_f
        g = np.zeros(len(theta))
        for k in range(len(theta)):
            e = np.zeros(len(theta))
            e[k] = self.fd_step * max(1.0, abs(theta[k]))
            self.u_hat = u_keep.copy()
            fp = self.fn(theta + e, update_warm_start=False)
            self.u_hat = u_keep.copy()
            fm = self.fn(theta - e, update_warm_start=False)
            g[k] = (fp - fm) / (2 * e[k])
        self.u_hat, self.last_hessian, self._cache_x, self._cache_f = u_keep, H_keep, cx, cf
        return g


class EngineLaplaceObjective(LaplaceObjective):
    """The same objective with fn / gr taken from the engine's own entry point, `ssde_laplace_eval` (include/ssde.h;
    csrc/ssde_laplace.hip) -- the code an R or C host runs for `random = "coeff_re"`: inner Newton solve and differenced
    H_uu in C++ on the device gradient, outer gradient = exact dg/dtheta at u_hat + a differenced log-determinant term
    (no inner solves per outer coordinate).  `joint`, `inner`, `_hess_uu` of the base class stay available (sdreport)."""

    def __init__(self, engine, par_full, idx_outer, idx_random, **kw):
        super().__init__(lambda p: engine.eval(p, order=1), par_full, idx_outer, idx_random, **kw)
        self.engine = engine
        self._cache_gx = None
        self._cache_g = None

    def _call(self, theta, order):
        p = self._full(theta, self.u_hat)
        f, g, p_hat, H = self.engine.laplace_eval(p, order=order, want_hessian=True, hess_step=self.hess_step,
                                                   fd_step=self.fd_step, newton_tol=self.newton_tol, max_newton=self.max_newton)
        return f, g[self.io], p_hat[self.ir], H

    def fn(self, theta=None, update_warm_start=True):
        theta = self.par if theta is None else np.asarray(theta, dtype=np.float64)
        if self._cache_x is not None and np.array_equal(theta, self._cache_x):
            return self._cache_f
        f, _, u, H = self._call(theta, 0)
        if update_warm_start and np.isfinite(f):
            self.u_hat, self.last_hessian = u, H
            self._cache_x, self._cache_f = theta.copy(), f
        return f

    def gr(self, theta=None):
        theta = self.par if theta is None else np.asarray(theta, dtype=np.float64)
        if self._cache_gx is not None and np.array_equal(theta, self._cache_gx):
            return self._cache_g
        f, g, u, H = self._call(theta, 1)
        if np.isfinite(f):
            self.u_hat, self.last_hessian = u, H
            self._cache_x, self._cache_f = theta.copy(), f
        self._cache_gx, self._cache_g = theta.copy(), g
        return g
