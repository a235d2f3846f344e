"""Replacement for the pieces of `TMB::sdreport` the reference consumes.

The reference calls `sdreport(tmb_obj, getJointPrecision = TRUE)` after the fit
(/root/reference/R/sde.R:702-704) and then only uses
  * `as.list(rep, "Estimate")`                 -> coefficient write-back       (R/sde.R:707-719)
  * `rep$par.fixed`, `rep$par.random`          -> logLik / AIC                 (R/utility.R:115-123, R/sde.R:1318-1324)
  * `rep$jointPrecision` / `rep$cov.fixed`     -> posterior draws, conditional EDF (R/sde.R:871-882, 1360-1375)
`sdreport` itself needs TMB's tape, so it cannot take a non-TMB objective.  Here the same quantities are built
from the GPU gradient: every Hessian is a central finite difference of `gr` (2 p gradient evaluations, 0.1-1 ms
each on the device), which is what makes this affordable without an AD tape.

  hessian.random  H_uu        joint Hessian block over coeff_re at (theta_hat, u_hat)
  hessian.fixed   d2 f / d theta2 of the Laplace marginal f (second differences of `fn`; for models without
                  random effects: the Hessian of the objective itself)
  cov.fixed       solve(hessian.fixed)
  jointPrecision  [[hessian.fixed + H_tu H_uu^-1 H_ut, H_tu], [H_ut, H_uu]]   (fixed first, then random:
                  the order R/sde.R:889-891 insists on), i.e. u | theta ~ N(u_hat(theta), H_uu^-1) with
                  d u_hat / d theta = -H_uu^-1 H_ut and theta ~ N(theta_hat, cov.fixed)

Accuracy: TMB's numbers are AD-exact; these carry finite-difference error (relative 1e-6 on joint Hessian
blocks, 1e-3 on hessian.fixed of a Laplace marginal, whose value already contains a finite-difference Hessian).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np


def fd_hessian(gr: Callable[[np.ndarray], np.ndarray], x: np.ndarray, rel_step: float = 1e-4) -> np.ndarray:
    """Symmetrised central-difference Hessian from a gradient function (`tmb_obj$he` counterpart)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    H = np.zeros((n, n))
    for k in range(n):
        h = rel_step * max(1.0, abs(x[k]))
        e = np.zeros(n)
        e[k] = h
        H[:, k] = (np.asarray(gr(x + e)) - np.asarray(gr(x - e))) / (2 * h)
    return 0.5 * (H + H.T)


def fd_hessian_fn(fn: Callable[[np.ndarray], float], x: np.ndarray, rel_step: float = 1e-2) -> np.ndarray:
    """Second central differences of a scalar function (used on the Laplace marginal, whose gradient is itself
    a finite difference)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    h = rel_step * np.maximum(1.0, np.abs(x))
    f0 = fn(x)
    H = np.zeros((n, n))
    for i in range(n):
        ei = np.zeros(n)
        ei[i] = h[i]
        H[i, i] = (fn(x + ei) - 2 * f0 + fn(x - ei)) / h[i] ** 2
        for j in range(i):
            ej = np.zeros(n)
            ej[j] = h[j]
            H[i, j] = H[j, i] = (fn(x + ei + ej) - fn(x + ei - ej) - fn(x - ei + ej) + fn(x - ei - ej)) / (4 * h[i] * h[j])
    return H


class SdReport:
    """The members of an `sdreport` object that the reference reads."""

    def __init__(self, par_fixed, names_fixed, par_random, names_random, cov_fixed, joint_precision,
                 hessian_fixed, hessian_random):
        self.par_fixed = np.asarray(par_fixed, dtype=np.float64)
        self.names_fixed: List[str] = list(names_fixed)
        self.par_random = np.asarray(par_random, dtype=np.float64)
        self.names_random: List[str] = list(names_random)
        self.cov_fixed = cov_fixed
        self.jointPrecision = joint_precision
        self.hessian_fixed = hessian_fixed
        self.hessian_random = hessian_random

    def par_all(self):
        """c(par.fixed, par.random) (R/sde.R:885, R/utility.R:116-117)"""
        return np.concatenate([self.par_fixed, self.par_random])

    def names_all(self):
        return self.names_fixed + self.names_random

    def as_list(self, what: str = "Estimate"):
        """as.list(rep, "Estimate") / "Std. Error": one vector per parameter block."""
        if what == "Estimate":
            vals = self.par_all()
        elif what == "Std. Error":
            if self.jointPrecision is not None:
                vals = np.sqrt(np.diag(np.linalg.inv(self.jointPrecision)))
            else:
                vals = np.sqrt(np.diag(self.cov_fixed))
        else:
            raise ValueError("what must be 'Estimate' or 'Std. Error'")
        out = {}
        names = self.names_all()
        for nm in dict.fromkeys(names):
            out[nm] = np.array([v for v, k in zip(vals, names) if k == nm])
        return out


def _block_name(pb, k: int) -> str:
    if k < len(getattr(pb, "lead_names", [])):
        return pb.lead_names[k]
    if not hasattr(pb, "lead_names") and pb.kalman and k == 0:
        return "log_sigma_obs"
    if pb.off_fe <= k < pb.off_fe + pb.n_fe:
        return "coeff_fe"
    if pb.off_lambda <= k < pb.off_lambda + pb.n_smooth:
        return "log_lambda"
    if getattr(pb, "n_decay", 0) and pb.off_decay <= k < pb.off_decay + pb.n_decay:
        return "log_decay"
    return "coeff_re"


def sdreport(problem, joint_eval: Callable[[np.ndarray], tuple], par_full: np.ndarray, idx_fixed, idx_random,
             marginal_fn: Optional[Callable[[np.ndarray], float]] = None, rel_step: float = 1e-4,
             marginal_step: float = 1e-2, joint_hess: Optional[Callable] = None) -> SdReport:
    """problem       capi.Problem (parameter layout / block names)
    joint_eval    par_full -> (value, grad_full): the joint penalised nllk (Engine.eval)
    par_full      full parameter vector at the optimum (theta_hat, u_hat)
    idx_fixed     indices of the free non-random parameters, idx_random of the free coeff_re entries
    marginal_fn   theta -> Laplace marginal (LaplaceObjective.fn); None when there are no random effects
    joint_hess    (par_full, idx) -> exact Hessian of the joint nllk over idx, or None where the engine has none
                  (Engine.hess = ssde_hess: direct families BM / OU); replaces the differenced gradient"""
    par_full = np.asarray(par_full, dtype=np.float64)
    io = np.asarray(idx_fixed, dtype=int)
    ir = np.asarray(idx_random, dtype=int)
    free = np.concatenate([io, ir])

    def gr_free(x):
        p = par_full.copy()
        p[free] = x
        return joint_eval(p)[1][free]

    names_f = [_block_name(problem, k) for k in io]
    names_r = [_block_name(problem, k) for k in ir]
    H = None
    if joint_hess is not None:
        H = joint_hess(par_full, free)
    if H is None:
        H = fd_hessian(gr_free, par_full[free], rel_step)
    nf = len(io)
    if len(ir) == 0:
        cov = np.linalg.inv(H)
        return SdReport(par_full[io], names_f, [], [], cov, None, H, None)
    Htt, Htu, Huu = H[:nf, :nf], H[:nf, nf:], H[nf:, nf:]
    if marginal_fn is not None:
        Hfix = fd_hessian_fn(marginal_fn, par_full[io], marginal_step)
    else:  # Schur complement: the marginal Hessian without the log-determinant term's curvature
        Hfix = Htt - Htu @ np.linalg.solve(Huu, Htu.T)
    cov = np.linalg.inv(Hfix)
    Q = np.zeros_like(H)
    Q[:nf, :nf] = Hfix + Htu @ np.linalg.solve(Huu, Htu.T)
    Q[:nf, nf:] = Htu
    Q[nf:, :nf] = Htu.T
    Q[nf:, nf:] = Huu
    return SdReport(par_full[io], names_f, par_full[ir], names_r, cov, Q, Hfix, Huu)
