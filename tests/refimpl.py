"""Independent float64 restatements used to cross-check the C++ oracle (test infrastructure).

These are deliberately NOT the recursion of oracle/ssde_oracle.hpp:

* Kalman families: the exact joint Gaussian density of all scored observations of a track,
  built as a dense covariance matrix (prior N(a0, P0) on the state at the track's 2nd row,
  SURVEY.md Appendix A / Q1), evaluated with torch.linalg -- no filter recursion at all.
  The CTCRW transition covariance is taken from the reference's *second* statement of it,
  R's CTCRW_cov (/root/reference/R/utility.R:188-196), not from makeQ_ctcrw.
* direct families: torch.distributions.Normal log-densities with the means / sds of
  /root/reference/R/sde.R:1436-1446 (the simulator's statement of the same transition).
* gradients: torch autograd through those expressions.

Sizes: O(T^2) blocks per track, for fixtures of a few dozen rows per track only.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from smoothsde_amd.capi import Problem

torch.set_default_dtype(torch.float64)


def _is_na(x, na_mode):
    if not np.isnan(x):
        return False
    if na_mode == 1:
        return True
    return (np.array([x]).view(np.uint64)[0] & 0xFFFFFFFF) == 1954


def linear_predictor(pb: Problem, par: torch.Tensor) -> torch.Tensor:
    """par_mat (n x q): /root/reference/src/nllk/nllk_ctcrw.hpp:143-149."""
    cols = []
    for j in range(pb.q):
        cf = par[pb.off_fe + pb.fe_off[j]: pb.off_fe + pb.fe_off[j] + pb.ncol_fe[j]]
        if pb.X_fe[j] is None:
            col = cf[0] * torch.ones(pb.n)
        else:
            col = torch.as_tensor(pb.X_fe[j]) @ cf
        if pb.ncol_re[j] > 0:
            cr = par[pb.off_re + pb.re_off[j]: pb.off_re + pb.re_off[j] + pb.ncol_re[j]]
            Xr = torch.as_tensor(pb.X_re[j])
            if getattr(pb, "n_decay", 0) > 0:
                # X_re_decay of the R class (R/sde.R:303-321): whole columns scaled by exp(-rho * t_decay)
                scaled = []
                for c in range(pb.ncol_re[j]):
                    k = pb.decay_of_col[pb.re_off[j] + c]
                    xc = Xr[:, c]
                    if k >= 0:
                        rho = torch.exp(par[pb.off_decay + k])
                        xc = xc * torch.exp(-rho * torch.as_tensor(pb.t_decay[j * pb.n:(j + 1) * pb.n]))
                    scaled.append(xc)
                Xr = torch.stack(scaled, dim=1)
            col = col + Xr @ cr
        cols.append(col)
    return torch.stack(cols, dim=1)


def penalty(pb: Problem, par: torch.Tensor) -> torch.Tensor:
    pen = torch.zeros(())
    if pb.n_smooth == 0:
        return pen
    if not pb.kalman and not pb.include_penalty:
        return pen
    start = 0
    for s, S in enumerate(pb.S_list):
        Sn = S.shape[0]
        b = par[pb.off_re + start: pb.off_re + start + Sn]
        ll = par[pb.off_lambda + s]
        St = torch.as_tensor(S)
        pen = pen - 0.5 * Sn * ll + 0.5 * torch.exp(ll) * (b @ (St @ b))
        if not pb.kalman:  # nllk_sde.hpp:108-116
            pen = pen + 0.5 * Sn * math.log(2 * math.pi) - 0.5 * torch.linalg.slogdet(St)[1]
        start += Sn
    return pen


def _ctcrw_blocks(beta, sigma, dt):
    """2x2 per-dimension (position, velocity) transition, drift loading and covariance."""
    e = torch.exp(-beta * dt)
    T = torch.stack([torch.stack([torch.ones(()), (1 - e) / beta]),
                     torch.stack([torch.zeros(()), e])])
    B = torch.stack([dt - (1 - e) / beta, 1 - e])
    # CTCRW_cov (R/utility.R:188-196) is ordered (velocity, position)
    qvv = sigma ** 2 / (2 * beta) * (1 - torch.exp(-2 * beta * dt))
    qzz = (sigma / beta) ** 2 * (dt + (1 - torch.exp(-2 * beta * dt)) / (2 * beta) - 2 * (1 - e) / beta)
    qvz = sigma ** 2 / (2 * beta ** 2) * (1 - 2 * e + torch.exp(-2 * beta * dt))
    Q = torch.stack([torch.stack([qzz, qvz]), torch.stack([qvz, qvv])])
    return T, B, Q


def _step_matrices(pb: Problem, pm_row: torch.Tensor, dt: float):
    d, sdim = pb.n_dim, pb.sdim
    if pb.model == "CTCRW":
        tau, nu = torch.exp(pm_row[d]), torch.exp(pm_row[d + 1])
        beta = 1 / tau
        sigma = 2 * nu / torch.sqrt(math.pi * tau)
        T2, B2, Q2 = _ctcrw_blocks(beta, sigma, dt)
        T = torch.zeros(sdim, sdim)
        Q = torch.zeros(sdim, sdim)
        c = torch.zeros(sdim)
        for a in range(d):
            T = T + torch.nn.functional.pad(T2, (2 * a, sdim - 2 * a - 2, 2 * a, sdim - 2 * a - 2))
            Q = Q + torch.nn.functional.pad(Q2, (2 * a, sdim - 2 * a - 2, 2 * a, sdim - 2 * a - 2))
            c = c + torch.nn.functional.pad(B2 * pm_row[a], (2 * a, sdim - 2 * a - 2))
        return T, Q, c
    if pb.model == "OU_SSM":
        tau, kappa = torch.exp(pm_row[d]), torch.exp(pm_row[d + 1])
        e = torch.exp(-dt / tau)
        return e * torch.eye(d), kappa * (1 - torch.exp(-2 * dt / tau)) * torch.eye(d), (1 - e) * pm_row[:d]
    sigma = torch.exp(pm_row[d])
    return torch.eye(d), sigma ** 2 * dt * torch.eye(d), pm_row[:d] * dt


def kalman_dense_nllk(pb: Problem, par: torch.Tensor) -> torch.Tensor:
    """-log p(scored observations) without 2*pi constants (Q2), summed over tracks."""
    d, sdim, n = pb.n_dim, pb.sdim, pb.n
    pm = linear_predictor(pb, par)
    Z = torch.zeros(d, sdim)
    for a in range(d):
        Z[a, 2 * a if pb.model == "CTCRW" else a] = 1.0
    h_iso = torch.exp(par[0]) ** 2
    if pb.P0 is not None:
        P0 = torch.as_tensor(pb.P0)
    elif pb.model == "CTCRW":
        P0 = torch.diag(torch.tensor([1.0, 10.0] * d))
    else:
        P0 = 10.0 * torch.eye(d)
    total = torch.zeros(())
    bounds = list(pb.seg_start) + [n]
    for k in range(pb.n_seg):
        r0, r1 = bounds[k], bounds[k + 1]
        if r1 - r0 < 2:
            continue
        if pb.a0 is not None:
            m = torch.as_tensor(pb.a0[k])
        else:
            m = torch.zeros(sdim)
            for a in range(d):
                m[2 * a if pb.model == "CTCRW" else a] = pb.obs[r0, a]
        V = P0
        rows = list(range(r0 + 1, r1))
        means, Vs, Ts = [], [], []
        for i in rows:
            means.append(m)
            Vs.append(V)
            dt = pb.times[i + 1] - pb.times[i] if i < n - 1 else 1.0
            T, Q, c = _step_matrices(pb, pm[i], float(dt))
            Ts.append(T)
            m = T @ m + c
            V = T @ V @ T.T + Q
        keep = [t for t, i in enumerate(rows) if not _is_na(pb.obs[i, 0], pb.na_mode)]
        if not keep:
            continue
        L = len(keep)
        Sig = torch.zeros(L * d, L * d)
        r = torch.zeros(L * d)
        for a_, t in enumerate(keep):
            i = rows[t]
            H = torch.as_tensor(pb.H[:, :, i]) if pb.H is not None else h_iso * torch.eye(d)
            r[a_ * d:(a_ + 1) * d] = torch.as_tensor(pb.obs[i]) - Z @ means[t]
            # Cov(s_u, s_t) = Phi(u, t) V_t for u >= t
            C = Vs[t]
            ptr = a_
            for u in range(t, rows.__len__()):
                if ptr < L and keep[ptr] == u:
                    blk = Z @ C @ Z.T
                    if u == t:
                        blk = blk + H
                    Sig[ptr * d:(ptr + 1) * d, a_ * d:(a_ + 1) * d] = blk
                    Sig[a_ * d:(a_ + 1) * d, ptr * d:(ptr + 1) * d] = blk.T
                    ptr += 1
                C = Ts[u] @ C
        sign, logdet = torch.linalg.slogdet(Sig)
        total = total + 0.5 * (logdet + r @ torch.linalg.solve(Sig, r))
    return total


def eseal_dense_nllk(pb: Problem, par: torch.Tensor) -> torch.Tensor:
    """ESEAL_SSM (nllk_e_seal_ssm.hpp:139-207) as a dense joint Gaussian per track: the first state component is the
    constant 1 (P0 = diag(0, 10)), so  y_i = a1 + (a2 / R_i) L_i + N(0, tau^2 / h_i),
    L_{i+1} = L_i + mu_i dt_i + N(0, sigma_i^2 dt_i), and the prior N(a0_L, P0[1,1]) sits on the track's SECOND row (Q1)."""
    n = pb.n
    pm = linear_predictor(pb, par)
    tau, a1, a2 = torch.exp(par[0]), par[1], torch.exp(par[2])
    p0 = 10.0 if pb.P0 is None else float(pb.P0[1, 1])
    total = torch.zeros(())
    starts = list(pb.seg_start) + [n]
    for k in range(pb.n_seg):
        lo, hi = starts[k], starts[k + 1]
        rows = list(range(lo + 1, hi))
        if not rows:
            continue
        m = len(rows)
        dts = [float(pb.times[i + 1] - pb.times[i]) if i < n - 1 else 1.0 for i in rows]
        mean = [torch.as_tensor(float(pb.a0[k, 1]))]
        var = [torch.as_tensor(p0)]
        for t in range(m - 1):
            i = rows[t]
            mean.append(mean[-1] + pm[i, 0] * dts[t])
            var.append(var[-1] + torch.exp(pm[i, 1]) ** 2 * dts[t])
        mean, var = torch.stack(mean), torch.stack(var)
        idx = torch.arange(m)
        covL = var[torch.minimum(idx[:, None], idx[None, :])]            # Cov(L_j, L_k) = Var(L_min(j,k))
        z = a2 / torch.as_tensor(pb.eseal_R[rows])
        Sig = z[:, None] * covL * z[None, :] + torch.diag(tau ** 2 / torch.as_tensor(pb.eseal_h[rows]))
        keep = [t for t in range(m) if not _is_na(pb.obs[rows[t], 0], pb.na_mode)]
        if not keep:
            continue
        kk = torch.tensor(keep)
        r = torch.as_tensor(pb.obs[rows, 0])[kk] - (a1 + z * mean)[kk]
        S = Sig[kk][:, kk]
        total = total + 0.5 * (torch.linalg.slogdet(S)[1] + r @ torch.linalg.solve(S, r))
    return total


def eseal_priors(pb: Problem, par: torch.Tensor) -> torch.Tensor:
    """-(inverse-gamma log priors) of nllk_e_seal_ssm.hpp:212-216 via torch.distributions (an inverse gamma on x is a
    gamma on 1/x with the Jacobian x^-2); integer division n/2 as in the reference."""
    n = pb.n
    pm0 = linear_predictor(pb, par)[0]
    out = torch.zeros(())
    for x, shape, scale in ((torch.exp(pm0[1]) ** 2, float(10 * n), float(4 * (10 * n - 1))),
                            (torch.exp(par[0]) ** 2, float(n // 2), float(n // 2 - 1))):
        lp = torch.distributions.Gamma(shape, scale).log_prob(1.0 / x) - 2.0 * torch.log(x)
        out = out - lp
    return out


def _mp_log_bessel(x, nu, nx, nn):
    """d^(nx + nn) log I_nu(x) / dx^nx dnu^nn from mpmath at 30 digits (independent of the series in oracle/ and csrc/)"""
    import mpmath as mp
    mp.mp.dps = 30
    f = lambda a, b: mp.log(mp.besseli(b, a))
    xf, nf = mp.mpf(float(x)), mp.mpf(float(nu))
    if nx == 0 and nn == 0:
        return float(f(xf, nf))
    return float(mp.diff(f, (xf, nf), (nx, nn)))


class _LogBesselIDeriv(torch.autograd.Function):
    """a first derivative of log I_nu(x) as a differentiable function of (x, nu): its own derivatives are the second ones"""

    @staticmethod
    def forward(ctx, x, nu, wrt_nu):
        ctx.save_for_backward(x, nu)
        ctx.wrt_nu = wrt_nu
        return torch.tensor(_mp_log_bessel(x, nu, 0 if wrt_nu else 1, 1 if wrt_nu else 0), dtype=torch.float64)

    @staticmethod
    def backward(ctx, g):
        x, nu = ctx.saved_tensors
        k = 1 if ctx.wrt_nu else 0
        return g * _mp_log_bessel(x, nu, 2 - k, k), g * _mp_log_bessel(x, nu, 1 - k, 1 + k), None


class _LogBesselI(torch.autograd.Function):
    """log I_nu(x); first derivatives by mpmath, and (through _LogBesselIDeriv) second ones for autograd Hessians"""

    @staticmethod
    def forward(ctx, x, nu):
        ctx.save_for_backward(x, nu)
        return torch.tensor(_mp_log_bessel(x, nu, 0, 0), dtype=torch.float64)

    @staticmethod
    def backward(ctx, g):
        x, nu = ctx.saved_tensors
        return g * _LogBesselIDeriv.apply(x, nu, False), g * _LogBesselIDeriv.apply(x, nu, True)


def direct_nllk(pb: Problem, par: torch.Tensor) -> torch.Tensor:
    """BM / OU transition densities (nllk_sde.hpp:77-84) via torch.distributions."""
    d, n = pb.n_dim, pb.n
    pm = linear_predictor(pb, par)
    total = torch.zeros(())
    for i in range(1, n):
        if pb.id[i] != pb.id[i - 1]:
            continue
        dt = float(pb.times[i] - pb.times[i - 1])
        p = pm[i - 1]
        for a in range(d):
            z0, z1 = pb.obs[i - 1, a], pb.obs[i, a]
            if _is_na(z0, pb.na_mode) or _is_na(z1, pb.na_mode):
                continue
            if pb.model == "CIR":
                # tr_dens.hpp:53-67: c Z1 is non-central chi-square; written with log I_q (mpmath) instead of besselI
                mu, beta, sigma = torch.exp(p[a]), torch.exp(p[d]), torch.exp(p[d + 1])
                c = 2 * beta / ((1 - torch.exp(-beta * dt)) * sigma ** 2)
                q = 2 * beta * mu / sigma ** 2 - 1
                u, v = c * z0 * torch.exp(-beta * dt), c * z1
                ld = torch.log(c) - u - v + q / 2 * (torch.log(v) - torch.log(u)) + _LogBesselI.apply(2 * torch.sqrt(u * v), q)
                total = total - ld
                continue
            if pb.model == "BM_t":
                # scaled Student-t increment (tr_dens.hpp:38-44): torch.distributions.StudentT(df, loc, scale)
                df = float(pb.other_data[0])
                sd = torch.exp(p[1]) * math.sqrt(dt)
                scale = sd / math.sqrt(df / (df - 2))
                total = total - torch.distributions.StudentT(df, z0 + p[0] * dt, scale).log_prob(torch.tensor(z1))
                continue
            if pb.model == "BM":
                mean = z0 + p[a] * dt
                sd = torch.exp(p[d]) * math.sqrt(dt)
            else:
                tau, kappa = torch.exp(p[d]), torch.exp(p[d + 1])
                mean = torch.exp(-dt / tau) * z0 + (1 - torch.exp(-dt / tau)) * p[a]
                sd = torch.sqrt(kappa * (1 - torch.exp(-2 * dt / tau)))
            total = total - torch.distributions.Normal(mean, sd).log_prob(torch.tensor(z1))
    return total


def ref_eval(pb: Problem, par, with_penalty: bool = True):
    """(value, grad) by the independent restatement; grad over the full vector, fixed entries 0."""
    p = torch.tensor(np.asarray(par, dtype=np.float64), requires_grad=True)
    if pb.model == "ESEAL_SSM":
        val = eseal_dense_nllk(pb, p)
    else:
        val = kalman_dense_nllk(pb, p) if pb.kalman else direct_nllk(pb, p)
    if with_penalty:
        val = val + penalty(pb, p)
        if pb.model == "ESEAL_SSM":
            val = val + eseal_priors(pb, p)
    (g,) = torch.autograd.grad(val, p, allow_unused=True)
    g = np.zeros(pb.n_par_full) if g is None else g.numpy().copy()
    g[pb.par_fixed != 0] = 0.0
    return float(val.detach()), g
