"""Seeded synthetic track batches.

The generators follow the reference's exact-transition simulator
(/root/reference/R/sde.R:1434-1478, CTCRW covariance /root/reference/R/utility.R:188-196),
vectorised over tracks instead of looping over IDs.  `backend="torch"` builds the batch
directly in HBM on the current GPU (no 2.4 GB host->device copy for the 1e4 x 1e4 case);
`backend="numpy"` builds it on the host; `backend="hip"` is the engine's own simulator kernel
(ssde_simulate, csrc/k_sim.hip: counter-based, so any shard of a batch can be generated on its own).

Output is the reference's long format: all tracks concatenated, `ID` constant within a
track, `time` increasing globally (1..n scaled by dt, like /root/reference/inst/example.R:17,
which keeps quirk Q4 out of play), obs of shape (n, d).
"""
from __future__ import annotations

import math

import numpy as np


def _ctcrw_noise_chol(beta, sigma, dt):
    """Cholesky factor of CTCRW_cov(beta, sigma, dt) in (velocity, position) order."""
    e = math.exp(-beta * dt)
    e2 = math.exp(-2 * beta * dt)
    qvv = sigma ** 2 / (2 * beta) * (1 - e2)
    qzz = (sigma / beta) ** 2 * (dt + (1 - e2) / (2 * beta) - 2 * (1 - e) / beta)
    qvz = sigma ** 2 / (2 * beta ** 2) * (1 - 2 * e + e2)
    l11 = math.sqrt(qvv)
    l21 = qvz / l11
    l22 = math.sqrt(max(qzz - l21 * l21, 0.0))
    return e, l11, l21, l22


def simulate(model: str, n_tracks: int, n_steps: int, n_dim: int = 2, *, mu=0.0, tau=2.0, nu=1.0,
             kappa=1.0, sigma=1.0, sigma_obs=0.1, dt: float = 1.0, z0=0.0, seed: int = 1,
             backend: str = "numpy", device=None):
    """Simulate `n_tracks` tracks of `n_steps` rows each.

    model: "CTCRW" | "OU" | "OU_SSM" | "BM" | "BM_SSM".  The *_SSM models and CTCRW get
    N(0, sigma_obs^2) observation noise.  Returns (ID, times, obs) with shapes
    (n,), (n,), (n, d), n = n_tracks * n_steps, as numpy arrays or torch tensors.
    """
    M, T, d = int(n_tracks), int(n_steps), int(n_dim)
    if backend == "hip":
        from . import capi
        return capi.simulate_device(model, M, T, d, mu=mu, tau=tau, nu=nu, kappa=kappa, sigma=sigma, sigma_obs=sigma_obs,
                                    dt=dt, z0=z0, seed=seed, device=device)
    mu = np.broadcast_to(np.asarray(mu, dtype=np.float64), (d,))
    z0 = np.broadcast_to(np.asarray(z0, dtype=np.float64), (d,))
    use_torch = backend == "torch"
    if use_torch:
        import torch
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed))

        def randn(*shape):
            return torch.randn(*shape, dtype=torch.float64, device=device, generator=gen)

        def empty(*shape):
            return torch.empty(*shape, dtype=torch.float64, device=device)
    else:
        rng = np.random.default_rng(int(seed))

        def randn(*shape):
            return rng.standard_normal(shape)

        def empty(*shape):
            return np.empty(shape, dtype=np.float64)

    out = empty(M, T, d)  # track-major so that the long format is a plain reshape
    for a in range(d):
        if model == "CTCRW":
            beta = 1.0 / tau
            sig = 2.0 * nu / math.sqrt(tau * math.pi)
            e, l11, l21, l22 = _ctcrw_noise_chol(beta, sig, dt)
            v = randn(M) * 0.0
            z = v + float(z0[a])
            out[:, 0, a] = z
            for t in range(1, T):
                n1, n2 = randn(M), randn(M)
                z = z + mu[a] * dt + (v - mu[a]) / beta * (1 - e) + l21 * n1 + l22 * n2
                v = e * v + (1 - e) * mu[a] + l11 * n1
                out[:, t, a] = z
        elif model in ("OU", "OU_SSM"):
            e = math.exp(-dt / tau)
            sd = math.sqrt(kappa * (1 - math.exp(-2 * dt / tau)))
            z = randn(M) * 0.0 + float(z0[a])
            out[:, 0, a] = z
            for t in range(1, T):
                z = e * z + (1 - e) * mu[a] + sd * randn(M)
                out[:, t, a] = z
        elif model in ("BM", "BM_SSM"):
            sd = sigma * math.sqrt(dt)
            z = randn(M) * 0.0 + float(z0[a])
            out[:, 0, a] = z
            for t in range(1, T):
                z = z + mu[a] * dt + sd * randn(M)
                out[:, t, a] = z
        else:
            raise ValueError(f"no simulator for model {model!r}")
    if model in ("CTCRW", "OU_SSM", "BM_SSM") and sigma_obs > 0:
        out = out + sigma_obs * randn(M, T, d)
    n = M * T
    obs = out.reshape(n, d)
    if use_torch:
        import torch
        ID = torch.arange(M, dtype=torch.float64, device=device).repeat_interleave(T)
        times = torch.arange(1, n + 1, dtype=torch.float64, device=device) * dt
    else:
        ID = np.repeat(np.arange(M, dtype=np.float64), T)
        times = np.arange(1, n + 1, dtype=np.float64) * dt
    return ID, times, obs


def bspline_basis(x, n_basis: int = 9, degree: int = 3):
    """Clamped cubic B-spline design block on [0, 1] (stand-in for an mgcv smooth in the
    synthetic C3 configuration; mgcv itself stays on the R side and is out of scope).
    The first column is dropped after centring so the block has `n_basis` columns and no
    intercept, like the X_list_re of /root/reference/R/sde.R:417."""
    from scipy.interpolate import BSpline
    x = np.clip(np.asarray(x, dtype=np.float64), 0.0, 1.0)
    k = n_basis + 1
    inner = np.linspace(0, 1, k - degree + 1)
    knots = np.concatenate([[0.0] * degree, inner, [1.0] * degree])
    B = BSpline.design_matrix(x, knots, degree).toarray()
    return np.asfortranarray(B[:, 1:] - B[:, 1:].mean(axis=0))


class PPBasis:
    """A design block as a piecewise-cubic function of one covariate (include/ssde.h: ssde_ppbasis):
    X[i, k] = sum_m coef[iv, k, m] (x[i] - knots[iv])^m."""

    def __init__(self, x, knots, coef):
        self.x = x                                                   # (n,) numpy array or CUDA tensor
        self.knots = np.ascontiguousarray(knots, dtype=np.float64)
        self.coef = np.ascontiguousarray(coef, dtype=np.float64)     # (n_knots - 1, K, 4)
        assert self.coef.shape[0] == len(self.knots) - 1 and self.coef.shape[2] == 4
        self.n_cols = self.coef.shape[1]

    def dense(self):
        """the n x K matrix the table stands for (host evaluation, for the oracle and for comparisons)"""
        x = np.asarray(self.x.cpu().numpy() if hasattr(self.x, "cpu") else self.x, dtype=np.float64)
        iv = np.clip(np.searchsorted(self.knots, x, side="right") - 1, 0, len(self.knots) - 2)
        t = x - self.knots[iv]
        return np.asfortranarray(sum(self.coef[iv, :, m] * t[:, None] ** m for m in range(4)))


def bspline_ppbasis(x, n_basis: int = 9, degree: int = 3, centre=None):
    """`bspline_basis` as a PPBasis: the same clamped cubic B-spline block (first column dropped, columns centred),
    as per-interval polynomial coefficients.  `centre`: column means to subtract (default: the means over x)."""
    from scipy.interpolate import BSpline, PPoly
    k = n_basis + 1
    inner = np.linspace(0, 1, k - degree + 1)
    knots = np.concatenate([[0.0] * degree, inner, [1.0] * degree])
    tab = np.zeros((len(inner) - 1, k, 4))
    for j in range(k):
        c = np.zeros(k)
        c[j] = 1.0
        pp = PPoly.from_spline((knots, c, degree))
        for iv in range(len(inner) - 1):
            idx = np.where((pp.x[:-1] == inner[iv]) & (pp.x[1:] == inner[iv + 1]))[0][0]
            tab[iv, j, :] = pp.c[::-1, idx]
    tab = tab[:, 1:, :].copy()
    basis = PPBasis(x, inner, tab)
    if centre is None:
        xs = np.clip(np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x, dtype=np.float64), 0.0, 1.0)
        centre = PPBasis(xs, inner, tab).dense().mean(axis=0)
    basis.coef[:, :, 0] -= np.asarray(centre)[None, :]
    return basis


def second_difference_penalty(k: int):
    """S = D2' D2 + small ridge (full rank, like mgcv's shrinkage bases "ts"/"cs")."""
    D = np.diff(np.eye(k), n=2, axis=0)
    return D.T @ D + 1e-2 * np.eye(k)
