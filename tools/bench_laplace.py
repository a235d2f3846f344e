#!/usr/bin/env python3
"""tools/bench_laplace.py [tracks] [rows] [K] [model] -- one ssde_laplace_eval (marginal nllk + gradient, coeff_re integrated
out: what `random = "coeff_re"` makes TMB's fn / gr be) on a smooth-drift batch, with the exact Hessian of the drift
coefficients (ssde_hess / k_iso_drift.hip) against a differenced one (SSDE_NO_EXACT_HESS=1), and the same for the direct
family (C3: OU with a smooth mean)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import second_difference_penalty  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 9
model = sys.argv[4] if len(sys.argv) > 4 else "OU_SSM"
dev = torch.device("cuda:0")
d = 1
ID, times, obs = capi.simulate_device(model, M, T, d, mu=2.0, tau=2.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, z0=2.0, seed=2, device=dev)
n = M * T
x = 0.5 + 0.45 * torch.sin(torch.arange(n, device=dev, dtype=torch.float64) * (2 * np.pi / 977.0))
X = torch.stack([torch.cos(np.pi * k * x) for k in range(1, K + 1)], dim=1)
q = capi.n_sde_par(model, d)
X_re = [None] * q
X_re[0] = X
pb = capi.Problem.from_torch(model, ID, times, obs, X_re=X_re, S_list=[second_difference_penalty(K)])
par = np.zeros(pb.n_par_full)
if pb.kalman:
    par[0] = np.log(0.1)
par[pb.off_fe] = 2.0
par[pb.off_fe + d] = np.log(2.0)
par[pb.off_lambda] = 1.0
for label, env in (("exact H_uu", None), ("differenced H_uu (SSDE_NO_EXACT_HESS=1)", "1")):
    if env:
        os.environ["SSDE_NO_EXACT_HESS"] = env
    else:
        os.environ.pop("SSDE_NO_EXACT_HESS", None)
    eng = capi.Engine(pb)
    p = par.copy()
    f, g, p_hat = eng.laplace_eval(p, order=1)                   # cold start (u = 0)
    n0 = eng.info()["n_evals"]
    t0 = time.perf_counter()
    reps = 3
    for r in range(reps):
        pp = p_hat.copy()
        pp[pb.off_lambda] += 0.01 * (r + 1)                       # an outer step, warm-started inner solve
        f2, g2, _ = eng.laplace_eval(pp, order=1)
    dt = (time.perf_counter() - t0) / reps
    inf = eng.info()
    print(f"{model} {M} x {T}, K = {K}, path {capi.PATH_NAMES[inf['path']]}: {label}: {1e3 * dt:.2f} ms per marginal value + gradient "
          f"({(inf['n_evals'] - n0) / reps:.0f} device evaluations each), f = {f2:.6f}, |g| = {np.max(np.abs(g2)):.3e}", flush=True)
    eng.close()
