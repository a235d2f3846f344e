#!/usr/bin/env python3
"""tools/bench_drift.py [tracks] [rows] [K] [model] [d] [na_fraction] [table] -- the shared-covariance kernel with a streamed row-varying drift
(k_iso_drift.hip): C3's model in its state-space form -- mu a K-column smooth of a covariate, tau / kappa / sigma_obs
constant -- at batch scale, against the lane = direction path (SSDE_NO_DRIFT=1) on a smaller batch of the same shape.
`table` as the seventh argument: the drift's block is a K-column cubic B-spline of the covariate given as an ssde_ppbasis table -- the tiles
carry the covariate (8 B/row), the lanes evaluate the columns (k_iso_drift_pp.hip).  Prints ms per nllk + gradient, the kernel's own time, rows/s and the fraction of 8 TB/s on the 8 (d + K) B/row it reads."""
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
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 9
model = sys.argv[4] if len(sys.argv) > 4 else "OU_SSM"
d = int(sys.argv[5]) if len(sys.argv) > 5 else 1
na_frac = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0      # > 0: missing rows -> the lanes carry their own covariance
table = len(sys.argv) > 7 and sys.argv[7] == "table"
dev = torch.device("cuda:0")


def build(M, T):
    ID, times, obs = capi.simulate_device(model, M, T, d, mu=2.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, z0=2.0, seed=2, device=dev)
    n = M * T
    if na_frac > 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(7)
        na = torch.rand(n, device=dev, generator=gen) < na_frac
        na[::T] = False
        obs = obs.contiguous()
        obs[na] = float("nan")
    x = 0.5 + 0.45 * torch.sin(torch.arange(n, device=dev, dtype=torch.float64) * (2 * np.pi / 977.0))
    q = capi.n_sde_par(model, d)
    if table:
        from smoothsde_amd.synth import bspline_ppbasis
        basis = [None] * q
        basis[0] = bspline_ppbasis(x, K, centre=np.zeros(K))
        return capi.Problem.from_torch(model, ID, times, obs, basis_re=basis, S_list=[second_difference_penalty(K)])
    X = torch.stack([torch.cos(np.pi * k * x) for k in range(1, K + 1)], dim=1)      # a smooth K-column basis of the covariate
    X_re = [None] * q
    X_re[0] = X
    pb = capi.Problem.from_torch(model, ID, times, obs, X_re=X_re, S_list=[second_difference_penalty(K)])
    return pb


def run(pb, evals=20):
    eng = capi.Engine(pb)
    inf = eng.info()
    rng = np.random.default_rng(1)
    par0 = np.zeros(pb.n_par_full)
    par0[0] = np.log(0.1)
    par0[pb.off_fe] = 2.0
    par0[pb.off_fe + d] = np.log(2.0)
    par0[pb.off_re:] = 0.05 * rng.standard_normal(pb.n_re)
    ths = [np.ascontiguousarray(par0 + 1e-3 * np.sin(k + np.arange(pb.n_par_full))) for k in range(evals + 3)]
    call = eng.bound_eval(order=1)
    for k in range(3):
        call(ths[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(evals):
        call(ths[3 + k])
    wall = (time.perf_counter() - t0) / evals
    kms = [m for m in eng.kernel_ms_history(min(evals, 64)) if m > 0]
    inf = eng.info()
    eng.close()
    return wall, float(np.mean(kms)) if kms else float("nan"), inf


pb = build(M, T)
wall, kms, inf = run(pb)
rows = inf["n_rows"]
print(f"{model} d={d} K={K}{', block as a table' if table else ''}{'' if na_frac == 0 else f', {100 * na_frac:.0f} % missing rows'}: {M} tracks x {T} rows, path {capi.PATH_NAMES[inf['path']]} const_coeff={inf['const_coeff']}: "
      f"{1e3 * wall:.3f} ms per nllk+gradient (kernel {kms:.3f} ms), {rows / wall:.3e} rows/s, "
      f"{inf['required_bytes_per_row']:.0f} B/row read: {inf['required_bytes_per_row'] * rows / (kms * 1e-3) / 1e12:.2f} TB/s = "
      f"{inf['required_bytes_per_row'] * rows / (kms * 1e-3) / 8e12:.3f} of 8 TB/s, windows {inf['lanes_per_track']}, warm-up {inf['window']}, "
      f"check {inf['window_check_max']:.1e}", flush=True)
del pb
torch.cuda.empty_cache()
if not os.environ.get("SSDE_NO_DRIFT"):
    os.environ["SSDE_NO_DRIFT"] = "1"
    Ms = min(M, 10_000)
    Ts = min(T, 1000)
    pb = build(Ms, Ts)
    wall, kms, inf = run(pb, evals=5)
    print(f"  lane = direction path (SSDE_NO_DRIFT=1), {Ms} tracks x {Ts} rows: {1e3 * wall:.3f} ms per nllk+gradient, "
          f"{inf['n_rows'] / wall:.3e} rows/s, path {capi.PATH_NAMES[inf['path']]}", flush=True)
