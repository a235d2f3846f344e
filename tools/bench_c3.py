#!/usr/bin/env python3
"""tools/bench_c3.py -- C3 (1e4 OU tracks x 1e4 rows, mu = 9-column spline of a covariate): the design block streamed
(88 B/row) against the same block handed over as a B-spline table and evaluated on the device (24 B/row resident)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate, second_difference_penalty, bspline_ppbasis  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
M, T = 10_000, 10_000
fast_cov = len(sys.argv) > 1 and sys.argv[1] == "fast"
ID, times, obs = simulate("OU", M, T, 1, mu=1.0, tau=2.0, kappa=1.0, seed=2, backend="torch", device=dev)
n = len(ID)
if fast_cov:   # a covariate that cycles every 24 rows (time of day): neighbouring rows of a wave sit in every interval
    x = 0.5 + 0.5 * torch.sin(torch.arange(n, device=dev, dtype=torch.float64) * (2 * np.pi / 24.0))
else:          # slowly wandering covariate (bench_configs.py's C3)
    x = torch.cumsum(torch.randn(n, device=dev, dtype=torch.float64) * 0.01, 0)
    x = (x - x.min()) / (x.max() - x.min())
par = np.concatenate([[1.0, np.log(2.0), 0.0], [0.0], 0.05 * np.sin(np.arange(9))])
basis = bspline_ppbasis(x, 9, centre=np.zeros(9))
pb = capi.Problem.from_torch("OU", ID, times, obs, basis_re=[basis, None, None], S_list=[second_difference_penalty(9)])
eng = capi.Engine(pb)
wall, inf = timed(eng, par, 8)
print(f"table  ({'fast' if fast_cov else 'slow'} covariate) ms/eval {1e3 * wall:.4f} kernel_ms {inf['main_kernel_ms']:.4f} resident GB {inf['hbm_bytes'] / 1e9:.2f}", flush=True)
v1, g1 = eng.eval(par)
eng.close()
if os.environ.get("SSDE_C3_STREAMED"):
    B = torch.as_tensor(basis.dense(), device=dev) if n <= 20_000_000 else None
