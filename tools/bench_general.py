#!/usr/bin/env python3
"""tools/bench_general.py -- the two configurations that run on the general register kernel (k_iso.hip): an irregular
time grid, and a regular grid with 5 % missing rows (C5's CTCRW piece).  For A/B runs of that kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate  # noqa: E402
from bench_configs import report  # noqa: E402

dev = torch.device("cuda:0")
M, T = 10_000, 10_000
ID, times, obs = simulate("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=2, backend="torch", device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(5)
t_irr = torch.cumsum(0.5 + torch.rand(len(ID), device=dev, dtype=torch.float64, generator=gen), 0)
eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, t_irr, obs, par_fixed=[0, 1, 1, 0, 0]))
report("CTCRW 1e4 x 1e4, irregular grid, mu fixed", eng, np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0]), M * T, 5)
eng.close()
del t_irr
for model, par in (("CTCRW", [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0]), ("OU_SSM", [np.log(0.1), 5.0, -5.0, np.log(2.0), 0.0]),
                   ("BM_SSM", [np.log(0.1), 0.1, 0.1, 0.0])):
    ID, times, obs = simulate(model, M, T, 2, mu=[5.0, -5.0] if model == "OU_SSM" else 0.1 if model == "BM_SSM" else 0.0, sigma=1.0, tau=2.0, nu=1.0, kappa=1.0,
                              sigma_obs=0.1, seed=4, backend="torch", device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    na = torch.rand(len(ID), device=dev, generator=gen) < 0.05
    na[::T] = False
    obs[na] = float("nan")
    eng = capi.Engine(capi.Problem.from_torch(model, ID, times, obs))
    report(f"{model} 1e4 x 1e4, 5% NA rows, all free", eng, np.array(par), M * T, 5)
    eng.close()
