#!/usr/bin/env python3
"""tools/soak.py -- many evaluations in one process: device memory, host RSS and per-evaluation time must stay flat
(event / buffer leaks, plan thrash).  Covers the register, row-varying (hipGraph) and direct paths."""
import os
import sys
import time

import numpy as np
import psutil
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate, bspline_basis, second_difference_penalty  # noqa: E402

proc = psutil.Process()


def snapshot():
    free, _ = torch.cuda.mem_get_info(0)
    return free / 1e6, proc.memory_info().rss / 1e6


def soak(name, eng, par, n_eval, use_device=False):
    out = torch.zeros(2 + len(par), dtype=torch.float64, device="cuda:0")
    rng = np.random.default_rng(0)
    for k in range(50):
        eng.eval(par + 0.05 * rng.standard_normal(len(par)))
    f0, r0 = snapshot()
    t0 = time.perf_counter()
    first = None
    for k in range(n_eval):
        p = par + 0.05 * rng.standard_normal(len(par))       # plans (window counts) move with the parameters
        if use_device and k % 2:
            eng.eval_device(p, out.data_ptr())
            torch.cuda.synchronize()
        else:
            eng.eval(p)
        if k == n_eval // 10:
            first = (time.perf_counter() - t0) / (k + 1)
    dt = (time.perf_counter() - t0) / n_eval
    f1, r1 = snapshot()
    print(f"{name}: {n_eval} evaluations, {1e3 * dt:.3f} ms each (first tenth {1e3 * first:.3f}), device free {f0:.0f} -> {f1:.0f} MB, "
          f"host RSS {r0:.0f} -> {r1:.0f} MB", flush=True)
    assert abs(f1 - f0) < 64 and r1 - r0 < 64, "memory grew"


ID, times, obs = simulate("CTCRW", 2000, 2000, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, backend="torch", device="cuda:0")
eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
soak("register path (shared covariance)", eng, np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0]), 20000, use_device=True)
eng.close()
obs2 = obs.clone()
obs2[torch.rand(len(ID), device=obs.device) < 0.05] = float("nan")
obs2[::2000] = obs[::2000]
eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs2))
soak("register path (general kernel, missing rows)", eng, np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0]), 5000, use_device=True)
eng.close()
ID1, t1, o1 = simulate("CTCRW", 1, 3672, 2, tau=1.0, nu=1.0, sigma_obs=0.05, seed=342)
x = (np.sin(np.arange(3672) * 0.01) + 1) / 2
B = bspline_basis(x, 9)
pb = capi.Problem("CTCRW", ID1, t1, o1, X_re=[None, None, B, B], S_list=[second_difference_penalty(9)] * 2)
eng = capi.Engine(pb)
par = np.zeros(pb.n_par_full)
par[0] = np.log(0.05)
soak("row-varying path (hipGraph replay)", eng, par, 20000)
eng.close()
IDo, to, oo = simulate("OU", 500, 2000, 1, mu=1.0, tau=2.0, kappa=1.0, seed=2)
xo = (np.sin(np.arange(len(IDo)) * 0.003) + 1) / 2
pb = capi.Problem("OU", IDo, to, oo, X_re=[bspline_basis(xo, 9), None, None], S_list=[second_difference_penalty(9)])
eng = capi.Engine(pb)
par = np.concatenate([[1.0, np.log(2.0), 0.0], [0.0], np.zeros(9)])
soak("direct path", eng, par, 20000)
eng.close()
print("soak ok")
