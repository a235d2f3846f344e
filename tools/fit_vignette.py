#!/usr/bin/env python3
"""tools/fit_vignette.py -- the vignette's elephant model (smoothSDE.rmd:476-490) end to end on the GPU engine:
one 2-D CTCRW track of 3672 hourly fixes (synthetic stand-in), tau ~ s(temp, k = 10), nu ~ s(temp, k = 10),
mu fixed at 0, observation error estimated; $fit() with the smooths' coefficients integrated out (Laplace layer),
then the sdreport counterpart.  Prints wall times and evaluation counts."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smoothsde_amd.sde import SDE  # noqa: E402
from smoothsde_amd.synth import simulate  # noqa: E402

n = 3672
rng = np.random.default_rng(342)
temp = 30 + 10 * np.sin(np.arange(n) * 2 * np.pi / 24) + rng.normal(0, 2, n)
ID, times, obs = simulate("CTCRW", 1, n, 2, tau=1.0, nu=1.0, sigma_obs=0.05, z0=[572.34, 1675.42], seed=342)
data = dict(ID=ID, time=times, x=obs[:, 0], y=obs[:, 1], temp=temp)
t0 = time.perf_counter()
sde = SDE(formulas={"mu1": "~1", "mu2": "~1", "tau": "~ s(temp, k = 10, bs = \"ts\")", "nu": "~ s(temp, k = 10, bs = \"ts\")"}, data=data,
          type="CTCRW", response=["x", "y"], par0=[0, 0, 1, 1], fixpar=["mu1", "mu2"])
sde.setup()
t1 = time.perf_counter()
out = sde.fit(maxiter=200)
t2 = time.perf_counter()
rep = sde.report()
t3 = time.perf_counter()
obj = sde.tmb_obj()
print(f"setup {t1 - t0:.2f} s, fit {t2 - t1:.2f} s ({out['counts']} outer fn/gr calls, {sde.engine_.info()['n_evals']} GPU evaluations of "
      f"the joint nllk+gradient), report {t3 - t2:.2f} s; value {out['value']:.4f}, convergence {out['convergence']}")
print("tau(temp) range", np.round(np.percentile(sde.par()["tau"], [0, 50, 100]), 3), " nu(temp) range",
      np.round(np.percentile(sde.par()["nu"], [0, 50, 100]), 3), " sigma_obs", round(float(np.exp(sde.log_sigma_obs_)), 4),
      " lambda", np.round(sde.lambda_(), 3), " edf", round(sde.edf_conditional(), 2))
