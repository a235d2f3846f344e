import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from smoothsde_amd import capi
from smoothsde_amd.synth import simulate
M, T = 2000, 3000
ID, times, obs = simulate("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1)
rng = np.random.default_rng(0)
for k in range(0, M, 10):
    obs[k * T + rng.integers(1, T, size=20), :] = np.nan
keep = np.ones(len(ID), bool)
for k in range(5, M, 10):
    keep[k * T + rng.integers(2, T - 1, size=30)] = False
ID, times, obs = ID[keep], times[keep], obs[keep]
eng = capi.Engine(capi.Problem("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
par = np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0])
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
vals = []
for k in range(6000):
    v, g = eng.eval(par + 1e-3 * np.sin(k + np.arange(5)) * np.array([1, 0, 0, 1, 1]))
    if k % 1000 == 0:
        vals.append(v)
dt = (time.perf_counter() - t0) / 6000
inf = eng.info()
print(f"mixed batch soak: 6000 evaluations, {1e3 * dt:.4f} ms each, device free {free0 >> 20} -> {torch.cuda.mem_get_info()[0] >> 20} MB, "
      f"retries {inf['window_retries']}, check max {inf['window_check_max']:.1e}, uniform_dt {inf['uniform_dt']}")
v1, g1 = eng.eval(par); eng.forget(); v2, g2 = eng.eval(par)
assert v1 == v2 and np.array_equal(g1, g2), "evaluation is not deterministic"
eng.close()
print("ok")
