#!/usr/bin/env python3
"""tools/bench_wide.py [d] -- the headline batch shape with a response of d columns (default 3): 1e4 CTCRW tracks x 1e4 rows,
regular grid, mu fixed.  A response wider than two columns runs as column pairs behind one handle: the
evaluation should cost what its parts cost one after the other."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
ID, times, obs = simulate("CTCRW", 10_000, 10_000, d, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, backend="torch", device=dev)
fixed = [0] + [1] * d + [0, 0]
eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=fixed))
par = np.array([np.log(0.1)] + [0.0] * d + [np.log(2.0), 0.0])
for k in range(5):
    eng.eval(par + 1e-3 * k)
n = 50
ths = [par + 1e-3 * np.sin(k + np.arange(len(par))) * (1 - np.array(fixed)) for k in range(n)]
t0 = time.perf_counter()
for th in ths:
    v, g = eng.eval(th)
wall = (time.perf_counter() - t0) / n
inf = eng.info()
print(f"d={d} ms/eval {1e3 * wall:.4f} kernels_ms {inf['main_kernel_ms']:.4f} rows/s {1e8 / wall:.3e} required B/row {inf['required_bytes_per_row']:.0f} "
      f"-> {inf['required_bytes_per_row'] * 1e8 / wall / 1e12:.2f} TB/s, check {inf['window_check']:.1e}", flush=True)
eng.close()
