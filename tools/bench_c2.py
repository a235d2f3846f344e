#!/usr/bin/env python3
"""tools/bench_c2.py [rows_per_track] -- C2 (1e4 CTCRW tracks x 1e3 rows, regular grid): ms per synchronous evaluation,
kernel time, and (SSDE_TRACE=1) the host-side phases.  For latency work on the isotropic path."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
ID, times, obs = simulate("CTCRW", 10_000, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, backend="torch", device=dev)
eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
par = np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0])
for k in range(5):
    eng.eval(par + 1e-3 * k)
torch.cuda.synchronize()
n = 200
ths = [par + 1e-3 * np.sin(k + np.arange(5)) for k in range(n)]
t0 = time.perf_counter()
for th in ths:
    eng.eval(th)
wall = (time.perf_counter() - t0) / n
inf = eng.info()
print(f"T={T} ms/eval {1e3 * wall:.4f} kernel_ms {inf['main_kernel_ms']:.4f} windows {inf['lanes_per_track']} "
      f"rows/s {10_000 * T / wall:.3e} lib={os.path.basename(capi.lib_path())}", flush=True)
eng.close()
