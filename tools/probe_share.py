#!/usr/bin/env python3
"""tools/probe_share.py TRACKS ROWS [EVALS] -- one rank's share of a strong-scaled batch by itself: the engine's own account of the
layout (ssde_info), the evaluation and kernel times, and -- with SSDE_WAVE_CLOCK=FILE -- every wave's start / end / rows."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402

M, T = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda:0")
ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, track0=0, device=dev)
eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
del ID, times, obs
par0 = np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0])
ths = [np.ascontiguousarray(par0 + 1e-3 * np.sin(k + np.arange(5))) for k in range(K + 5)]
call = eng.bound_eval(order=1)
eng.set_option(capi.OPT_KERNEL_STAMPS, 0)
for k in range(5):
    call(ths[k])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    call(ths[5 + k])
wall = (time.perf_counter() - t0) / K
eng.set_option(capi.OPT_KERNEL_STAMPS, 1)
for k in range(min(64, K)):
    call(ths[k])
kms = [m for m in eng.kernel_ms_history(min(64, K)) if m > 0]
inf = eng.info()
eng.close()
print(json.dumps({"tracks": M, "rows": T, "ms_per_eval": 1e3 * wall, "kernel_ms": float(np.mean(kms)), "kernel_ms_min": float(np.min(kms)),
                  "info": {k: (v if not isinstance(v, (np.integer, np.floating)) else v.item()) for k, v in inf.items()}}))
