#!/usr/bin/env python3
"""tools/bench_regroup.py [frac_tracks] -- 1e4 CTCRW tracks x 1e4 rows, regular grid; a fraction of the TRACKS (default 0.3,
scattered through the batch) has 5 % missing rows, the others have every row.  ms per evaluation with the tracks dealt to
wavefronts clean ones first (default) and in the caller's order (SSDE_NO_REGROUP=1)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
dev = torch.device("cuda:0")
M, T = 10_000, 10_000
ID, times, obs = simulate("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, backend="torch", device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
dirty_track = torch.rand(M, device=dev, generator=gen) < frac
na = (torch.rand(M * T, device=dev, generator=gen) < 0.05) & dirty_track.repeat_interleave(T)
na[::T] = False
obs[na] = float("nan")
par = np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0])
for label, env in (("clean tracks first", None), ("caller's order", "1")):
    if env:
        os.environ["SSDE_NO_REGROUP"] = env
    else:
        os.environ.pop("SSDE_NO_REGROUP", None)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
    for k in range(4):
        eng.eval(par + 1e-3 * k)
    reps = 10
    ths = [par + 1e-3 * np.sin(k + np.arange(5)) for k in range(reps)]
    t0 = time.perf_counter()
    for th in ths:
        v, g = eng.eval(th)
    wall = (time.perf_counter() - t0) / reps
    inf = eng.info()
    print(f"{label:20s} {frac:.2f} of the tracks with missing rows: ms/eval {1e3 * wall:.4f} rows/s {M * T / wall:.3e} windows {inf['lanes_per_track']} "
          f"check {inf['window_check']:.1e} value {v:.6f}", flush=True)
    eng.close()
