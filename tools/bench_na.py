#!/usr/bin/env python3
"""tools/bench_na.py [--per-track N] MODEL [MODEL ...] -- 1e4 x 1e4 tracks with 5 % missing rows (the general register
kernel), or with N missing rows in every track (quiet rows); one line per model: ms per evaluation and the
kernel's own time.  For A/B runs (SSDE_LIB / SSDE_CHUNKS / SSDE_NO_QUIET / SSDE_NO_NA_SORT)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
M, T = 10_000, 10_000
PAR = {"CTCRW": [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0], "OU_SSM": [np.log(0.1), 5.0, -5.0, np.log(2.0), 0.0],
       "BM_SSM": [np.log(0.1), 0.1, 0.1, 0.0]}
argv = sys.argv[1:]
per_track = 0
if argv and argv[0] == "--per-track":
    per_track = int(argv[1])
    argv = argv[2:]
for model in argv:
    ID, times, obs = simulate(model, M, T, 2, mu=[5.0, -5.0] if model == "OU_SSM" else 0.1 if model == "BM_SSM" else 0.0,
                              sigma=1.0, tau=2.0, nu=1.0, kappa=1.0, sigma_obs=0.1, seed=4, backend="torch", device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    if per_track > 0:
        rows = torch.randint(1, T, (M, per_track), device=dev, generator=gen) + T * torch.arange(M, device=dev)[:, None]
        obs[rows.reshape(-1)] = float("nan")
    else:
        na = torch.rand(len(ID), device=dev, generator=gen) < 0.05
        na[::T] = False
        obs[na] = float("nan")
    eng = capi.Engine(capi.Problem.from_torch(model, ID, times, obs))
    wall, inf = timed(eng, np.array(PAR[model]), 8)
    print(f"{model:7s} lib={os.path.basename(capi.lib_path())} chunks={os.environ.get('SSDE_CHUNKS', '-')} "
          f"ms/eval {1e3 * wall:.4f} kernel_ms {inf['main_kernel_ms']:.4f} windows {inf['lanes_per_track']} "
          f"warmup {inf['window']} check {inf['window_check']:.1e} quiet {inf['quiet_window']} share {inf['quiet_share']:.2f} "
          f"retries {inf['window_retries']}", flush=True)
    eng.close()
    del ID, times, obs
