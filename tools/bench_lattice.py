#!/usr/bin/env python3
"""tools/bench_lattice.py [drop] -- 1e4 CTCRW tracks on a regular schedule of 1e4 slots, each fix ABSENT from the data with
probability `drop` (default 0.05; not NA-padded): ms per evaluation with the lattice layout (default) and, for comparison,
through the irregular-grid kernel (SSDE_NO_LATTICE=1)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate  # noqa: E402

drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
dev = torch.device("cuda:0")
M, T = 10_000, 10_000
ID, times, obs = simulate("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, backend="torch", device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
keep = torch.rand(len(ID), device=dev, generator=gen) >= drop
keep[::T] = True                                   # every track keeps its first fix
ID, times, obs = ID[keep].contiguous(), times[keep].contiguous(), obs[keep].contiguous()
n = len(ID)
par = np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0])
for label, env in (("lattice layout", None), ("irregular-grid kernel", "1")):
    if env:
        os.environ["SSDE_NO_LATTICE"] = env
    else:
        os.environ.pop("SSDE_NO_LATTICE", None)
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
    print(f"{label:24s} rows {n} ({n / (M * T):.3f} of the slots) ms/eval {1e3 * wall:.4f} kernel_ms {inf['main_kernel_ms']:.4f} "
          f"rows/s {n / wall:.3e} uniform_dt {inf['uniform_dt']} windows {inf['lanes_per_track']} check {inf['window_check']:.1e} value {v:.6f}", flush=True)
    eng.close()
