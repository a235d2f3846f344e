#!/usr/bin/env python3
"""tools/bench_configs.py -- nllk+gradient timings of the other BASELINE.json configurations on ONE MI355X
(the bench.py line is the 1e4 x 1e4 CTCRW case only).  Prints one JSON object per configuration.
    python tools/bench_configs.py [--quick]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from smoothsde_amd.synth import simulate, second_difference_penalty  # noqa: E402


def timed(eng, par, reps=10):
    eng.eval(par)
    eng.eval(par + 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        eng.eval(par + 1e-3 * np.sin(k + np.arange(len(par))))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    inf = eng.info()
    return wall, inf


def report(name, eng, par, rows, reps=10):
    wall, inf = timed(eng, par, reps)
    # fractions of peak are formed on the bytes the resident layout HAS TO READ (required_bytes_per_row: no `times` stream on a
    # regular grid, 8 B/row of covariate instead of a streamed block that is evaluated from its table), never on SURVEY
    # 8(d)'s algorithmic bytes, which can exceed the peak precisely because part of them is never moved
    bpr, rpr = inf["algo_bytes_per_row"], inf["required_bytes_per_row"]
    out = dict(config=name, rows=rows, ms_per_eval=1e3 * wall, track_timesteps_per_s=rows / wall,
               main_kernel_ms=inf["main_kernel_ms"], algo_bytes_per_row=bpr, required_bytes_per_row=rpr,
               algo_GBps=rows * bpr / wall / 1e9, required_GBps=rows * rpr / wall / 1e9, frac_of_8TBps=rows * rpr / wall / 8e12,
               path=capi.PATH_NAMES[inf["path"]], uniform_dt=inf["uniform_dt"],
               lanes_per_track=inf["lanes_per_track"], window=inf["window"], window_check=inf["window_check"],
               window_retries=inf["window_retries"], hbm_resident_GB=inf["hbm_bytes"] / 1e9)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M = 10_000
    T = 1_000 if a.quick else 10_000

    # C1: elephant-like single 2-D CTCRW track, 3672 hourly fixes, tau and nu smooth in a temperature covariate
    # (vignette model, smoothSDE.rmd:476-490): time-varying coefficients -> dense kernel, 19 free parameters
    from smoothsde_amd.synth import bspline_basis
    ID1, t1, o1 = simulate("CTCRW", 1, 3672, 2, tau=1.0, nu=1.0, sigma_obs=0.05, z0=[572.34, 1675.42], seed=342)
    temp = 30 + 10 * np.sin(np.arange(3672) * 2 * np.pi / 24) + np.random.default_rng(342).normal(0, 2, 3672)
    B = bspline_basis((temp - temp.min()) / (temp.max() - temp.min()), 9)
    S9 = second_difference_penalty(9)
    pb1 = capi.Problem("CTCRW", ID1, t1, o1, X_re=[None, None, B, B], S_list=[S9, S9],
                       par_fixed=np.r_[0, 1, 1, 0, 0, 1, 1, np.zeros(18)].astype(np.uint8))
    eng = capi.Engine(pb1)
    par1 = np.r_[np.log(0.05), 0, 0, 0, 0, 0, 0, 0.05 * np.sin(np.arange(18))]
    report("C1: single elephant-like CTCRW track x 3672, tau/nu splines (row-varying path, hipGraph replay)", eng, par1, 3672, 5)
    eng.close()

    # C2: 1e4 CTCRW tracks x 1e3 rows, constant coefficients, regular grid
    ID, times, obs = simulate("CTCRW", M, 1000, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, backend="torch", device=dev)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
    report("C2: 1e4 CTCRW x 1e3, regular grid", eng, np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0]), M * 1000)
    eng.close()

    # one-off cost of handing HOST arrays over the boundary (PCIe upload + re-tiling), same C2 batch
    IDh, th, oh = ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy()
    t0 = time.perf_counter()
    eng = capi.Engine(capi.Problem("CTCRW", IDh, th, oh, par_fixed=[0, 1, 1, 0, 0]))
    torch.cuda.synchronize()
    t_create = time.perf_counter() - t0
    eng.eval(np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0]))
    print(json.dumps(dict(config="C2 create from host arrays (PCIe upload + segment scan + re-tiling)", rows=M * 1000,
                          seconds=t_create, host_bytes=IDh.nbytes + th.nbytes + oh.nbytes,
                          effective_GBps=(th.nbytes + oh.nbytes + IDh.nbytes) / t_create / 1e9)), flush=True)
    eng.close()
    del IDh, th, oh

    # irregular time grid: general per-lane kernel (covariance half per lane, exp per row)
    ID, times, obs = simulate("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=2, backend="torch", device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    times = torch.cumsum(0.5 + torch.rand(len(ID), device=dev, dtype=torch.float64, generator=gen), 0)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
    report(f"CTCRW 1e4 x {T}, irregular grid (general kernel)", eng, np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0]), M * T, 5)
    eng.close()

    # C5 pieces: BM_SSM / OU_SSM / CTCRW, 5 % NA rows, regular grid (NaN-carrying groups -> general kernel)
    for model, par in (("BM_SSM", [np.log(0.1), 0.1, 0.1, 0.0]), ("OU_SSM", [np.log(0.1), 5.0, -5.0, np.log(2.0), 0.0]),
                       ("CTCRW", [np.log(0.1), 0.0, 0.0, np.log(2.0), 0.0])):
        ID, times, obs = simulate(model, M, T, 2, mu=[5.0, -5.0] if model == "OU_SSM" else 0.1 if model == "BM_SSM" else 0.0,
                                  tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=4, backend="torch", device=dev)
        gen = torch.Generator(device=dev); gen.manual_seed(7)
        na = torch.rand(len(ID), device=dev, generator=gen) < 0.05
        na[::T] = False
        obs[na] = float("nan")
        eng = capi.Engine(capi.Problem.from_torch(model, ID, times, obs))
        report(f"C5 piece: {model} 1e4 x {T}, 5% NA rows, all parameters free", eng, np.array(par), M * T, 5)
        eng.close()
        del ID, times, obs

    # C3: 1e4 OU tracks x T rows, mu = spline(x_t) with a 9-column streamed design block (88 B/row)
    ID, times, obs = simulate("OU", M, T, 1, mu=1.0, tau=2.0, kappa=1.0, seed=2, backend="torch", device=dev)
    n = len(ID)
    x = torch.cumsum(torch.randn(n, device=dev, dtype=torch.float64) * 0.01, 0)
    x = (x - x.min()) / (x.max() - x.min())
    B = torch.stack([torch.cos((k + 1) * np.pi * x) for k in range(9)], dim=1)   # cosine basis stand-in, built in HBM
    pb = capi.Problem.from_torch("OU", ID, times, obs, X_re=[B, None, None], S_list=[second_difference_penalty(9)])
    eng = capi.Engine(pb)
    par = np.concatenate([[1.0, np.log(2.0), 0.0], [0.0], 0.05 * np.sin(np.arange(9))])
    report(f"C3: 1e4 OU x {T}, 9 streamed design columns", eng, par, n, 5)
    eng.close()
    del B, pb

    # C3 with the smooth given as a FUNCTION of the covariate (ssde_ppbasis: cubic B-spline table, 9 columns): the
    # kernel reads x (8 B/row) and evaluates the block on the fly; algorithmic bytes stay those of the streamed contract
    from smoothsde_amd.synth import bspline_ppbasis
    basis = bspline_ppbasis(x, 9, centre=np.zeros(9))
    pb = capi.Problem.from_torch("OU", ID, times, obs, basis_re=[basis, None, None], S_list=[second_difference_penalty(9)])
    eng = capi.Engine(pb)
    report(f"C3 with the design block evaluated on the device from its B-spline table: 1e4 OU x {T}", eng, par, n, 5)
    eng.close()


if __name__ == "__main__":
    main()
