#!/usr/bin/env python3
"""tools/bench_tv.py -- nllk+gradient timings of the row-varying-coefficient Kalman path (k_tv.hip) on ONE MI355X:
C1 (one elephant-like CTCRW track x 3672 rows, tau and nu smooth in a covariate, 19 free parameters) and a
multi-track batch, with and without a per-row H_array, each also on the dense kernel (SSDE_NO_TV=1) for comparison.  One JSON object per line.
    python tools/bench_tv.py [--tracks M --rows T]
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
from smoothsde_amd.synth import simulate, second_difference_penalty, bspline_basis  # noqa: E402


def timed(eng, par, reps):
    eng.eval(par)
    eng.eval(par + 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        eng.eval(par + 1e-3 * np.sin(k + np.arange(len(par))))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    # synchronous evaluations of this path replay a hipGraph (no per-kernel events): one plain evaluation for the
    # filter kernel's own duration
    os.environ["SSDE_NO_GRAPH"] = "1"
    eng.eval(par)
    inf = eng.info()
    os.environ.pop("SSDE_NO_GRAPH", None)
    return wall, inf


def problem(M, T, seed=342, with_H=False):
    ID, t, o = simulate("CTCRW", M, T, 2, tau=1.0, nu=1.0, sigma_obs=0.05, z0=[572.34, 1675.42], seed=seed)
    n = M * T
    temp = 30 + 10 * np.sin(np.arange(n) * 2 * np.pi / 24) + np.random.default_rng(seed).normal(0, 2, n)
    B = bspline_basis((temp - temp.min()) / (temp.max() - temp.min()), 9)
    S9 = second_difference_penalty(9)
    H = None
    if with_H:   # per-row 2 x 2 error ellipses (Argos-like), random but positive definite
        A = np.random.default_rng(seed + 1).standard_normal((n, 2, 2)) * 0.04
        H = np.einsum("nij,nkj->ikn", A, A) + 0.0015 * np.eye(2)[:, :, None]
    pb = capi.Problem("CTCRW", ID, t, o, X_re=[None, None, B, B], S_list=[S9, S9], H=H,
                      par_fixed=np.r_[0, 1, 1, 0, 0, 1, 1, np.zeros(18)].astype(np.uint8))
    par = np.r_[np.log(0.05), 0, 0, 0, 0, 0, 0, 0.05 * np.sin(np.arange(18))]
    return pb, par


def run(name, M, T, reps, with_H=False):
    pb, par = problem(M, T, with_H=with_H)
    for label, env in (("tv", None), ("dense", "1")):
        if env:
            os.environ["SSDE_NO_TV"] = env
        else:
            os.environ.pop("SSDE_NO_TV", None)
        eng = capi.Engine(pb)
        wall, inf = timed(eng, par, reps if label == "tv" else max(2, reps // 10))
        print(json.dumps(dict(config=name, kernel=label, path=capi.PATH_NAMES[inf["path"]], rows=M * T, ms_per_eval=1e3 * wall,
                              track_timesteps_per_s=M * T / wall, main_kernel_ms=inf["main_kernel_ms"],
                              workgroups=inf["n_kernel_blocks"], lanes_per_track=inf["lanes_per_track"],
                              window=inf["window"], window_check=inf["window_check"], retries=inf["window_retries"],
                              hbm_resident_MB=inf["hbm_bytes"] / 1e6)), flush=True)
        eng.close()
    os.environ.pop("SSDE_NO_TV", None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=1000)
    ap.add_argument("--rows", type=int, default=1000)
    a = ap.parse_args()
    run("C1: one elephant-like CTCRW track x 3672 rows, tau/nu splines, 19 free parameters", 1, 3672, 50)
    run(f"{a.tracks} CTCRW tracks x {a.rows} rows, tau/nu splines, 19 free parameters", a.tracks, a.rows, 10)
    run("C1 with per-row H_array (error ellipses): full-covariance lanes, 18 free parameters", 1, 3672, 50, with_H=True)
    run(f"{a.tracks} CTCRW tracks x {a.rows} rows with H_array, tau/nu splines", a.tracks, a.rows, 10, with_H=True)


if __name__ == "__main__":
    main()
