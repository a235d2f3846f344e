#!/usr/bin/env python3
"""tools/sweep_dispatch.py [--quick] -- every threshold ssde_create dispatches on, with BOTH sides timed on several batch shapes
(VERDICT r03 #8).  For each case the same problem is evaluated (nllk + gradient, synchronous ssde_eval) with the engine's own choice and
with each side forced through the environment switch that exists for it; a threshold is WRONG on a shape when the engine's choice is
more than 10 % slower than the best forced side.  One line per (case, shape); a JSON summary at the end.

    drift      mu smooth (K = 9 columns), tau / nu / kappa constant: register lanes with streamed columns (k_iso_drift.hip, >= 32 tracks)
               against the lane = direction path (k_tv.hip)
    colvar     tau and nu smooth (2 x 9 columns): the eight-wave pipeline (k_iso_colvar.hip, >= max(160, 3400 / directions) tracks)
               against the lane = direction path
    few        tau ~ 1 + x (two streamed columns): iso_few_kernel (>= 1200 tracks) against the lane = direction path
    h_array    per-row 2 x 2 measurement covariances, constant tau / nu: iso_full_kernel (>= 64 tracks) against the full-covariance
               lane = direction lanes
    quiet      k missing rows in every track: quiet rows of the general kernel (share of quiet blocks >= 0.25) against plain general lanes
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
from smoothsde_amd.synth import bspline_basis, second_difference_penalty  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true", help="fewer shapes (a smoke run)")
ap.add_argument("--cases", default="drift,colvar,few,h_array,quiet")
args = ap.parse_args()
dev = torch.device("cuda:0")
SWITCHES = ("SSDE_NO_DRIFT", "SSDE_DRIFT_MIN_TRACKS", "SSDE_NO_COLVAR", "SSDE_NO_COLVAR_FULL", "SSDE_CV_NO_FEW", "SSDE_NO_QUIET", "SSDE_QUIET_ALWAYS")


def timed(pb, par, env, evals):
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = capi.Engine(pb)
    call = eng.bound_eval(order=1)
    ths = [np.ascontiguousarray(par + 1e-3 * np.sin(k + np.arange(len(par)))) for k in range(evals + 2)]
    call(ths[0]); call(ths[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(evals):
        call(ths[2 + k])
    ms = 1e3 * (time.perf_counter() - t0) / evals
    inf = eng.info()
    eng.close()
    for k in env:
        os.environ.pop(k, None)
    return ms, capi.KERNEL_NAMES.get(inf["kernel_id"], "?"), inf["window_check_max"]


def covariate(n):
    temp = 30 + 10 * np.sin(np.arange(n) * 2 * np.pi / 24) + np.random.default_rng(342).normal(0, 2, n)
    return (temp - temp.min()) / (temp.max() - temp.min())


def sim(model, M, T, d, **kw):
    ID, times, obs = capi.simulate_device(model, M, T, d, sigma_obs=0.05, seed=342, device=dev, **kw)
    return ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy()


def case_drift(M, T, model, d):
    ID, times, obs = sim(model, M, T, d, mu=2.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, z0=2.0)
    x = 0.5 + 0.45 * np.sin(np.arange(M * T) * (2 * np.pi / 977.0))
    X = np.stack([np.cos(np.pi * k * x) for k in range(1, 10)], axis=1)
    q = capi.n_sde_par(model, d)
    X_re = [None] * q
    X_re[0] = X
    pb = capi.Problem(model, ID, times, obs, X_re=X_re, S_list=[second_difference_penalty(9)])
    par = np.zeros(pb.n_par_full)
    par[0] = np.log(0.05); par[pb.off_fe] = 2.0; par[pb.off_fe + d] = np.log(2.0)
    par[pb.off_re:] = 0.05 * np.random.default_rng(1).standard_normal(pb.n_re)
    return pb, par, {"engine": {}, "lane=direction": {"SSDE_NO_DRIFT": "1"}, "register": {"SSDE_DRIFT_MIN_TRACKS": "1"}}


def case_colvar(M, T, model, d):
    ID, times, obs = sim(model, M, T, d, tau=1.0, nu=1.0, kappa=1.0, sigma=1.0)
    u = covariate(M * T)
    q = capi.n_sde_par(model, d)
    B = bspline_basis(u, 9)
    X_re = [None] * q
    S = []
    for j in range(d, q):
        X_re[j] = B; S.append(second_difference_penalty(9))
    fixed = np.r_[0, np.ones(d), np.zeros(q - d), np.ones(len(S)), np.zeros(9 * len(S))].astype(np.uint8)
    pb = capi.Problem(model, ID, times, obs, X_re=X_re, S_list=S, par_fixed=fixed)
    par = np.r_[np.log(0.05), np.zeros(q), np.zeros(len(S)), 0.05 * np.sin(np.arange(9 * len(S)))]
    return pb, par, {"engine": {}, "lane=direction": {"SSDE_NO_COLVAR": "1"}, "register": {"SSDE_DRIFT_MIN_TRACKS": "1"}}


def case_few(M, T, model, d):
    ID, times, obs = sim(model, M, T, d, tau=1.0, nu=1.0, kappa=1.0, sigma=1.0)
    u = covariate(M * T)
    q = capi.n_sde_par(model, d)
    X_fe = [None] * q
    X_fe[d] = np.column_stack([np.ones(M * T), u])
    fixed = np.r_[0, np.ones(d), np.zeros(q - d + 1)].astype(np.uint8)
    pb = capi.Problem(model, ID, times, obs, X_fe=X_fe, par_fixed=fixed)
    par = np.r_[np.log(0.05), np.zeros(d), 0.0, 0.3, np.zeros(q - d - 1)]
    return pb, par, {"engine": {}, "lane=direction": {"SSDE_NO_COLVAR": "1"}, "register": {"SSDE_DRIFT_MIN_TRACKS": "1"}}


def case_h(M, T, model, d):
    ID, times, obs = sim("CTCRW", M, T, 2, tau=2.0, nu=1.0)
    n = M * T
    A = np.random.default_rng(343).standard_normal((n, 2, 2)) * 0.05
    H = np.einsum("nij,nkj->ikn", A, A) + 0.0025 * np.eye(2)[:, :, None]
    pb = capi.Problem("CTCRW", ID, times, obs, par_fixed=[1, 1, 1, 0, 0], H=H)
    par = np.array([0.0, 0.0, 0.0, np.log(2.0), 0.0])
    return pb, par, {"engine": {}, "lane=direction": {"SSDE_NO_COLVAR_FULL": "1"}, "register": {"SSDE_DRIFT_MIN_TRACKS": "1"}}


def case_quiet(M, T, model, k):
    ID, times, obs = sim(model, M, T, 2, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0)
    rng = np.random.default_rng(8)
    rows = (rng.integers(1, T, size=(M, k)) + T * np.arange(M)[:, None]).ravel()
    obs[rows] = np.nan
    q = capi.n_sde_par(model, 2)
    fixed = np.r_[0, 1, 1, np.zeros(q - 2)].astype(np.uint8)
    pb = capi.Problem(model, ID, times, obs, par_fixed=fixed)
    par = np.r_[np.log(0.05), 0, 0, np.log(2.0), np.zeros(q - 3)]
    return pb, par, {"engine": {}, "general lanes": {"SSDE_NO_QUIET": "1"}, "quiet rows": {"SSDE_QUIET_ALWAYS": "1"}}


Q = args.quick
PLAN = {
    "drift": (case_drift, [(M, T, m, d) for m, d in (("OU_SSM", 1), ("CTCRW", 2)) for M, T in ([(16, 1000), (64, 1000)] if Q else [(16, 1000), (32, 1000), (64, 1000), (64, 10000), (1000, 1000), (10000, 1000)])]),
    "colvar": (case_colvar, [(M, T, m, d) for m, d in (("CTCRW", 2), ("OU_SSM", 1)) for M, T in ([(64, 1000), (256, 1000)] if Q else [(64, 1000), (128, 1000), (160, 1000), (256, 1000), (64, 10000), (1000, 1000), (10000, 1000)])]),
    "few": (case_few, [(M, T, m, d) for m, d in (("CTCRW", 2), ("OU_SSM", 1)) for M, T in ([(1000, 1000)] if Q else [(64, 1000), (1000, 1000), (1200, 1000), (2500, 1000), (10000, 1000), (1000, 10000)])]),
    "h_array": (case_h, [(M, T, "CTCRW", 2) for M, T in ([(64, 1000)] if Q else [(32, 1000), (64, 1000), (128, 1000), (1000, 1000), (64, 10000), (10000, 1000)])]),
    "quiet": (case_quiet, [(M, T, m, k) for m in ("CTCRW", "OU_SSM") for M, T in ([(1000, 1000)] if Q else [(1000, 1000), (1000, 10000), (10000, 1000), (10000, 10000)]) for k in ((1, 3) if Q else (1, 2, 3, 5))]),
}
out = []
for name in args.cases.split(","):
    build, shapes = PLAN[name]
    for shp in shapes:
        M, T = shp[0], shp[1]
        try:
            pb, par, sides = build(*shp)
        except Exception as e:  # noqa: BLE001
            print(f"{name} {shp}: could not build: {e}", flush=True)
            continue
        evals = max(3, min(20, int(2e8 / (M * T))))
        res = {}
        for label, env in sides.items():
            try:
                res[label] = timed(pb, par, env, evals)
            except Exception as e:  # noqa: BLE001
                res[label] = (float("nan"), f"failed: {str(e)[:80]}", 0.0)
        best = min((v[0], k) for k, v in res.items() if k != "engine" and v[0] == v[0])
        eng_ms, eng_kernel, chk = res["engine"]
        verdict = "ok" if eng_ms <= 1.10 * best[0] else f"WRONG by {100 * (eng_ms / best[0] - 1):.0f} % ({best[1]} is faster)"
        line = {"case": name, "shape": list(shp), "engine_ms": eng_ms, "engine_kernel": eng_kernel,
                "sides": {k: {"ms": v[0], "kernel": v[1]} for k, v in res.items() if k != "engine"}, "verdict": verdict, "check_max": chk}
        out.append(line)
        print(f"{name:8s} {str(shp):32s} engine {eng_ms:8.4f} ms on {eng_kernel:34s} | " +
              " | ".join(f"{k} {v[0]:8.4f} ({v[1]})" for k, v in res.items() if k != "engine") + f" | {verdict}", flush=True)
        del pb
wrong = [l for l in out if l["verdict"] != "ok"]
print(json.dumps({"tool": "sweep_dispatch", "cases": len(out), "wrong": len(wrong), "wrong_cases": [(l["case"], l["shape"], l["verdict"]) for l in wrong]}))
