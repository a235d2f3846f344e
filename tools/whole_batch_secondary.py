#!/usr/bin/env python3
"""tools/whole_batch_secondary.py [names...] -- run ON THE GPU BOX: the `secondary` workloads bench.py times, compared WHOLE with the
literal oracle at their timed sizes (10^4 tracks x 10^4 rows unless said), value and gradient at the parameter vector of the first
timed step.  tests/test_gpu_whole_batch.py does this for the metric's batch, configs 2 and 3 and the row-varying batch inside the suite;
these five would add ~6 minutes of oracle time to it, so they are a tool whose output is kept under profiles/.  Uses oracle/ as the
CHECKER (tests/oracle_lib.py), like the tests.  One JSON line per workload.
    names: irregular missing missing_one absent argos  (default: these five);  c5_bm c5_ou c5_ctcrw: the three sub-batches of BASELINE config 5;
    report_headline report_irregular report_missing: ssde_report (aest_all) of a whole batch against the oracle's filtered states"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from smoothsde_amd import capi  # noqa: E402
from oracle_lib import keep_P_symmetric, oracle_eval  # noqa: E402

M, T = int(os.environ.get("WB_TRACKS", 10_000)), int(os.environ.get("WB_ROWS", 10_000))
THREADS = min(16, os.cpu_count() or 8)
dev = torch.device("cuda:0")


def irregular(ID, times, obs):
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    return ID, torch.cumsum(0.5 + torch.rand(len(ID), device=dev, dtype=torch.float64, generator=gen), 0), obs


def missing(ID, times, obs):
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    na = torch.rand(len(ID), device=dev, generator=gen) < 0.05
    na[::T] = False
    obs[na] = float("nan")
    return ID, times, obs


def missing_one(ID, times, obs):
    gen = torch.Generator(device=dev); gen.manual_seed(8)
    rows = torch.randint(1, T, (M,), device=dev, generator=gen) + T * torch.arange(M, device=dev)
    obs[rows] = float("nan")
    return ID, times, obs


def absent(ID, times, obs):
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    keep = torch.rand(len(ID), device=dev, generator=gen) >= 0.05
    keep[::T] = True
    return ID[keep].contiguous(), times[keep].contiguous(), obs[keep].contiguous()


def general(mutate):
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=11, device=dev)
    ID, times, obs = mutate(ID, times, obs.contiguous())
    fixed = np.zeros(5, dtype=np.uint8); fixed[1:3] = 1
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=fixed))
    host = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), par_fixed=fixed)
    return eng, host, bench.theta_for(5, 2, 4, 0)


def argos():
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(17)
    A = 0.05 * torch.randn(M * T, 2, 2, device=dev, dtype=torch.float64, generator=gen)
    Hn = A @ A.transpose(1, 2)
    Hn[:, 0, 0] += 0.0025
    Hn[:, 1, 1] += 0.0025
    H = Hn.permute(1, 2, 0)
    fixed = np.array([1, 1, 1, 0, 0], dtype=np.uint8)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs.contiguous(), par_fixed=fixed, H=H))
    host = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), par_fixed=fixed, H=np.ascontiguousarray(H.cpu().numpy()))
    return eng, host, np.ascontiguousarray(np.array([0.0, 0.0, 0.0, np.log(2.0), 0.0]) + 1e-3 * np.sin(np.arange(5)))


def c5(i):
    """BASELINE config 5 as bench.py --config c5 builds it on one GPU: sub-batch i of (BM_SSM, OU_SSM, CTCRW), 30000 ragged tracks of
    5000-10000 rows, 5 % of the rows missing (column 0 only / every column), every parameter free"""
    import argparse
    a = argparse.Namespace(config="c5", tracks=int(os.environ.get("WB_C5_TRACKS", bench.CONFIGS["c5"]["tracks"])), rows=T, scaling="strong", seed=1,
                           warmup=1, steps=1, model="CTCRW")
    hs = bench.build_handles(a, dev, 0, 1)
    for j, h in enumerate(hs):
        if j != i:
            h.eng.close()
    h = hs[i]
    pb = h.pb
    host = capi.Problem(h.model, pb._t_id.cpu().numpy(), pb._t_times.cpu().numpy(), pb._t_obs.t().contiguous().cpu().numpy())
    return h.eng, host, h.thetas[0]


def c3_table():
    """BASELINE config 3 with the design block evaluated by the lanes from its B-spline table (ssde_ppbasis); the oracle gets the
    n x 9 matrix the table stands for (PPBasis.dense: ~4 temporaries of n x 9 doubles on the host -- run it with WB_ROWS=1000)"""
    from smoothsde_amd.synth import bspline_ppbasis, second_difference_penalty, simulate
    ID, times, obs = simulate("OU", M, T, 1, mu=1.0, tau=2.0, kappa=1.0, seed=2, backend="torch", device=dev)
    x = torch.cumsum(torch.randn(len(ID), device=dev, dtype=torch.float64) * 0.01, 0)
    x = (x - x.min()) / (x.max() - x.min())
    basis = bspline_ppbasis(x, 9, centre=np.zeros(9))
    S = [second_difference_penalty(9)]
    eng = capi.Engine(capi.Problem.from_torch("OU", ID, times, obs, basis_re=[basis, None, None], S_list=S))
    host = capi.Problem("OU", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), X_re=[basis.dense(), None, None], S_list=S)
    return eng, host, np.concatenate([[1.0, np.log(2.0), 0.0], [0.0], 0.05 * np.sin(np.arange(9))])


def drift(model, d, K, na_frac=0.0, table=False):
    """tools/bench_drift.py's batches: mu a K-column smooth of a covariate, everything else constant -- iso_drift_kernel (regular grid,
    complete tracks), iso_drift_general_kernel (missing rows), the block as a table (k_iso_drift_pp.hip)"""
    from smoothsde_amd.synth import bspline_ppbasis, second_difference_penalty
    ID, times, obs = capi.simulate_device(model, M, T, d, mu=2.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, z0=2.0, seed=2, device=dev)
    n = M * T
    obs = obs.contiguous()
    if na_frac > 0:
        gen = torch.Generator(device=dev); gen.manual_seed(7)
        na = torch.rand(n, device=dev, generator=gen) < na_frac
        na[::T] = False
        obs[na] = float("nan")
    x = 0.5 + 0.45 * torch.sin(torch.arange(n, device=dev, dtype=torch.float64) * (2 * np.pi / 977.0))
    q = capi.n_sde_par(model, d)
    S = [second_difference_penalty(K)]
    if table:
        basis = [None] * q
        basis[0] = bspline_ppbasis(x, K, centre=np.zeros(K))
        pb = capi.Problem.from_torch(model, ID, times, obs, basis_re=basis, S_list=S)
        Xh = basis[0].dense()
    else:
        X = torch.stack([torch.cos(np.pi * k * x) for k in range(1, K + 1)], dim=1)
        X_re = [None] * q
        X_re[0] = X
        pb = capi.Problem.from_torch(model, ID, times, obs, X_re=X_re, S_list=S)
        Xh = X.cpu().numpy()
    Xr = [None] * q
    Xr[0] = Xh
    host = capi.Problem(model, ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), X_re=Xr, S_list=S)
    par0 = np.zeros(pb.n_par_full)
    par0[0] = np.log(0.1)
    par0[pb.off_fe] = 2.0
    par0[pb.off_fe + d] = np.log(2.0)
    par0[pb.off_re:] = 0.05 * np.random.default_rng(1).standard_normal(pb.n_re)
    return capi.Engine(pb), host, par0


WORK = {"drift_ou": lambda: drift("OU_SSM", 1, 9), "drift_ou_na": lambda: drift("OU_SSM", 1, 9, na_frac=0.02),
        "drift_ou_table": lambda: drift("OU_SSM", 1, 9, table=True), "drift_ctcrw": lambda: drift("CTCRW", 2, 9),
        "drift_bm_na": lambda: drift("BM_SSM", 2, 6, na_frac=0.02),
        "c3_table": c3_table, "c5_bm": lambda: c5(0), "c5_ou": lambda: c5(1), "c5_ctcrw": lambda: c5(2),
        "irregular": lambda: general(irregular), "missing": lambda: general(missing), "missing_one": lambda: general(missing_one),
        "absent": lambda: general(absent), "argos": argos}
def headline():
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=1, device=dev)
    fixed = np.zeros(5, dtype=np.uint8); fixed[1:3] = 1
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=fixed))
    host = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), par_fixed=fixed)
    return eng, host, bench.theta_for(5, 2, 4, 0)


def row_varying():
    ID, times, obs, B, S, fixed = bench.row_varying_batch(M, T, dev, 9)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, X_re=[None, None, B, B], S_list=[S, S], par_fixed=fixed))
    Bh = B.cpu().numpy()
    host = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), X_re=[None, None, Bh, Bh], S_list=[S, S], par_fixed=fixed)
    return eng, host, bench.row_varying_theta(0, 9)


WORK["headline"] = headline
WORK["row_varying"] = row_varying
REPORT = {"report_headline": headline, "report_irregular": lambda: general(irregular), "report_missing": lambda: general(missing)}
for name in [a for a in sys.argv[1:] if a in REPORT]:
    # REPORT(aest_all) (nllk_ctcrw.hpp:192-194, 246, 249) of a whole timed batch: ssde_report against the oracle's filtered states
    eng, host, theta = REPORT[name]()
    aest = eng.report(theta)
    inf = eng.info()
    eng.close()
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    _, _, oaest = oracle_eval(host, np.asarray(theta, dtype=float), order=1, threads=THREADS, report=True)
    secs = time.perf_counter() - t0
    sc = float(np.nanmax(np.abs(oaest)))
    err = float(np.nanmax(np.abs(aest - oaest)))
    same_nan = bool(np.array_equal(np.isnan(aest), np.isnan(oaest)))
    print(json.dumps({"workload": name, "rows": inf["n_rows"], "states": list(aest.shape), "max_abs_err": err, "scale": sc, "rel": err / sc, "same_nan_pattern": same_nan,
                      "ok": bool(err <= 1e-9 * sc and same_nan), "oracle_seconds": round(secs, 1)}), flush=True)
    del host, aest, oaest
for name in ([a for a in sys.argv[1:] if a not in REPORT] or ([] if sys.argv[1:] else [k for k in ('irregular', 'missing', 'missing_one', 'absent', 'argos')])):
    eng, host, theta = WORK[name]()
    val, grad = eng.eval(theta)
    inf = eng.info()
    eng.close()
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    extra = {}
    if name == "argos":
        # a coupling H_array on 10^4-row tracks: the literal recursion (P a full matrix, nllk_ctcrw.hpp:241) amplifies rounding there and
        # is percent-level noise; the comparison is with the restatement in ARBITER mode (P kept symmetric = binary128 = joint Gaussian
        # where those can be computed: tests/test_oracle_golden.py::test_reference_form_loses_...), the literal value printed beside it
        lit = oracle_eval(host, np.asarray(theta, dtype=float), order=0, threads=THREADS)
        keep_P_symmetric(True)
        extra = {"oracle_mode": "arbiter (P kept symmetric)", "literal_oracle_value": lit}
    oval, ograd = oracle_eval(host, np.asarray(theta, dtype=float), order=1, threads=THREADS)
    keep_P_symmetric(False)
    if extra:
        extra["literal_vs_arbiter_rel"] = abs(extra["literal_oracle_value"] - oval) / abs(oval)
    secs = time.perf_counter() - t0
    rel_v = abs(val - oval) / abs(oval)
    rel_g = float(np.max(np.abs(grad - ograd)) / np.max(np.abs(ograd)))
    print(json.dumps({"workload": name, "rows": inf["n_rows"], "kernel": capi.KERNEL_NAMES.get(inf["kernel_id"], "?"), "windows": inf["lanes_per_track"],
                      "window_check": inf["window_check"], "value": val, "oracle_value": oval, "value_rel": rel_v, "grad_rel_of_max": rel_g,
                      "ok": bool(rel_v <= 1e-10 and rel_g <= 1e-8 and inf["window_check"] <= capi.WINDOW_TOL),
                      "oracle_seconds": round(secs, 1), "oracle_threads": THREADS, **extra}), flush=True)
    del host
