#!/usr/bin/env python3
"""tools/bench_colvar.py [--tracks M --rows T --k1 K --k2 K] -- nllk + gradient of a CTCRW batch with ROW-VARYING tau and nu
(splines of a covariate: the batch-scale form of BASELINE's C1) on ONE MI355X: the lane = track kernel with one filter
tangent per design column (k_iso_colvar.hip) against the lane = direction path (k_tv.hip, SSDE_NO_COLVAR=1) on the same
problem and parameters.  One JSON object per line."""
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
from smoothsde_amd.synth import second_difference_penalty, bspline_basis  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tracks", type=int, default=10_000)
ap.add_argument("--rows", type=int, default=1_000)
ap.add_argument("--k1", type=int, default=9)
ap.add_argument("--k2", type=int, default=9)
ap.add_argument("--evals", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--linear", action="store_true", help="tau ~ 1 + x as a fixed-effect slope instead of the splines (two streamed columns: iso_few_kernel)")
ap.add_argument("--irregular", action="store_true", help="irregular time grid (dt ~ U[0.5, 1.5] per row)")
ap.add_argument("--with-h", action="store_true", help="per-row 2 x 2 measurement covariances (H_array): the full-covariance lanes")
args = ap.parse_args()

M, T = args.tracks, args.rows
dev = torch.device("cuda:0")
ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, tau=1.0, nu=1.0, sigma_obs=0.05, seed=342, device=dev)
ID, times, obs = ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy()
if args.irregular:
    times = np.cumsum(np.random.default_rng(5).uniform(0.5, 1.5, len(times)))
n = M * T
temp = 30 + 10 * np.sin(np.arange(n) * 2 * np.pi / 24) + np.random.default_rng(342).normal(0, 2, n)
u = (temp - temp.min()) / (temp.max() - temp.min())
X_re, S = [None, None, None, None], []
X_fe = None
if args.linear:
    args.k1 = args.k2 = 0
    X_fe = [None, None, np.column_stack([np.ones(n), u]), None]
if args.k1:
    X_re[2] = bspline_basis(u, args.k1); S.append(second_difference_penalty(args.k1))
if args.k2:
    X_re[3] = bspline_basis(u, args.k2); S.append(second_difference_penalty(args.k2))
nre = args.k1 + args.k2
fixed = np.r_[0, 1, 1, 0, 0, np.ones(len(S)), np.zeros(nre)].astype(np.uint8)
if args.linear:
    fixed = np.array([0, 1, 1, 0, 0, 0], dtype=np.uint8)
H = None
if args.with_h:   # error ellipses of the size of the simulated measurement noise
    A = np.random.default_rng(343).standard_normal((n, 2, 2)) * 0.05
    H = np.einsum("nij,nkj->ikn", A, A) + 0.0025 * np.eye(2)[:, :, None]
    del A
pb = capi.Problem("CTCRW", ID, times, obs, X_fe=X_fe, X_re=X_re if S else None, S_list=S or None, par_fixed=fixed, H=H)
par = np.r_[np.log(0.05), 0, 0, 0, 0, np.zeros(len(S)), 0.05 * np.sin(np.arange(nre))]
if args.linear:
    par = np.array([np.log(0.05), 0, 0, 0.0, 0.3, 0.0])
bytes_row = 8.0 * (2 + nre + (4 if args.with_h else 0))
ref = None
for label, env in (("lane=track adjoint", {"SSDE_CV_ADJ": "2"}), ("lane=track tangents", {"SSDE_CV_ADJ": "0"}), ("lane=direction", {"SSDE_NO_COLVAR": "1"})):
    if args.only and args.only not in label:
        continue
    for k in ("SSDE_NO_COLVAR", "SSDE_CV_ADJ"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    eng = capi.Engine(pb)
    t_create = time.perf_counter() - t0
    reps = args.evals if "direction" not in label else max(3, args.evals // 4)
    v0, g0 = eng.eval(par)
    if ref is None:
        ref = (v0, g0)
    eng.eval(par + 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        eng.eval(par + 1e-3 * np.sin(k + np.arange(len(par))))
    wall = (time.perf_counter() - t0) / reps
    inf = eng.info()
    phases = {}
    for k in range(8):                                     # where an evaluation's time goes (median of 8 stamped evaluations)
        eng.eval(par + 1e-3 * np.cos(k + np.arange(len(par))))
        for name, ms in eng.last_phase_ms().items():
            phases.setdefault(name, []).append(ms)
    phases = {name: round(float(np.median(v)), 5) for name, v in phases.items()}
    print(json.dumps(dict(kernel=label, phases_ms=phases, engine_kernel=capi.KERNEL_NAMES.get(inf["kernel_id"], "?"), path=capi.PATH_NAMES[inf["path"]], tracks=M, rows=T,
                          columns=nre, ms_per_eval=1e3 * wall,
                          rows_per_s=n / wall, main_kernel_ms=inf["main_kernel_ms"], windows=inf["lanes_per_track"], warm_up=inf["window"],
                          window_check=inf["window_check"], retries=inf["window_retries"], create_s=t_create,
                          frac_of_8TBps_required=bytes_row * n / wall / 8e12, value=v0, grad_norm=float(np.linalg.norm(g0)),
                          value_rel_vs_first=abs(v0 - ref[0]) / max(1.0, abs(ref[0])),
                          grad_rel_vs_first=float(np.max(np.abs(g0 - ref[1])) / max(1e-300, np.max(np.abs(ref[1])))))), flush=True)
    eng.close()
for k in ("SSDE_NO_COLVAR", "SSDE_CV_ADJ"):
    os.environ.pop(k, None)
