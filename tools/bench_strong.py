#!/usr/bin/env python3
"""tools/bench_strong.py [--comm] [--rows T] [--tracks M] [--evals K] -- the strong-scaling curve of BASELINE's metric,
PREDICTED on one GPU: the fixed batch (M = 10^4 CTCRW tracks x T = 10^4 rows) split over N = 1, 2, 4, 8 ranks gives every
rank M / N whole tracks (tracks are independent: nllk_ctcrw.hpp:196-200, 234), so one rank's share is timed here for
every N -- synchronous ssde_eval, order 1, a fresh parameter vector per call -- and the curve is
    efficiency(N) = t(1) / (N * t(N)).
With --comm every share runs behind a ONE-rank RCCL communicator (ssde_comm_init_rank), so the all-reduce launch is on
the path; the wire latency of a real N-rank all-reduce over xGMI is not (no multi-GPU box in this pool).
Prints one line per N and a JSON summary."""
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

ap = argparse.ArgumentParser()
ap.add_argument("--comm", action="store_true")
ap.add_argument("--rows", type=int, default=10_000)
ap.add_argument("--tracks", type=int, default=10_000)
ap.add_argument("--evals", type=int, default=200)
ap.add_argument("--ranks", default="1,2,4,8")
args = ap.parse_args()

dev = torch.device("cuda:0")
par0 = np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0])
res = []
for N in [int(x) for x in args.ranks.split(",")]:
    M = args.tracks // N
    # rank 0's shard of the one batch (ssde_simulate: the numbers are a function of seed / global track / row)
    ID, times, obs = capi.simulate_device("CTCRW", M, args.rows, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, track0=0, device=dev)
    eng = capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0]))
    del ID, times, obs
    if args.comm:
        eng.comm_init(1, 0, capi.comm_unique_id())
    call = eng.bound_eval(order=1)
    ths = [np.ascontiguousarray(par0 + 1e-3 * np.sin(k + np.arange(5))) for k in range(args.evals + 5)]
    eng.set_option(capi.OPT_KERNEL_STAMPS, 0)             # the timed evaluations: plain launches (what a fitting host runs)
    for k in range(5):
        call(ths[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.evals):
        call(ths[5 + k])
    wall = (time.perf_counter() - t0) / args.evals
    eng.set_option(capi.OPT_KERNEL_STAMPS, 1)             # the kernel's own duration: a second pass with stamped launches
    for k in range(min(64, args.evals)):
        call(ths[k])
    kms = [m for m in eng.kernel_ms_history(min(64, args.evals)) if m > 0]
    inf = eng.info()
    eng.close()
    res.append({"ranks": N, "tracks_per_rank": M, "rows": args.rows, "ms_per_eval": 1e3 * wall,
                "kernel_ms": float(np.mean(kms)) if kms else None, "windows": inf["lanes_per_track"], "window_rows": inf["window"],
                "check_max": inf["window_check_max"], "retries": inf["window_retries"]})
t1 = res[0]["ms_per_eval"] * res[0]["ranks"]
for r in res:
    r["predicted_efficiency"] = t1 / (r["ranks"] * r["ms_per_eval"])
    r["predicted_rows_per_s"] = args.tracks * args.rows / (r["ms_per_eval"] * 1e-3)
    k = r["kernel_ms"]
    print(f"N={r['ranks']}: {r['tracks_per_rank']} tracks x {r['rows']} rows per rank: {r['ms_per_eval']:.4f} ms/eval "
          f"(kernel {k if k is None else round(k, 4)}, outside the kernel {'' if k is None else round(1e3 * (r['ms_per_eval'] - k), 1)} us, "
          f"{r['windows']} windows, warm-up {r['window_rows']}), predicted efficiency {r['predicted_efficiency']:.3f}, "
          f"{r['predicted_rows_per_s']:.3e} rows/s", flush=True)
print(json.dumps({"tool": "bench_strong", "comm": bool(args.comm), "lib": os.path.basename(capi.lib_path()), "shares": res}))
