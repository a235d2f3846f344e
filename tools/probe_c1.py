#!/usr/bin/env python3
"""tools/probe_c1.py [--reps N] -- C1 (one CTCRW track x 3672 rows, tau and nu smooth, 19 free parameters) evaluated through the
synchronous ssde_eval (hipGraph replay): median / min / p90 of the wall time per evaluation, the window plan, and -- one plain
evaluation with stamps -- the phases ssde_last_phase_ms reports.  Run under `rocprofv3 --kernel-trace --memory-copy-trace --stats`
for the per-node durations."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402,F401
from smoothsde_amd import capi  # noqa: E402
from bench_tv import problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=400)
ap.add_argument("--tracks", type=int, default=1)
ap.add_argument("--rows", type=int, default=3672)
ap.add_argument("--with-h", action="store_true")
a = ap.parse_args()
pb, par = problem(a.tracks, a.rows, with_H=a.with_h)
eng = capi.Engine(pb)
call = eng.bound_eval(order=1)
ths = [np.ascontiguousarray(par + 1e-3 * np.sin(k + np.arange(len(par)))) for k in range(a.reps + 4)]
for k in range(4):
    call(ths[k])
ts = []
for k in range(a.reps):
    t0 = time.perf_counter()
    call(ths[4 + k])
    ts.append(time.perf_counter() - t0)
ts = 1e3 * np.sort(np.array(ts))
inf = eng.info()
out = dict(tool="probe_c1", rows=a.tracks * a.rows, reps=a.reps, ms_median=float(np.median(ts)), ms_min=float(ts[0]), ms_p90=float(ts[int(0.9 * len(ts))]),
           window=inf["window"], chunks=inf["n_chunks"] if "n_chunks" in inf else None, workgroups=inf["n_kernel_blocks"], lanes_per_track=inf["lanes_per_track"],
           check=inf["window_check"], check_max=inf["window_check_max"], retries=inf["window_retries"], n_evals=inf["n_evals"],
           slow_calls=int(np.sum(ts > 1.5 * np.median(ts))), kernel=capi.KERNEL_NAMES.get(inf["kernel_id"]))
os.environ["SSDE_NO_GRAPH"] = "1"
eng.eval(par); eng.eval(par)
out["plain_kernel_ms"] = eng.info()["main_kernel_ms"]
try:
    ph = eng.last_phase_ms(); out["plain_phases_ms"] = ph if isinstance(ph, dict) else [float(v) for v in ph]
except Exception as e:  # noqa: BLE001
    out["plain_phases_ms"] = str(e)
print(json.dumps(out))
eng.close()
