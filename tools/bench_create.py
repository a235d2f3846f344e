#!/usr/bin/env python3
"""tools/bench_create.py -- ssde_create from HOST arrays (pageable numpy memory): upload + segment scan + re-tiling.
SSDE_NO_FAST_UPLOAD=1 switches the pipelined upload off (plain pageable hipMemcpy)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smoothsde_amd import capi  # noqa: E402

for M, T in ((10_000, 1_000), (10_000, 10_000)):
    n = M * T
    rng = np.random.default_rng(1)
    ID = np.repeat(np.arange(M, dtype=np.float64), T)
    times = np.arange(1.0, n + 1)
    obs = np.cumsum(rng.standard_normal((n, 2)), axis=0)
    obs = np.asfortranarray(obs)
    pb = capi.Problem("CTCRW", ID, times, obs, par_fixed=np.array([0, 1, 1, 0, 0], dtype=np.uint8))
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        eng = capi.Engine(pb)
        dt = time.perf_counter() - t0
        best = min(best, dt)
        v, g = eng.eval(np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0]))
        eng.close()
    host_bytes = ID.nbytes + times.nbytes + obs.nbytes
    print(f"create {M} x {T}: {1e3 * best:.1f} ms, {host_bytes / 1e6:.0f} MB of host arrays, {host_bytes / best / 1e9:.1f} GB/s effective, nllk {v:.6f}", flush=True)
