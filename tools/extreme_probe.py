#!/usr/bin/env python3
"""Probe (GPU box): random problems of the fuzz suite evaluated with one coordinate pushed to an extreme or NaN;
prints every disagreement with the oracle in finiteness or value.  Progress goes to stdout line by line."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from smoothsde_amd import capi                      # noqa: E402
from oracle_lib import oracle_eval                  # noqa: E402
from test_gpu_fuzz import random_problem            # noqa: E402

bad = 0
t00 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 64):
    pb, par0 = random_problem(seed)
    eng = capi.Engine(pb)
    rng = np.random.default_rng(1000 + seed)
    free = np.flatnonzero(pb.par_fixed == 0)
    for trial in range(6):
        k = int(rng.choice(free))
        delta = [-100.0, -30.0, 30.0, 100.0, np.nan, -12.0][trial]
        par = par0.copy()
        par[k] = delta
        t0 = time.time()
        v, g = eng.eval(par)
        tg = time.time() - t0
        ov, og = oracle_eval(pb, par, order=1, threads=4)
        okv = (np.isfinite(v) == np.isfinite(ov)) and (not np.isfinite(ov) or abs(v - ov) <= 1e-9 * max(1, abs(ov)))
        gf = np.all(np.isfinite(g)) == np.all(np.isfinite(og))
        okg = gf and (not np.all(np.isfinite(og)) or np.max(np.abs(g - og)) <= 1e-7 * np.max(np.abs(og)) + 1e-9)
        if not (okv and okg) or tg > 2.0:
            bad += 1
            print("MISMATCH" if not (okv and okg) else "SLOW", seed, pb.model, pb.n, "par", k, delta, "gpu", v, "oracle", ov,
                  "path", eng.info()["path"], "t", round(tg, 2), "\n   g", g, "\n  og", og, flush=True)
    eng.close()
    print("seed", seed, pb.model, "done", round(time.time() - t00, 1), flush=True)
print("issues", bad)
