#!/usr/bin/env python3
"""tools/bench_fused_finalize.py -- A/B of the finalising work fused into iso_shared_kernel (SSDE_FUSED_FINALIZE=1) against the dependent
iso_finalize_kernel launch (default), on the headline batch, one rank's share of it at N = 8 and BASELINE config 2; profiles/r05_fused_finalize_ab.txt."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from smoothsde_amd import capi
import bench
dev = torch.device("cuda:0")
res = {}
for M, T in ((10_000, 10_000), (1250, 10_000), (10_000, 1000)):
    ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, mu=0.0, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=1, track0=0, device=dev)
    pb = capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0])
    for mode in ("1", "0"):
        os.environ["SSDE_FUSED_FINALIZE"] = mode
        eng = capi.Engine(pb)
        ths = [bench.theta_for(5, 2, 4, k) for k in range(40)]
        for k in range(3):
            eng.eval(ths[k])
        f = eng.bound_eval(order=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(40):
            f(ths[k])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        v, g = eng.eval(ths[5])
        inf = eng.info()
        res[(M, T, mode)] = (v, g.copy(), inf["window_check"])
        print(M, T, "fused" if mode == "1" else "two launches", "ms/eval %.4f" % ms, "kernel", capi.KERNEL_NAMES[inf["kernel_id"]], "windows", inf["lanes_per_track"], "chk %.2e" % inf["window_check"], flush=True)
        eng.close()
    a, b = res[(M, T, "1")], res[(M, T, "0")]
    print("   bitwise equal:", a[0] == b[0] and np.array_equal(a[1], b[1]), " chk equal:", a[2] == b[2])
