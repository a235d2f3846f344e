#!/usr/bin/env python3
"""tools/probe_coresidency.py -- do two waves of the shared-covariance kernel share a SIMD usefully?  Two engines, each on
1250 tracks x 10^4 rows (one rank's share of the metric's batch at N = 8: ~860 waves, one per SIMD), evaluated (a) one
after the other and (b) at the same time on two streams.  (b) ~ (a) / 2 means the second wave fills the first one's stalls."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from smoothsde_amd import capi  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
engs, outs, streams = [], [], []
for k in range(2):
    ID, times, obs = capi.simulate_device("CTCRW", M, 10_000, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, track0=k * M, device=dev)
    engs.append(capi.Engine(capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0])))
    engs[-1].set_option(capi.OPT_KERNEL_STAMPS, 0)
    outs.append(torch.zeros(7, dtype=torch.float64, device=dev))
    streams.append(torch.cuda.Stream(dev))
par0 = np.array([np.log(0.1), 0, 0, np.log(2.0), 0.0])
ths = [np.ascontiguousarray(par0 + 1e-3 * np.sin(k + np.arange(5))) for k in range(64)]


def run(concurrent, n=50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        for e, o, s in zip(engs, outs, streams):
            e.eval_device(ths[k % 64], o.data_ptr(), order=1, stream=(s if concurrent else streams[0]).cuda_stream)
        for s in streams:
            s.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for _ in range(2):
    run(True, 5); run(False, 5)
a = run(False)
b = run(True)
print(f"{M} tracks x 10^4 rows per engine, two engines: one after the other {a:.1f} us per pair, side by side on two streams {b:.1f} us per pair "
      f"(ratio {b / a:.2f})")
for e in engs:
    e.close()
