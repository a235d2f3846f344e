#!/usr/bin/env python3
"""tools/tuning/probe_general.py [--model CTCRW] [--what irregular|missing] [--tracks M --rows T]: the general per-lane kernels
(k_iso.hip) on the bench's secondary workloads; SSDE_LIB / SSDE_CHUNKS choose the build and the number of windows.  One line."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="CTCRW")
ap.add_argument("--what", default="irregular")
ap.add_argument("--tracks", type=int, default=10_000)
ap.add_argument("--rows", type=int, default=10_000)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")


def irregular(ID, times, obs):
    g = torch.Generator(device=dev); g.manual_seed(5)
    return ID, torch.cumsum(torch.rand(times.numel(), generator=g, device=dev, dtype=torch.float64) + 0.5, 0), obs


def missing(ID, times, obs):
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    na = torch.rand(len(ID), device=dev, generator=gen) < 0.05
    na[::a.rows] = False
    obs[na] = float("nan")
    return ID, times, obs


r = bench.secondary_workload(a.what, a.model, a.tracks, a.rows, dev, a.steps, irregular if a.what == "irregular" else missing)
print(json.dumps({k: r[k] for k in ("workload", "ms_per_step", "kernel_ms", "frac_hbm", "frac_fp64_issue", "window_check_max", "window_retries", "groups")} |
                 {"lib": os.environ.get("SSDE_LIB", ""), "chunks": os.environ.get("SSDE_CHUNKS", "")}))
