#!/usr/bin/env python3
"""tools/valu_workload.py NAME [STEPS] -- run ONE of the workloads bench.py's line names (the headline batch, one rank's share of
it, and every `secondary` entry) a few times and print what ran: {"name", "kernel_id", "kernel", "rows_in_launch", "kernel_ms"}.
tools/collect_valu.sh runs it under `rocprofv3 --pmc SQ_INSTS_VALU ...` per kernel family; tools/summarise_valu.py divides the
counter by rows_in_launch -> profiles/valu_per_row.json, the table behind `frac_fp64_issue` in the bench line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from smoothsde_amd import capi  # noqa: E402

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
M, T = 10_000, 10_000


def plain(tracks, rows, model="CTCRW"):
    ID, times, obs = capi.simulate_device(model, tracks, rows, 2, tau=2.0, nu=1.0, kappa=1.0, sigma=1.0, sigma_obs=0.1, seed=1, device=dev)
    q = capi.n_sde_par(model, 2)
    fixed = np.zeros(1 + q, dtype=np.uint8)
    fixed[1:3] = 1
    eng = capi.Engine(capi.Problem.from_torch(model, ID, times, obs, par_fixed=fixed))
    del ID, times, obs
    kms = []
    for k in range(steps + 2):
        eng.eval(bench.theta_for(1 + q, 2, q, k))
        kms.append(eng.last_kernel_ms())
    inf = eng.info()
    eng.close()
    return {"kernel_id": inf["kernel_id"], "rows_in_launch": inf["main_kernel_rows"], "kernel_ms": float(np.mean(kms[2:]))}


def irregular(ID, times, obs):
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    return ID, torch.cumsum(0.5 + torch.rand(len(ID), device=dev, dtype=torch.float64, generator=gen), 0), obs


def missing(ID, times, obs):
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    na = torch.rand(len(ID), device=dev, generator=gen) < 0.05
    na[::T] = False
    obs[na] = float("nan")
    return ID, times, obs


def missing_one(ID, times, obs):
    gen = torch.Generator(device=dev)
    gen.manual_seed(8)
    rows = torch.randint(1, T, (M,), device=dev, generator=gen) + T * torch.arange(M, device=dev)
    obs[rows] = float("nan")
    return ID, times, obs


if name == "headline":
    out = plain(M, T)
elif name == "share8":
    out = plain(M // 8, T)
elif name == "headline_ou":
    out = plain(M, T, "OU_SSM")
elif name == "c2":
    out = plain(M, 1000)
elif name in ("irregular", "missing", "missing_one"):
    r = bench.secondary_workload(name, "CTCRW", M, T, dev, steps, {"irregular": irregular, "missing": missing, "missing_one": missing_one}[name])
    out = {"kernel": r["kernel"], "rows_in_launch": r["rows_in_launch"], "kernel_ms": r["kernel_ms"]}
elif name == "row_varying":
    r = bench.row_varying_workload(M, T // 10, dev, steps)
    out = {"kernel": r["kernel"], "rows_in_launch": r["rows_in_launch"], "kernel_ms": r["kernel_ms"]}
elif name == "argos":
    r = bench.argos_workload(M, T, dev, steps)
    out = {"kernel": r["kernel"], "rows_in_launch": r["rows_in_launch"], "kernel_ms": r["kernel_ms"]}
else:
    raise SystemExit(f"unknown workload {name}")
if "kernel" not in out:
    out["kernel"] = capi.KERNEL_NAMES.get(out["kernel_id"], "?")
out["name"] = name
print(json.dumps(out))
