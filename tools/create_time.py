import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from smoothsde_amd import capi
from smoothsde_amd.synth import simulate
dev = torch.device("cuda:0")
ID, times, obs = simulate("CTCRW", 10000, 10000, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=1, backend="torch", device=dev)
pb = capi.Problem.from_torch("CTCRW", ID, times, obs, par_fixed=[0, 1, 1, 0, 0])
e = capi.Engine(pb); e.close()
torch.cuda.synchronize()
for label, mut in (("regular, complete", None), ("5 % missing rows", "na"), ("5 % of the fixes absent", "drop")):
    I, t, o = ID, times, obs
    if mut == "na":
        o = obs.clone(); m = torch.rand(len(ID), device=dev) < 0.05; m[::10000] = False; o[m] = float("nan")
    if mut == "drop":
        k = torch.rand(len(ID), device=dev) >= 0.05; k[::10000] = True; I, t, o = ID[k].contiguous(), times[k].contiguous(), obs[k].contiguous()
    pb = capi.Problem.from_torch("CTCRW", I, t, o, par_fixed=[0, 1, 1, 0, 0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = capi.Engine(pb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"ssde_create, 1e8 device-resident rows, {label}: {dt:.3f} s", flush=True)
    e.close()
