import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from smoothsde_amd import capi
from oracle_lib import oracle_eval
dev = torch.device("cuda:0")
M, T = int(sys.argv[1]), int(sys.argv[2])
ID, times, obs = capi.simulate_device("CTCRW", M, T, 2, tau=2.0, nu=1.0, sigma_obs=0.1, seed=13, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(17)
A = 0.05 * torch.randn(M * T, 2, 2, device=dev, dtype=torch.float64, generator=gen)
Hn = A @ A.transpose(1, 2)
Hn[:, 0, 0] += 0.0025; Hn[:, 1, 1] += 0.0025
H = Hn.permute(1, 2, 0)
fixed = np.array([1, 1, 1, 0, 0], dtype=np.uint8)
theta = np.ascontiguousarray(np.array([0.0, 0.0, 0.0, np.log(2.0), 0.0]) + 1e-3 * np.sin(np.arange(5)))
Hh = np.ascontiguousarray(H.cpu().numpy())
host = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), par_fixed=fixed, H=Hh)
ov, og = oracle_eval(host, theta, order=1, threads=16)
for label, env in (("default", {}), ("chunks1", {"SSDE_CHUNKS": "1"}), ("no colvar", {"SSDE_NO_COLVAR": "1"})):
    for k in ("SSDE_CHUNKS", "SSDE_NO_COLVAR"): os.environ.pop(k, None)
    os.environ.update(env)
    for src in ("torch view", "torch contiguous", "host"):
        if src == "host": pb = host
        elif src == "torch view": pb = capi.Problem.from_torch("CTCRW", ID, times, obs.contiguous(), par_fixed=fixed, H=H)
        else: pb = capi.Problem.from_torch("CTCRW", ID, times, obs.contiguous(), par_fixed=fixed, H=H.contiguous())
        e = capi.Engine(pb); v, g = e.eval(theta); inf = e.info(); e.close()
        print(label, src, capi.KERNEL_NAMES.get(inf["kernel_id"]), "windows", inf["lanes_per_track"], "W", inf["window"], "value rel", abs(v - ov) / abs(ov), "grad rel", np.max(np.abs(g - og)) / np.max(np.abs(og)))
