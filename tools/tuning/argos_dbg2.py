import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from smoothsde_amd import capi
from oracle_lib import oracle_eval, oracle_eval_quad
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
host = capi.Problem("CTCRW", ID.cpu().numpy(), times.cpu().numpy(), obs.cpu().numpy(), par_fixed=fixed, H=np.ascontiguousarray(H.cpu().numpy()))
ov = oracle_eval(host, theta, order=0, threads=16)
qv = oracle_eval_quad(host, theta, order=0)
e = capi.Engine(host); v, g = e.eval(theta); e.close()
print("engine", repr(v)); print("oracle double", repr(ov)); print("oracle quad", repr(qv))
print("engine - quad rel", abs(v - qv) / abs(qv), " double oracle - quad rel", abs(ov - qv) / abs(qv))
