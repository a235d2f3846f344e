#!/usr/bin/env python3
"""tools/chunk_probe.py -- how the value and the gradient of one batch (2000 tracks x 1e4 rows, 5 % missing rows) depend
on the time-window plan: 1 window (sequential filter) against 6 / 12 / 32 windows, a longer warm-up, and without the
derived variance direction.  One engine per child process (the knobs are read at create).  Result: DESIGN.md section 4,
"Summation"."""
import os, sys, numpy as np, torch, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from smoothsde_amd import capi
from smoothsde_amd.synth import simulate
if len(sys.argv) > 1:
    model = sys.argv[1]
    M,T=2000,10000
    kw = dict(mu=[5.0,-5.0],tau=2.0,kappa=1.0,sigma_obs=0.1) if model=="OU_SSM" else dict(mu=0.1,sigma=1.0,sigma_obs=0.1) if model=="BM_SSM" else dict(mu=0.0,tau=2.0,nu=1.0,sigma_obs=0.1)
    ID,times,obs = simulate(model,M,T,2,seed=1,backend="torch",device="cuda:0",**kw)
    gen=torch.Generator(device=ID.device); gen.manual_seed(7)
    na = torch.rand(ID.numel(),device=ID.device,generator=gen) < 0.05; na[::T]=False; obs[na]=float("nan")
    par = {"OU_SSM":[np.log(0.1),5.0,-5.0,np.log(2.0),0.0],"BM_SSM":[np.log(0.1),0.1,0.1,0.0],"CTCRW":[np.log(0.1),0,0,np.log(2.0),0.0]}[model]
    e = capi.Engine(capi.Problem.from_torch(model,ID,times,obs))
    v,g = e.eval(np.array(par)); i=e.info()
    print(json.dumps(dict(v=v,g=list(g),lanes=i["lanes_per_track"],window=i["window"],chk=i["window_check"])))
    sys.exit(0)
for model in ("OU_SSM","BM_SSM","CTCRW"):
    ref=None
    for env in ({"SSDE_CHUNKS":"1"},{"SSDE_CHUNKS":"6"},{"SSDE_CHUNKS":"12"},{"SSDE_CHUNKS":"32"},{"SSDE_CHUNKS":"32","SSDE_NO_DERIVE":"1"},{"SSDE_CHUNKS":"32","SSDE_WINDOW":"128"}):
        out = subprocess.run([sys.executable, __file__, model], env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip().splitlines()[-1]
        j=json.loads(out); g=np.array(j["g"])
        if ref is None: ref=(j["v"],g)
        print(model, env, "lanes",j["lanes"],"W",j["window"],"chk %.1e"%j["chk"], "dv %.2e"%(abs(j["v"]-ref[0])/abs(ref[0])), "dg", np.array2string(np.abs(g-ref[1])/np.max(np.abs(ref[1])),precision=1), flush=True)
