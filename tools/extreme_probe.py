import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from smoothsde_amd import capi
from smoothsde_amd.synth import simulate
from oracle_lib import oracle_eval
bad=0
for model in ["CTCRW","OU_SSM","BM_SSM","OU","BM","BM_t","CIR"]:
    d = 1 if model in ("BM_t","CIR") else 2
    sim = {"BM_t":"BM","CIR":"BM"}.get(model, model)
    for (M,T) in [(3,50),(200,400)]:
        ID,times,obs = simulate(sim, M, T, d, seed=5)
        if model=="CIR": obs = np.exp(0.2*obs % 2.0)
        kw = {"other_data":4.0} if model=="BM_t" else {}
        pb = capi.Problem(model, ID, times, obs, **kw)
        eng = capi.Engine(pb)
        p0 = np.zeros(pb.n_par_full)
        for k in range(pb.n_par_full):
            for delta in (-100,-30,-10,10,30,100,np.nan):
                par = p0.copy(); par[k] = delta
                v,g = eng.eval(par)
                ov,og = oracle_eval(pb, par, order=1, threads=4)
                okv = (np.isfinite(v)==np.isfinite(ov)) and (not np.isfinite(ov) or abs(v-ov) <= 1e-9*max(1,abs(ov)))
                gf = np.all(np.isfinite(g))==np.all(np.isfinite(og))
                okg = gf and (not np.all(np.isfinite(og)) or np.max(np.abs(g-og)) <= 1e-7*np.max(np.abs(og))+1e-9)
                if not (okv and okg):
                    bad+=1
                    print(model,M,T,"par",k,delta,"gpu",v,"oracle",ov, "path",eng.info()["path"], "g",g,"og",og)
        eng.close()
print("mismatches",bad)
