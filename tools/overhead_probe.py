import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, ctypes as C
from smoothsde_amd import capi
from smoothsde_amd.synth import simulate
ID, t, o = simulate("CTCRW", 200, 300, 2, seed=1)
pb = capi.Problem("CTCRW", ID, t, o, par_fixed=[0,1,1,0,0])
eng = capi.Engine(pb)
par = np.array([np.log(0.1),0,0,np.log(2.0),0.0])
eng.eval(par)
N=20000
t0=time.perf_counter()
for _ in range(N): eng.eval(par)
t1=time.perf_counter()
for _ in range(N): eng.last_kernel_ms()
t2=time.perf_counter()
val=C.c_double(); grad=np.zeros(5); pp=par.ctypes.data_as(C.POINTER(C.c_double)); gp=grad.ctypes.data_as(C.POINTER(C.c_double))
f=eng.lib.ssde_eval; h=eng._h
for _ in range(N): f(h, pp, 5, 1, C.byref(val), gp)
t3=time.perf_counter()
for _ in range(N): eng.info()
t4=time.perf_counter()
print(f"Engine.eval (memo hit) {1e6*(t1-t0)/N:.2f} us | last_kernel_ms {1e6*(t2-t1)/N:.2f} us | raw ctypes ssde_eval (memo hit) {1e6*(t3-t2)/N:.2f} us | info() {1e6*(t4-t3)/N:.2f} us")
