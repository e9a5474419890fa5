import os, sys, ctypes, numpy as np, torch, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mcmc_amd._lib as L
L.LIB_PATH=os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
from mcmc_amd.engine import DeviceProblem
lib=L.lib()
rng=np.random.default_rng(106)
nl,P,B=6,40,1024
periods=np.flip(1/np.logspace(0.3,1.3,P)).copy()
d=np.sort(rng.uniform(0.001,0.15,(B,nl)),axis=1); v=rng.uniform(0.1,2.0,(B,nl+1))
params=torch.tensor(np.concatenate([d,v],axis=1),device='cuda')
prob=DeviceProblem(nl,1.75,periods)
ne=torch.zeros(2,dtype=torch.int64,device='cuda')
out=(ctypes.c_ulonglong*16)()
for search in (2,1):
    lib.mhd_set_search(search)
    for i in range(3):
        lib.mhd_debug_profile(out,1); ne.zero_()
        torch.cuda.synchronize(); t0=time.perf_counter()
        vel,st=prob.forward(params,n_evals=ne)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)*1e3
        lib.mhd_debug_profile(out,0)
        o=list(out)
        print("search",search,"ms %.3f"%dt, ne.tolist(), "cycles/wave: coarse %.0f refine %.0f replay %.0f verify %.0f fallback %.0f giveup_v1 %.0f | mismatches %d (periods redone %d) clean %d"%tuple([o[i]/B for i in range(6)]+[o[6],o[7],o[8]]))
