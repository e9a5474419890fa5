import sys, numpy as np, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd.engine import DeviceProblem
rng=np.random.default_rng(106)
nl,P,B=6,40,1024
periods=np.flip(1/np.logspace(0.3,1.3,P)).copy()
d=np.sort(rng.uniform(0.001,0.15,(B,nl)),axis=1); v=rng.uniform(0.1,2.0,(B,nl+1))
params=torch.tensor(np.concatenate([d,v],axis=1),device='cuda')
prob=DeviceProblem(nl,1.75,periods)
ne=torch.zeros(2,dtype=torch.int64,device='cuda')
for i in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    vel,st=prob.forward(params,n_evals=ne)
    torch.cuda.synchronize(); print("forward ms",(time.perf_counter()-t0)*1e3, ne.tolist())
