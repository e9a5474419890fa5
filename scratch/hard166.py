import os, sys, time, ctypes, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mcmc_amd._lib as L
L.LIB_PATH=os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
import bench
from mcmc_amd.engine import DeviceChains, DeviceProblem
lib=L.lib()
periods,bounds=bench.problem_inputs()
prob0=DeviceProblem(6,1.75,periods)
tc,st=prob0.forward(torch.tensor(bench.true_model(6)[None,:],device='cuda'))
np.random.seed(0); obs=tc[0].cpu().numpy()+0.025*np.random.randn(40)
prob=DeviceProblem(6,1.75,periods,obs,"normal",{"std":0.025},bounds)
rng=np.random.default_rng(1000)
starts=bench.prior_starts(prob,bounds,1024,rng)
ch=DeviceChains(prob,starts,0.05); ch.init_likelihood()
drng=np.random.default_rng(0); S=1200
idx=torch.tensor(drng.integers(0,13,(S,1024)).astype(np.int32),device='cuda')
cz=torch.tensor(np.tan(np.pi*(drng.uniform(size=(S,1024))-0.5)),device='cuda')
ua=torch.tensor(drng.uniform(size=(S,1024)),device='cuda')
a=0
for n in [200]+[250]*4:
    ch.steps(idx[a:a+n],cz[a:a+n],ua[a:a+n],want_flags=False); a+=n
torch.cuda.synchronize()
gc = (ctypes.c_uint * 32768)(); lib.mhd_debug_chain(gc)
ev = np.array(list(gc), dtype=np.float64).reshape(1024, 32)
names = {0:"curves",1:"hint tried",2:"hint gave up",4:"repairs",5:"predicted again",6:"same root",7:"not plain",8:"same answer",10:"finished sequentially",26:"resynchronised",11:"no bracket",12:"long scan",13:"floor",16:"above",17:"below"}
bb=(ctypes.c_ulonglong*8192)(); lib.mhd_debug_blk(bb,0); blk=np.array(list(bb),dtype=np.float64).reshape(1024,8)
for i in (166, 461, 500):
    n=max(ev[i,0],1)
    print("chain", i, "m", np.round(ch.m[i].cpu().numpy(),4))
    print("   curve", np.round(ch.pred[i].cpu().numpy(),3))
    print("   ", ", ".join("%s %.2f"%(v, ev[i,k]/n) for k,v in names.items() if k), "| curves", int(ev[i,0]), "| kcycles/curve by phase", [int(blk[i,j]/n/1000) for j in range(5)])
