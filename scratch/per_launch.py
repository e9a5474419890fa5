import os, sys, time, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from mcmc_amd.engine import DeviceChains, DeviceProblem
from mcmc_amd import _lib as L
import ctypes
L.LIB_PATH=os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
search=int(sys.argv[1]) if len(sys.argv)>1 else 2
L.check(L.lib().mhd_set_search(search))
periods,bounds=bench.problem_inputs()
prob0=DeviceProblem(6,1.75,periods)
tc,st=prob0.forward(torch.tensor(bench.TRUE_M[None,:],device='cuda'))
np.random.seed(0); obs=tc[0].cpu().numpy()+0.025*np.random.randn(40)
prob=DeviceProblem(6,1.75,periods,obs,"normal",{"std":0.025},bounds)
rng=np.random.default_rng(1000)
starts=bench.prior_starts(prob,bounds,1024,rng)
ch=DeviceChains(prob,starts,0.05); ch.init_likelihood()
drng=np.random.default_rng(0); S=1000
lib=L.lib()
idx=torch.tensor(drng.integers(0,13,(S,1024)).astype(np.int32),device='cuda')
cz=torch.tensor(np.tan(np.pi*(drng.uniform(size=(S,1024))-0.5)),device='cuda')
ua=torch.tensor(drng.uniform(size=(S,1024)),device='cuda')
for a in range(0,S,250):
    lib.mhd_debug_blk(None,1)
    torch.cuda.synchronize(); t0=time.perf_counter()
    ch.steps(idx[a:a+250],cz[a:a+250],ua[a:a+250],want_flags=False)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("steps %5d-%5d: %.3f ms/step  median logL %.1f  p10 logL %.1f"%(a,a+250,dt*1e3/250, ch.logL.median().item(), ch.logL.kthvalue(102).values.item()))
    tt=(ctypes.c_ulonglong*4096)(); lib.mhd_debug_times(tt)
    arr=np.array(list(tt),dtype=np.float64).reshape(2048,2)[:1024]; dur=(arr[:,1]-arr[:,0])/100.0/250
    bb=(ctypes.c_ulonglong*8192)(); lib.mhd_debug_blk(bb,0); b=np.array(list(bb),dtype=np.float64).reshape(1024,8)
    print("   per-chain us/step: p10 %.0f median %.0f p90 %.0f p99 %.0f max %.0f | chains with >50%% v1 steps: %d, any v1 steps: %d | mismatches/step median %.2f p99 %.2f"%(np.percentile(dur,10),np.median(dur),np.percentile(dur,90),np.percentile(dur,99),dur.max(),(b[:,7]>125).sum(),(b[:,7]>0).sum(),np.median(b[:,6])/250,np.percentile(b[:,6],99)/250))
