import os, sys, ctypes, json, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch; torch.zeros(1,device="cuda")
import mcmc_amd._lib as L
L.LIB_PATH=os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
lib=L.lib()
import bench
from mcmc_amd.engine import DeviceChains, DeviceProblem
periods,bounds=bench.problem_inputs()
prob0=DeviceProblem(6,1.75,periods)
tc,st=prob0.forward(torch.tensor(bench.TRUE_M[None,:],device='cuda'))
np.random.seed(0); obs=tc[0].cpu().numpy()+0.025*np.random.randn(40)
prob=DeviceProblem(6,1.75,periods,obs,"normal",{"std":0.025},bounds)
rng=np.random.default_rng(1000)
starts=bench.prior_starts(prob,bounds,1024,rng)
ch=DeviceChains(prob,starts,0.05); ch.init_likelihood()
drng=np.random.default_rng(0); S=300
idx=torch.tensor(drng.integers(0,13,(S,1024)).astype(np.int32),device='cuda')
cz=torch.tensor(np.tan(np.pi*(drng.uniform(size=(S,1024))-0.5)),device='cuda')
ua=torch.tensor(drng.uniform(size=(S,1024)),device='cuda')
ch.steps(idx[:200],cz[:200],ua[:200],want_flags=False)
lib.mhd_debug_blk(None,1)
m_before=ch.m.cpu().numpy().copy()
out=ch.steps(idx[200:],cz[200:],ua[200:])
bb=(ctypes.c_ulonglong*8192)(); lib.mhd_debug_blk(bb,0)
b=np.array(list(bb),dtype=np.float64).reshape(1024,8)
tot=b[:,:6].sum(axis=1)
order=np.argsort(tot)[::-1]
res=[]
for i in order[:6]:
    res.append({"chain":int(i),"Mcycles":(b[i,:6]/1e6).round(1).tolist(),"mismatches":int(b[i,6]),"m":m_before[i].tolist(),
                "idx":idx[200:,i].cpu().numpy().tolist(),"cz":cz[200:,i].cpu().numpy().tolist(),"ua":ua[200:,i].cpu().numpy().tolist(),
                "acc":out["accepted"][:,i].cpu().numpy().tolist(),"status":out["status"][:,i].cpu().numpy().tolist()})
    print(i, res[-1]["Mcycles"], res[-1]["mismatches"])
json.dump({"periods":periods.tolist(),"obs":obs.tolist(),"bounds":bounds.tolist(),"slow":res,"median_Mcycles":float(np.median(tot)/1e6)},open(os.path.join(ROOT,"gpurun_out","slow_chains.json"),"w"))
print("median", np.median(tot)/1e6)
