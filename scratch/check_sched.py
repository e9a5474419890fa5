import os, sys, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from mcmc_amd.engine import DeviceChains, DeviceProblem
import mcmc_amd._lib as L
periods,bounds=bench.problem_inputs()
prob0=DeviceProblem(6,1.75,periods)
tc,st=prob0.forward(torch.tensor(bench.TRUE_M[None,:],device='cuda'))
np.random.seed(0); obs=tc[0].cpu().numpy()+0.025*np.random.randn(40)
prob=DeviceProblem(6,1.75,periods,obs,"normal",{"std":0.025},bounds)
rng=np.random.default_rng(1000)
B=1024
starts=bench.prior_starts(prob,bounds,B,rng)
drng=np.random.default_rng(0); S=300
idx=torch.tensor(drng.integers(0,13,(S,B)).astype(np.int32),device='cuda')
cz=torch.tensor(np.tan(np.pi*(drng.uniform(size=(S,B))-0.5)),device='cuda')
ua=torch.tensor(drng.uniform(size=(S,B)),device='cuda')
res=[]
for use_sched in (True, False):
    ch=DeviceChains(prob,starts,0.05); ch.init_likelihood()
    if not use_sched:
        ch.sched=None
    accs=[]
    for a in range(0,S,100):
        out=ch.steps(idx[a:a+100],cz[a:a+100],ua[a:a+100])
        accs.append(out["accepted"].cpu().numpy())
    torch.cuda.synchronize()
    if use_sched:
        n=B
        raw=ch.sched.cpu().numpy()
        cost=raw[:4*n].view(np.float32); n4=raw[4*n:4*n+4].view(np.int32)[0]; order=raw[4*n+16:4*n+16+4*n].view(np.int32)
        print("n4",n4,"nsplit",raw[4*n+4:4*n+8].view(np.int32)[0],"order is a permutation:", sorted(order.tolist())==list(range(n)), "cost min/median/max", cost.min(), np.median(cost), cost.max())
    res.append((ch.m.cpu().numpy().copy(), ch.logL.cpu().numpy().copy(), ch.counters.cpu().numpy().copy(), np.concatenate(accs), int(ch.n_evals[0])))
a,b=res
print("m equal", np.array_equal(a[0],b[0]), "logL equal", np.array_equal(a[1],b[1]), "counters equal", np.array_equal(a[2],b[2]), "accepted equal", np.array_equal(a[3],b[3]), "lit", a[4], b[4])
bad=np.where((a[2]!=b[2]).any(axis=1))[0]; print("chains differing:", bad[:20], len(bad))
if len(bad): print(a[2][bad[:5]], b[2][bad[:5]])
