# runs the bench's chains for 750 steps, then one hard chain alone with the repair printing build
import os, sys, json, subprocess, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
stage = sys.argv[1]
import mcmc_amd._lib as L
if stage == "two": L.LIB_PATH = os.path.join(ROOT, "scratch", "libs", "dbg_repair.so")
import bench
from mcmc_amd.engine import DeviceChains, DeviceProblem
periods,bounds=bench.problem_inputs()
prob0=DeviceProblem(6,1.75,periods)
tc,st=prob0.forward(torch.tensor(bench.true_model(6)[None,:],device='cuda'))
np.random.seed(0); obs=tc[0].cpu().numpy()+0.025*np.random.randn(40)
prob=DeviceProblem(6,1.75,periods,obs,"normal",{"std":0.025},bounds)
if stage == "one":
    rng=np.random.default_rng(1000)
    starts=bench.prior_starts(prob,bounds,1024,rng)
    ch=DeviceChains(prob,starts,0.05); ch.init_likelihood()
    drng=np.random.default_rng(0); S=750
    idx=torch.tensor(drng.integers(0,13,(S,1024)).astype(np.int32),device='cuda')
    cz=torch.tensor(np.tan(np.pi*(drng.uniform(size=(S,1024))-0.5)),device='cuda')
    ua=torch.tensor(drng.uniform(size=(S,1024)),device='cuda')
    for a in range(0,S,250): ch.steps(idx[a:a+250],cz[a:a+250],ua[a:a+250],want_flags=False)
    cost=ch.sched.cpu().numpy()[:4096].view(np.float32)
    top=np.argsort(-cost)[:3]
    print("most expensive chains", top, cost[top])
    np.save(os.path.join(ROOT,"gpurun_out","hard_m.npy"), ch.m.cpu().numpy()[top])
else:
    m=np.load(os.path.join(ROOT,"gpurun_out","hard_m.npy"))[:1]
    print("model: depths", m[0,:6], "vs", m[0,6:])
    ch=DeviceChains(prob,m,0.05); ch.sched=None; ch.init_likelihood()
    print("curve", ch.pred.cpu().numpy()[0])
    drng=np.random.default_rng(5); S=40
    idx=torch.tensor(drng.integers(0,13,(S,1)).astype(np.int32),device='cuda')
    cz=torch.tensor(np.tan(np.pi*(drng.uniform(size=(S,1))-0.5)),device='cuda')
    ua=torch.ones((S,1),dtype=torch.float64,device='cuda')   # reject everything: the state stays
    L.check(L.lib().mhd_set_candidates(1))
    out=ch.steps(idx,cz,ua)
    torch.cuda.synchronize()
    print("status", out["status"].cpu().numpy()[:,0])
