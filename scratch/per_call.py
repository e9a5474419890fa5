import os, sys, time, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from mcmc_amd.engine import DeviceChains, DeviceProblem
periods,bounds=bench.problem_inputs()
prob0=DeviceProblem(6,1.75,periods)
tc,st=prob0.forward(torch.tensor(bench.true_model(6)[None,:],device='cuda'))
np.random.seed(0); obs=tc[0].cpu().numpy()+0.025*np.random.randn(40)
prob=DeviceProblem(6,1.75,periods,obs,"normal",{"std":0.025},bounds)
rng=np.random.default_rng(1000)
starts=bench.prior_starts(prob,bounds,1024,rng)
ch=DeviceChains(prob,starts,0.05); ch.init_likelihood()
drng=np.random.default_rng(0); S=int(sys.argv[1]) if len(sys.argv)>1 else 3200
idx=torch.tensor(drng.integers(0,13,(S,1024)).astype(np.int32),device='cuda')
cz=torch.tensor(np.tan(np.pi*(drng.uniform(size=(S,1024))-0.5)),device='cuda')
ua=torch.tensor(drng.uniform(size=(S,1024)),device='cuda')
a=0
for n in [200]+[250]*((S-200)//250):
    torch.cuda.synchronize(); t0=time.perf_counter()
    ch.steps(idx[a:a+n],cz[a:a+n],ua[a:a+n],want_flags=False)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    raw=ch.sched.cpu().numpy(); hdr=raw[4*1024:4*1024+32].view(np.int32)
    cost=raw[:4096].view(np.float32)/100.0; top=np.argsort(-cost)[:4]
    print("steps %5d-%5d: %.3f ms/step  median logL %.1f | first-launch n4 %d, unfinished at the second %d, its n4 %d | cost us/step median %.0f p99 %.0f top %s"%(a,a+n,dt*1e3/n, ch.logL.median().item(), hdr[4], hdr[5], hdr[0], np.median(cost), np.percentile(cost,99), [(int(i), int(cost[i])) for i in top]))
    a+=n
# the same calls again from the same start, queued without waiting in between
ch=DeviceChains(prob,starts,0.05); ch.init_likelihood()
a=0
ch.steps(idx[a:a+200],cz[a:a+200],ua[a:a+200],want_flags=False); a=200
torch.cuda.synchronize(); t0=time.perf_counter()
for n in [250]*8:
    ch.steps(idx[a:a+n],cz[a:a+n],ua[a:a+n],want_flags=False); a+=n
torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("queued: %.3f ms/step over 2000 steps" % (dt*1e3/2000))
ch=DeviceChains(prob,starts,0.05); ch.init_likelihood()
a=0
ch.steps(idx[a:a+200],cz[a:a+200],ua[a:a+200],want_flags=False); a=200
torch.cuda.synchronize(); t0=time.perf_counter()
for n in [250]*8:
    ch.steps(idx[a:a+n],cz[a:a+n],ua[a:a+n],want_flags=False); a+=n
    torch.cuda.synchronize()
dt=time.perf_counter()-t0
print("waiting after each call: %.3f ms/step over 2000 steps" % (dt*1e3/2000))
