import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mcmc_amd._lib as L
L.LIB_PATH = '/root/repo/scratch/libs/one_phase.so'
from test_gpu_fullsize import _setup, NL, P, B
from mcmc_amd.engine import DeviceChains, DeviceProblem
rng, periods, bounds, prob0, cand = _setup(12)
_, st = prob0.forward(cand)
starts = cand[st == 0][:B].cpu().numpy()
obs = prob0.forward(torch.tensor(starts[:1], device=prob0.device))[0][0].cpu().numpy() + 0.025 * rng.standard_normal(P)
prob = DeviceProblem(NL, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
dev, NP, S = prob.device, 2 * NL + 1, 100
ch = DeviceChains(prob, starts, 0.05); ch.init_likelihood()
for call in range(3):
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    torch.cuda.synchronize(); import time; t0 = time.time()
    ch.steps(idx, cauchy, u_acc); torch.cuda.synchronize(); dt = time.time() - t0
    raw = ch.sched.cpu().numpy()
    hdr = raw[4*B:4*B+32].view(np.int32); prog = raw[4*B+32+4*B:4*B+32+8*B].view(np.int32)
    cost = raw[:4*B].view(np.float32)
    print("call", call, "ms %.1f" % (dt*1e3), "header", hdr, "deadline", (int(hdr[2]) & 0xffffffff) | (int(hdr[3]) << 32), "progress min/median/max", prog.min(), np.median(prog), prog.max(), "cost median", np.median(cost))
import ctypes
tt = (ctypes.c_ulonglong * 4096)()
L.lib().mhd_debug_times(tt)
a = np.array(list(tt), dtype=np.uint64).reshape(2048, 2)[:1024]
dl = (int(hdr[2]) & 0xffffffff) | (int(hdr[3]) << 32)
print("kernel start ticks min/max", a[:, 0].min(), a[:, 0].max(), "end max", a[:, 1].max(), "deadline", dl, "deadline - start", dl - int(a[:, 0].min()))
