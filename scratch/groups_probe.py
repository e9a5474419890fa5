"""Short calls as several concurrent launches: the chains of a GPU in G groups, each group its own
launch on its own stream (with fewer chains than wave slots a group's chains run four candidates,
and the groups' tails overlap).  usage: groups_probe.py G steps warmup chunk"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as Bn
from mcmc_amd.engine import DeviceChains, DeviceProblem

G, K, W, chunk = (int(a) for a in sys.argv[1:5])
B, nl = 1024, Bn.N_LAYERS
periods, bounds = Bn.problem_inputs(nl)
prob0 = DeviceProblem(nl, Bn.VPVS, periods)
true_curve, st = prob0.forward(torch.tensor(Bn.true_model(nl)[None, :], device=prob0.device))
np.random.seed(0)
obs = true_curve[0].cpu().numpy() + Bn.STD * np.random.randn(Bn.N_PERIODS)
prob = DeviceProblem(nl, Bn.VPVS, periods, obs, "normal", {"std": Bn.STD}, bounds)
starts = Bn.prior_starts(prob, bounds, B, np.random.default_rng(1000))
dev, NP, total = prob.device, 2 * nl + 1, W + K
drng = np.random.default_rng(0)
idx = drng.integers(0, NP, (total, B)).astype(np.int32)
cauchy = np.tan(np.pi * (drng.uniform(size=(total, B)) - 0.5))
u_acc = drng.uniform(size=(total, B))
groups = []
for g in range(G):
    sl = slice(g * B // G, (g + 1) * B // G)
    ch = DeviceChains(prob, starts[sl], 0.05)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    groups.append((ch, torch.tensor(np.ascontiguousarray(idx[:, sl]), device=dev),
                   torch.tensor(np.ascontiguousarray(cauchy[:, sl]), device=dev),
                   torch.tensor(np.ascontiguousarray(u_acc[:, sl]), device=dev),
                   torch.cuda.Stream(device=dev)))
torch.cuda.synchronize()

def run(lo, hi):
    for a in range(lo, hi, chunk):
        b = min(hi, a + chunk)
        for ch, i_, c_, u_, s_ in groups:
            with torch.cuda.stream(s_):
                ch.steps(i_[a:b], c_[a:b], u_[a:b], want_flags=False)
        # (a sweep over all groups would come here: every group's call has to be done)
        for *_, s_ in groups:
            s_.synchronize()

run(0, W)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(W, total)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("groups %d steps %d chunk %d: %.0f steps/s" % (G, K, chunk, B * K / dt))
