"""Duration of every call of the bench workload (200 warm-up steps, then calls of --chunk steps),
per-call throughput and acceptance; the library is the one MCMC_AMD_LIB names."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as Bn
from mcmc_amd.engine import DeviceChains, DeviceProblem
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ncall = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = int(sys.argv[3]) if len(sys.argv) > 3 else 200
nl = Bn.N_LAYERS
periods, bounds = Bn.problem_inputs(nl)
B, NP = Bn.CHAINS_PER_GPU, 2 * nl + 1
prob0 = DeviceProblem(nl, Bn.VPVS, periods)
true_curve, st = prob0.forward(torch.tensor(Bn.true_model(nl)[None, :], device=prob0.device))
np.random.seed(0)
obs = true_curve[0].cpu().numpy() + Bn.STD * np.random.randn(Bn.N_PERIODS)
prob = DeviceProblem(nl, Bn.VPVS, periods, obs, "normal", {"std": Bn.STD}, bounds)
starts = Bn.prior_starts(prob, bounds, B, np.random.default_rng(1000))
ch = DeviceChains(prob, starts, 0.05)
ch.init_likelihood()
drng = np.random.default_rng(0)
total = W + chunk * ncall
dev = prob.device
idx = torch.tensor(drng.integers(0, NP, (total, B)).astype(np.int32), device=dev)
cz = torch.tensor(np.tan(np.pi * (drng.uniform(size=(total, B)) - 0.5)), device=dev)
ua = torch.tensor(drng.uniform(size=(total, B)), device=dev)
edges = [0, W] + [W + chunk * (i + 1) for i in range(ncall)]
out = []
for a, b in zip(edges[:-1], edges[1:]):
    if b == a:
        continue
    acc0 = ch.counters[:, 0].sum().item()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ch.steps(idx[a:b], cz[a:b], ua[a:b], want_flags=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out.append("%d-%d: %.2f M/s acc %.2f" % (a, b, B * (b - a) / dt / 1e6, (ch.counters[:, 0].sum().item() - acc0) / (B * (b - a))))
print(os.path.basename(os.environ.get("MCMC_AMD_LIB", "default")), " | ".join(out))
