"""The launch schedule's own numbers after each call of the bench workload: per-chain costs
(evaluation rounds per step with one candidate), accepted shares, what the last launches were
made of.  python scratch/sched_dump.py [chunk] [ncall] [warmup]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as Bn
from mcmc_amd.engine import DeviceChains, DeviceProblem
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ncall = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
for a, b in zip(edges[:-1], edges[1:]):
    if b == a:
        continue
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ch.steps(idx[a:b], cz[a:b], ua[a:b], want_flags=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    raw = ch.sched.cpu().numpy()
    cost = raw[:4 * B].view(np.float32)
    hdr = raw[4 * B:4 * B + 32].view(np.int32)
    cal = raw[4 * B + 24:4 * B + 32].view(np.float32)
    level = raw[4 * B + 32 + 8 * B:4 * B + 32 + 9 * B]
    pacc = (level >> 2) / 63.0
    c2 = 1 + np.maximum(pacc, 1 - pacc)
    t2 = cost / c2
    print("%d-%d: %.2f M/s | cost p1 %.1f p10 %.1f med %.1f p90 %.1f p99 %.1f max %.1f | t2 (rounds/step, two candidates) p10 %.1f med %.1f p99 %.1f max %.1f | pacc med %.2f p90 %.2f | n4 %d nsplit %d n4(first launch) %d left %d | ticks/round %.0f" % (
        a, b, B * (b - a) / dt / 1e6, *np.percentile(cost, [1, 10, 50, 90, 99, 100]), *np.percentile(t2, [10, 50, 99, 100]),
        *np.percentile(pacc, [50, 90]), hdr[0], hdr[1], hdr[4], hdr[5], cal[0] / max(cal[1], 1)))
