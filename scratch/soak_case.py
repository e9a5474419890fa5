"""One disagreement of scratch/soak.py looked at closely: soak_case.py <round> <model>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import oracle
from mcmc_amd.engine import DeviceProblem

r, i = int(sys.argv[1]), int(sys.argv[2])
B = 4096
rng = np.random.default_rng(1000 + r)
nl = int(rng.integers(1, 13))
P = int(rng.choice([5, 20, 40, 64, 100]))
fmin, fmax = sorted(rng.uniform(-0.3, 1.6, 2))
if fmax - fmin < 0.3:
    fmax = fmin + 0.3
periods = np.flip(1 / np.logspace(fmin, fmax, P)).copy()
vpvs = float(rng.uniform(1.5, 2.2))
d = np.sort(rng.uniform(0.001, 0.15, (B, nl)), axis=1)
v = rng.uniform(0.1, 2.0, (B, nl + 1))
if r % 3 == 0:
    v = np.sort(v, axis=1)
params = np.concatenate([d, v], axis=1)
m = params[i]
st, oc, _ = oracle.forward(m, nl, vpvs, periods)
prob = DeviceProblem(nl, vpvs, periods, search=1)
gv, gs = prob.forward(torch.tensor(m[None, :], device=prob.device))
gc = gv[0].cpu().numpy()
print("nl", nl, "P", P, "vpvs", vpvs)
print("model", repr(m.tolist()))
dif = np.where(gc != oc)[0]
print("periods that differ:", dif.tolist())
for k in dif[:6]:
    print("  k %d T %.6f gpu %.12f oracle %.12f rel diff %.3e" % (k, periods[k], gc[k], oc[k], gc[k] / oc[k] - 1))
k = int(dif[0])
# secular values around the two answers on both sides
lt = oracle.layer_table(m, nl, vpvs)
lo, hi = sorted((gc[k], oc[k]))
xs = np.linspace(lo - 3 * (hi - lo) - 1e-9, hi + 3 * (hi - lo) + 1e-9, 15)
dev = prob.device
gd = prob.secular(torch.tensor(m[None, :], device=dev), torch.zeros(len(xs), dtype=torch.int32, device=dev),
                  torch.full((len(xs),), k, dtype=torch.int32, device=dev), torch.tensor(xs, device=dev)).cpu().numpy()
for x, g in zip(xs, gd):
    print("  c %.12f oracle %+.3e gpu %+.3e" % (x, oracle.secular(*lt, periods[k], x), g))
print("previous period answers: gpu %.12f oracle %.12f" % (gc[k - 1], oc[k - 1]) if k else "")
