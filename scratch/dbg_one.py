import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_fullsize import _setup, NL, P, B
from mcmc_amd import _lib as L
from mcmc_amd.engine import DeviceChains, DeviceProblem
rng, periods, bounds, prob0, cand = _setup(12)
_, st = prob0.forward(cand)
starts = cand[st == 0][:B].cpu().numpy()
ONE = int(sys.argv[1]) if len(sys.argv) > 1 else -1
obs = prob0.forward(torch.tensor(starts[:1], device=prob0.device))[0][0].cpu().numpy() + 0.025 * rng.standard_normal(P)
prob = DeviceProblem(NL, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
dev, NP, S = prob.device, 2 * NL + 1, 1
idx = torch.tensor(rng.integers(0, NP, (120, B)).astype(np.int32), device=dev)[:1].contiguous()
cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(120, B)) - 0.5)), device=dev)[:1].contiguous()
u_acc = torch.zeros((1, B), dtype=torch.float64, device=dev)   # accept everything that evaluates
if ONE >= 0:
    starts = starts[ONE:ONE+1]; idx = idx[:, ONE:ONE+1].contiguous(); cauchy = cauchy[:, ONE:ONE+1].contiguous(); u_acc = u_acc[:, :1].contiguous()
res = []
L.check(L.lib().mhd_set_candidates(1))
for search in (2, 1):
    L.check(L.lib().mhd_set_search(search))
    ch = DeviceChains(prob, starts, 0.05); ch.sched = None
    ch.init_likelihood()
    p0 = ch.pred.cpu().numpy().copy()
    out = ch.steps(idx, cauchy, u_acc)
    res.append((out["accepted"].cpu().numpy()[0], out["status"].cpu().numpy()[0], ch.pred.cpu().numpy().copy()))
a, b = res
bad = np.where((a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]).any(1))[0]
print("differing chains:", len(bad), bad[:20])
for ib in bad[:6]:
    d = a[2][ib] - b[2][ib]
    ks = np.where(d != 0)[0]
    print(a[2][ib], b[2][ib]); print("chain", ib, "cz %.3f" % float(cauchy[0, ib]), "status", a[1][ib], b[1][ib], "acc", a[0][ib], b[0][ib], "periods differing", ks[:10], "diff", d[ks[:5]], "shift vs hint/dc", ((b[2][ib]-p0[ib])/0.005)[ks[:5]])
