import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_fullsize import _setup, NL, P, B
from mcmc_amd import _lib as L
from mcmc_amd.engine import DeviceChains, DeviceProblem
rng, periods, bounds, prob0, cand = _setup(12)
_, st = prob0.forward(cand)
starts = cand[st == 0][:B].cpu().numpy()
obs = prob0.forward(torch.tensor(starts[:1], device=prob0.device))[0][0].cpu().numpy() + 0.025 * rng.standard_normal(P)
prob = DeviceProblem(NL, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
dev, NP, S = prob.device, 2 * NL + 1, 120
idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
res = []
for cands, search, sched in ((2, 2, True), (2, 2, False), (1, 2, False), (1, 1, False)):
    L.check(L.lib().mhd_set_candidates(cands)); L.check(L.lib().mhd_set_search(search))
    ch = DeviceChains(prob, starts, 0.05)
    if not sched: ch.sched = None
    ch.init_likelihood()
    acc, sts, hl = [], [], []
    for a in range(0, S, 40):
        out = ch.steps(idx[a:a+40], cauchy[a:a+40], u_acc[a:a+40], want_status=True) if 'want_status' in ch.steps.__code__.co_varnames else ch.steps(idx[a:a+40], cauchy[a:a+40], u_acc[a:a+40])
        acc.append(out["accepted"]); sts.append(out.get("status"))
    res.append((torch.cat(acc).cpu().numpy(), ch.logL.cpu().numpy().copy(), ch.pred.cpu().numpy().copy(), torch.cat(sts).cpu().numpy() if sts[0] is not None else None))
names = ["sched2", "plain2", "one/search2", "one/search1"]
for i in range(4):
    for j in range(i+1, 4):
        d = res[i][0] != res[j][0]
        ds = (res[i][3] != res[j][3]) if res[i][3] is not None else d
        print(names[i], names[j], "accept diffs", int(d.sum()), "status diffs", int(ds.sum()), "chains with logL diff", int((res[i][1] != res[j][1]).sum()))
ref = res[3]
for i in range(3):
    d = (res[i][0] != ref[0])
    if res[i][3] is not None: d |= (res[i][3] != ref[3])
    if d.any():
        s_, b_ = np.argwhere(d)[np.argmin(np.argwhere(d)[:, 0])]
        print(names[i], "first divergence step", s_, "chain", b_, "acc", res[i][0][s_, b_], ref[0][s_, b_], "status", None if ref[3] is None else (res[i][3][s_, b_], ref[3][s_, b_]), "cauchy", float(cauchy[s_, b_]))
