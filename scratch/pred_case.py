"""Configuration 123 of scratch/soak_sched.py: the curves every step leaves behind against the forward
entry point on the parameters it leaves behind (one launch per call), step by step."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mcmc_amd.engine import DeviceChains, DeviceProblem
import mcmc_amd.engine as E
r = 123
rng = np.random.default_rng(9000 + r)
nl = int(rng.integers(2, 8)); NP = 2 * nl + 1
P = int(rng.choice([8, 16, 24, 40]))
B = int(rng.choice([64, 65, 67, 96, 129, 200, 255, 256, 383, 512, 700, 1024, 1100]))
itype = str(rng.choice(["phase", "phase", "group", "phase+ellipticity"]))
rotate = int(rng.choice([0, 0, 3]))
calls = [int(x) for x in rng.choice([3, 7, 12, 20, 30, 47, 48, 60, 90, 130], size=int(rng.integers(3, 6)))]
periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
prob0 = DeviceProblem(nl, 1.75, periods, itype=itype)
starts = np.empty((0, NP))
while starts.shape[0] < B:
    cand = rng.uniform(bounds[:, 0], bounds[:, 1], size=(2 * B, NP))
    order = np.argsort(cand[:, :nl], axis=1, kind="stable")
    cand[:, :nl] = np.take_along_axis(cand[:, :nl], order, axis=1)
    cand[:, nl:2 * nl] = np.take_along_axis(cand[:, nl:2 * nl], order, axis=1)
    _, st = prob0.forward(torch.tensor(cand, device=prob0.device))
    starts = np.concatenate([starts, cand[st.cpu().numpy() == 0]])
starts = starts[:B]
true = prob0.forward(torch.tensor(starts[:1], device=prob0.device))[0][0].cpu().numpy()
obs = true + 0.03 * rng.standard_normal(true.shape[0])
prob = DeviceProblem(nl, 1.75, periods, obs, "normal", {"std": 0.03}, bounds, itype=itype)
dev = prob.device
beta = rng.uniform(0.05, 1.0, B) if rng.uniform() < 0.4 else None
S = sum(calls)
idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
cz = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
ua = torch.tensor(rng.uniform(size=(S, B)), device=dev)
print("cfg", nl, P, B, itype, rotate, calls, beta is not None)

# Every step's predicted curves against the forward entry point on the same parameters.
for search, cand in ((0, 0),):
    E.set_default_tuning(search=search, candidates=cand)
    ch = DeviceChains(prob, starts, 0.05)
    ch.sched = None
    ch.init_likelihood()
    a = 0
    for c, n in enumerate(calls):
        o = ch.steps(idx[a:a + n], cz[a:a + n], ua[a:a + n], hist_every=1, want_pred_hist=True)
        a += n
        hm = o["hist_m"].reshape(n * B, NP)
        E.set_default_tuning(search=0, candidates=0)
        pr, st = prob.forward(hm.contiguous())
        E.set_default_tuning(search=search, candidates=cand)
        bad = (pr != o["hist_pred"].reshape(n * B, P)).any(1).reshape(n, B)
        w = bad.nonzero().tolist()
        print("search", search, "candidates", cand, "call", c, "n", n, "records whose curves differ from forward(m):", len(w), w[:6])
        for s_, b_ in w[:3]:
            print("    m", [x.hex() for x in o["hist_m"][s_, b_].tolist()], "prev m", [x.hex() for x in o["hist_m"][s_ - 1, b_].tolist()] if s_ else None, "prev curve 39 %s" % o["hist_pred"][s_ - 1, b_, 39].item().hex())
            k = (pr.reshape(n, B, P)[s_, b_] != o["hist_pred"][s_, b_]).nonzero().flatten().tolist()
            print("    step", s_, "chain", b_, "periods", k, "steps", o["hist_pred"][s_, b_][k].tolist(), "forward", pr.reshape(n, B, P)[s_, b_][k].tolist(), "accepted", o["accepted"][max(0, s_ - 3):s_ + 1, b_].tolist())
