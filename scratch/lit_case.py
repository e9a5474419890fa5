"""Where the count of the sequential algorithm's evaluations of one configuration of
scratch/soak_sched.py differs between launches by the schedule and one launch per call: per chain
and step, from a library built with -DMHD_LIT_DEBUG (the count in slot 3 of the counters).
usage: lit_case.py <configuration>"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mcmc_amd import _lib as L
from mcmc_amd.engine import DeviceChains, DeviceProblem
r = int(sys.argv[1])
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
rot = torch.tensor(np.stack([np.linalg.qr(rng.standard_normal((NP, NP)))[0] for _ in range(B)]), device=dev) if rotate else None
print("cfg", r, nl, P, B, itype, rotate, calls, beta is not None)
for every in (0, 1):
    res = []
    for scheduled in (True, False):
        ch = DeviceChains(prob, starts, 0.05, beta=None if beta is None else torch.tensor(beta, device=dev))
        if not scheduled:
            ch.sched = None
        if rot is not None:
            ch.rot.copy_(rot)
        ch.init_likelihood()
        a, rec = 0, []
        for n in calls:
            he = every or max(1, n // 3)
            o = ch.steps(idx[a:a + n], cz[a:a + n], ua[a:a + n], rotate=rotate, hist_every=he, want_diag=True)
            rec.append((o["hist_counters"][:, :, 3].clone(), ch.counters[:, 3].clone(), o["status"].clone(), o["accepted"].clone(), o["hist_m"].clone()))
            a += n
        res.append((rec, int(ch.n_evals[0])))
    print("history every", every or "n/3", "lit", res[0][1], res[1][1])
    for c, (x, y) in enumerate(zip(res[0][0], res[1][0])):
        d = (x[1] != y[1]).nonzero().flatten().tolist()
        if d:
            print("  call", c, "n", calls[c], "chains whose count differs", d[:8], "scheduled - plain", (x[1] - y[1])[d[:8]].tolist())
            for b_ in d[:3]:
                w = (x[0][:, b_] != y[0][:, b_]).nonzero().flatten().tolist()
                if w:
                    h = w[0]
                    print("    chain", b_, "first record that differs", h, "of", x[0].shape[0], "counts", x[0][max(0, h - 2):h + 2, b_].tolist(), y[0][max(0, h - 2):h + 2, b_].tolist(),
                          "status", x[2][:, b_].tolist()[-12:] if every == 0 else x[2][max(0, h - 3):h + 2, b_].tolist(), "accepted", x[3][max(0, h - 3):h + 2, b_].tolist() if every else "")
                    if every:
                        print("    m after the step", [v.hex() for v in x[4][h, b_].tolist()])
                        print("    m before        ", [v.hex() for v in x[4][h - 1, b_].tolist()] if h else None)
                        # the candidate of that step, as propose_wave makes it (rotated frame)
                        j = sum(calls[:c]) + h
                        kk, z = int(idx[j, b_]), float(cz[j, b_])
                        mb = x[4][h - 1, b_].tolist()
                        lo, rg = bounds[:, 0].tolist(), bounds[:, 2].tolist()
                        mn = [(mb[l] - lo[l]) / rg[l] for l in range(NP)]
                        R = rot[b_].flatten().tolist() if rot is not None else None
                        if R is None:
                            mn[kk] = mn[kk] + 0.05 * z
                            test = [mn[l] * rg[l] + lo[l] for l in range(NP)]
                        else:
                            tmp = []
                            for l in range(NP):
                                acc = 0.0
                                for q in range(NP):
                                    acc += R[q * NP + l] * mn[q]
                                if l == kk:
                                    acc = acc + 0.05 * z
                                tmp.append(acc)
                            test = []
                            for l in range(NP):
                                acc = 0.0
                                for q in range(NP):
                                    acc += R[l * NP + q] * tmp[q]
                                test.append(acc * rg[l] + lo[l])
                        order = sorted(range(nl), key=lambda i: (test[i], i))
                        test = [test[i] for i in order] + [test[nl + i] for i in order] + [test[2 * nl]]
                        print("    candidate", [float(v).hex() for v in test], "in bounds", all(bounds[l, 0] <= test[l] <= bounds[l, 1] for l in range(NP)))
                        import oracle
                        import mcmc_amd.engine as E
                        ost, ovel, one = oracle.forward(np.array(test), nl, 1.75, periods)
                        print("    oracle: status", ost, "evaluations", one)
                        for srch in (1, 2):
                            E.set_default_tuning(search=srch)
                            ne = torch.zeros(2, dtype=torch.int64, device=dev)
                            v, st_ = prob.forward(torch.tensor([test], dtype=torch.float64, device=dev), n_evals=ne)
                            print("    device forward search", srch, "status", st_.tolist(), "evaluations", ne.tolist())
                        E.set_default_tuning(search=0)
