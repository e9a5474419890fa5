"""Soak test of the MH kernels: random configurations (layers, periods, datum kind, likelihood,
rotation, tempering betas, candidates in flight, scheduled calls) against the oracle driven with
the same draws; prints every disagreement.  usage: soak_mh.py [configurations]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import oracle
from mcmc_amd.engine import DeviceChains, DeviceProblem

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
t0 = time.time()
oracle.set_threads(16)
for r in (range(n_cfg) if len(sys.argv) < 3 else [int(sys.argv[2])]):
    rng = np.random.default_rng(7000 + r)
    nl = int(rng.integers(2, 9))
    NP = 2 * nl + 1
    P = int(rng.choice([10, 20, 40]))
    itype = int(rng.choice([0, 0, 1, 2, 3]))
    wave = int(itype == 0 and rng.uniform() < 0.25)
    name = {0: "phase", 1: "group", 2: "phase+group", 3: "phase+ellipticity"}[itype]
    ND = 2 * P if itype in (2, 3) else P
    rotate = int(rng.choice([0, 0, 3]))
    cand = int(rng.choice([1, 2, 2, 4]))
    sched = bool(rng.uniform() < 0.5) and cand == 2
    B, S, ncall = 64, 40, 3
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    true_m = np.concatenate([np.sort(rng.uniform(0.005, 0.14, nl)), np.sort(rng.uniform(0.15, 1.8, nl + 1))])
    st0, tc, _ = oracle.forward(true_m, nl, 1.75, periods, wave=wave, itype=itype)
    if st0 != 0:
        continue
    laplace = rng.uniform() < 0.3
    if laplace:
        noise = {"lambd_scale": 1.0, "lambd": rng.uniform(20, 60, ND), "kappa": 0.75}
        dist = "asym-laplace"
        obs = tc + 0.03 * rng.standard_normal(ND)
    else:
        noise = {"std": rng.uniform(0.02, 0.06, ND)}
        dist = "normal"
        obs = tc + noise["std"] * rng.standard_normal(ND)
    starts = np.tile(true_m, (B, 1)) * rng.uniform(0.9, 1.1, (B, NP))
    starts = np.clip(np.stack([oracle.sort_layers(s, nl) for s in starts]), bounds[:, 0], bounds[:, 1])
    beta = rng.uniform(0.2, 1.0, B) if rng.uniform() < 0.4 else None
    prob = DeviceProblem(nl, 1.75, periods, obs, dist, noise, bounds, wave="love" if wave else "rayleigh",
                         itype=name, candidates=cand)
    ch = DeviceChains(prob, starts, 0.05, beta=None if beta is None else torch.tensor(beta, device=prob.device))
    if not sched:
        ch.sched = None
    stg = ch.init_likelihood().cpu().numpy()
    ost, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods, wave=wave, itype=itype, nthreads=16)
    if (stg != ost).any() or (ost != 0).any():
        keep = ost == 0
        if (stg != ost).any():
            print("MISMATCH cfg %d: initial status" % r); bad += 1
        continue
    ologL = np.array([oracle.loglik(obs, v, dist, noise) for v in ovel])
    st = oracle.ChainState(starts, ologL, ovel, 0.05, bounds)
    if rotate:
        rot = np.stack([np.linalg.qr(rng.standard_normal((NP, NP)))[0] for _ in range(B)])
        ch.rot.copy_(torch.tensor(rot, device=prob.device))
        st.rot[:] = rot
    dev = prob.device
    ch.n_evals.zero_()
    lit = 0
    msg = []
    for c in range(ncall):
        idx = rng.integers(0, NP, (S, B)).astype(np.int32)
        cz = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
        ua = rng.uniform(size=(S, B))
        out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cz, device=dev), torch.tensor(ua, device=dev), rotate=rotate)
        acc = np.zeros((S, B), dtype=np.uint8)
        for s in range(S):
            acc[s], _, n = oracle.mh_step(st, idx[s], cz[s], ua[s], nl, 1.75, periods, obs, dist, noise, beta=beta,
                                          wave=wave, itype=itype, rotate=rotate)
            lit += n
        if not (out["accepted"].cpu().numpy() == acc).all():
            msg.append("decisions call %d (%d differ)" % (c, int((out["accepted"].cpu().numpy() != acc).sum())))
            break
    m = ch.m.cpu().numpy()
    if not msg:
        if rotate:
            if not np.allclose(m, st.m, rtol=1e-10, atol=0): msg.append("states")
        elif not np.array_equal(m, st.m): msg.append("states")
        if not np.allclose(ch.logL.cpu().numpy(), st.logL, rtol=1e-9): msg.append("logL")
        if not np.allclose(ch.pred.cpu().numpy(), st.pred, rtol=1e-8, atol=1e-12):
            g, o = ch.pred.cpu().numpy(), st.pred
            i = np.unravel_index(np.argmax(np.abs(g - o) / (np.abs(o) + 1e-300)), g.shape)
            msg.append("pred (worst: chain %d entry %d of %d: %.15g vs %.15g; neighbours %s)" % (i[0], i[1], g.shape[1], g[i], o[i], o[i[0], max(i[1]-2, 0):i[1]+3]))
        if not np.array_equal(ch.counters.cpu().numpy(), st.counters): msg.append("counters")
        if int(ch.n_evals[0]) != lit: msg.append("lit %d vs %d" % (int(ch.n_evals[0]), lit))
    tag = "nl %d P %d %s%s %s rot %d cand %d sched %d beta %d" % (nl, P, name, " love" if wave else "", dist, rotate, cand, sched, beta is not None)
    if msg:
        bad += 1
        print("MISMATCH cfg %d (%s): %s" % (r, tag, ", ".join(msg)), flush=True)
    else:
        print("cfg %d ok (%s) acc %.2f  (%.0f s)" % (r, tag, st.counters[:, 0].sum() / (B * S * ncall), time.time() - t0), flush=True)
print("disagreements:", bad)
