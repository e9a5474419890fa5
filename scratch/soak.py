"""Soak test: forward curves of many random models (all layer counts, unsorted velocities, both
searches) and MH steps at several seeds against the oracle; prints every disagreement.
usage: soak.py [rounds] [models per round]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import oracle
from mcmc_amd.engine import DeviceChains, DeviceProblem

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nt = 16
bad = 0
t0 = time.time()
for r in range(rounds):
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
    wave = int(r % 5 == 4)
    mode = int(r % 7 == 6)
    ost, ovel, olit = oracle.forward_batch(params, nl, vpvs, periods, wave=wave, mode=mode, nthreads=nt)
    for search in (2, 1):
        prob = DeviceProblem(nl, vpvs, periods, wave="love" if wave else "rayleigh", mode=mode, search=search)
        ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
        vel, st = prob.forward(torch.tensor(params, device=prob.device), n_evals=ne)
        vel, st = vel.cpu().numpy(), st.cpu().numpy()
        ok = ost == 0
        nst = int((st != ost).sum())
        nv = int((vel[ok & (st == 0)] != ovel[ok & (st == 0)]).any(axis=1).sum())
        nlit = int(ne[0]) - olit
        if nst or nv or nlit:
            bad += 1
            print("MISMATCH round %d search %d: nl %d P %d wave %d mode %d: status %d curves %d lit %d" % (r, search, nl, P, wave, mode, nst, nv, nlit), flush=True)
            rows = np.where((st != ost) | ((vel != ovel).any(axis=1) & ok))[0][:3]
            for i in rows:
                print("   model", i, repr(params[i].tolist()), "status", st[i], ost[i], flush=True)
    print("round %d nl %d P %d f %.2f..%.2f vpvs %.2f wave %d mode %d ok %.2f  (%.0f s)" % (r, nl, P, fmin, fmax, vpvs, wave, mode, (ost == 0).mean(), time.time() - t0), flush=True)
print("disagreements:", bad)
