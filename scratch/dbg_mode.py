import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle
from mcmc_amd import _lib as L
from mcmc_amd.engine import DeviceProblem
def _draws(rng, n, n_layers):
    d = np.sort(rng.uniform(0.001, 0.15, (n, n_layers)), axis=1)
    v = rng.uniform(0.1, 2.0, (n, n_layers + 1))
    return np.concatenate([d, v], axis=1)
for mode in (1, 2, 3):
    for search in (2, 1):
        rng = np.random.default_rng(41 + mode)
        nl, P, B = 4, 30, 192
        periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
        params = _draws(rng, B, nl)
        L.check(L.lib().mhd_set_search(search))
        prob = DeviceProblem(nl, 1.75, periods, mode=mode)
        ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
        vel, st = prob.forward(torch.tensor(params, device=prob.device), n_evals=ne)
        ost, ovel, one = oracle.forward_batch(params, nl, 1.75, periods, mode=mode, nthreads=8)
        vel, st = vel.cpu().numpy(), st.cpu().numpy()
        ok = (ost == 0) & (st == 0)
        d = vel[ok] != ovel[ok]
        rows = np.flatnonzero(d.any(1))
        print("mode", mode, "search", search, "status diffs", int((st != ost).sum()), "curves differing", len(rows), "of", int(ok.sum()), "evals", int(ne[0]), one)
        for r in rows[:3]:
            ks = np.flatnonzero(d[r]); print("   row", r, "periods", ks[:8], "gpu", vel[ok][r][ks[:4]], "oracle", ovel[ok][r][ks[:4]])
L.check(L.lib().mhd_set_search(2))
