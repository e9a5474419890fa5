"""Soak of the launch schedule: random numbers of chains (odd ones too), call lengths on both
sides of the two-launch threshold, layers, periods, rotated proposals, tempering betas -- the
chains of scheduled calls (through mhd_mh_steps and through mhd_mh_steps_static) must equal those
of the same calls without the schedule, bit for bit
(states, likelihoods, curves, counters, decisions, history, literal evaluation count), and no
wait between waves may have timed out.  usage: soak_sched.py [configurations] [first configuration]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from mcmc_amd import _lib as L
from mcmc_amd.engine import DeviceChains, DeviceProblem

# what a call does not write stays recognisable: the outputs start as NaN / 77 instead of whatever
# the allocator hands out (two runs could otherwise agree on an unwritten entry by getting the same
# block of memory back)
import mcmc_amd.engine as _E
class _PoisonedTorch:
    def __getattr__(self, name):
        if name != "empty":
            return getattr(torch, name)
        def empty(*a, **k):
            t = torch.empty(*a, **k)
            return t.fill_(float("nan") if t.is_floating_point() else 77)
        return empty
_E.torch = _PoisonedTorch()

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for r in range(first, first + n_cfg):
    rng = np.random.default_rng(9000 + r)
    nl = int(rng.integers(2, 8)); NP = 2 * nl + 1
    P = int(rng.choice([8, 16, 24, 40]))
    B = int(rng.choice([64, 65, 67, 96, 129, 200, 255, 256, 383, 512, 700, 1024, 1100]))
    itype = str(rng.choice(["phase", "phase", "group", "phase+ellipticity"]))
    rotate = int(rng.choice([0, 0, 3]))
    calls = [int(x) for x in rng.choice([3, 7, 12, 20, 30, 47, 48, 60, 90, 130], size=int(rng.integers(3, 6)))]
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    # (from a generator of its own, so that the configurations stay what they were: the wave
    # type, the search, the candidates in flight where the schedule leaves the choice, the
    # likelihood)
    # (configurations below 6000 are the plain ones of the earlier runs: Rayleigh waves, the
    # default search and candidates, the normal likelihood)
    rng3 = np.random.default_rng(29000 + r)
    vary = r >= 6000
    wave = "love" if vary and itype in ("phase", "group") and rng3.uniform() < 0.2 else "rayleigh"
    prob0 = DeviceProblem(nl, 1.75, periods, itype=itype, wave=wave)
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
    srch = int(rng3.choice([0, 0, 0, 0, 1])) if vary else 0
    cands = int(rng3.choice([0, 0, 0, 4, 1])) if vary else 0
    if vary and rng3.uniform() < 0.25:
        noise = ("asym-laplace", {"lambd": rng3.uniform(20.0, 60.0, true.shape[0]), "lambd_scale": 1.0, "kappa": rng3.uniform(0.6, 1.6, true.shape[0])})
    else:
        noise = ("normal", {"std": 0.03})
    prob = DeviceProblem(nl, 1.75, periods, obs, noise[0], noise[1], bounds, itype=itype, wave=wave, search=srch, candidates=cands)
    dev = prob.device
    beta = rng.uniform(0.05, 1.0, B) if rng.uniform() < 0.4 else None
    S = sum(calls)
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cz = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    ua = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    rot = torch.tensor(np.stack([np.linalg.qr(rng.standard_normal((NP, NP)))[0] for _ in range(B)]), device=dev) if rotate else None
    # every seventh configuration mixes layer counts (rows padded to nl; no rotation with those):
    # drawn from a generator of its own, so that the other configurations stay what they were
    nb = None
    if r % 7 == 5 and nl >= 3:
        rng2 = np.random.default_rng(19000 + r)
        rotate, rot = 0, None
        rows, nb = np.empty((0, NP)), np.empty(0, dtype=np.int32)
        while rows.shape[0] < B:
            n_c = rng2.integers(1, nl + 1, 2 * B).astype(np.int32)
            cand = np.full((2 * B, NP), 0.777)
            for i, n in enumerate(n_c):
                cand[i, :n] = np.sort(rng2.uniform(0.001, 0.15, n))
                cand[i, nl:nl + n + 1] = np.sort(rng2.uniform(0.1, 2.0, n + 1))
            _, st = prob0.forward(torch.tensor(cand, device=dev), n_layers_b=n_c)
            good = st.cpu().numpy() == 0
            rows, nb = np.concatenate([rows, cand[good]]), np.concatenate([nb, n_c[good]])
        rows, nb = rows[:B], nb[:B].astype(np.int32)
        starts = rows
        jj = rng2.integers(0, 2 ** 31, (S, B))
        ic = jj % (2 * nb[None, :] + 1)
        idx = torch.tensor(np.where(ic < nb[None, :], ic, nl + ic - nb[None, :]).astype(np.int32), device=dev)
    res = []
    # the caller's schedule buffer; none (one launch per call); the library's static buffer
    for scheduled in (True, False, "static"):
        ch = DeviceChains(prob, starts, 0.05, beta=None if beta is None else torch.tensor(beta, device=dev), n_layers_b=nb)
        if not scheduled:
            ch.sched = None
        if scheduled == "static":
            L.check(L.lib().mhd_static_reset(None))
        if rot is not None:
            ch.rot.copy_(rot)
        assert int((ch.init_likelihood() != 0).sum()) == 0
        outs, a = [], 0
        for n in calls:
            o = ch.steps(idx[a:a + n], cz[a:a + n], ua[a:a + n], rotate=rotate, hist_every=max(1, n // 3),
                         static_scratch=scheduled == "static")
            outs.append(torch.cat([o["accepted"].flatten().double(), o["status"].flatten().double(), o["hist_logL"].flatten(), o["hist_m"].flatten()]))
            assert not bool(torch.isnan(outs[-1]).any()) and not bool((o["accepted"] == 77).any()), "an output was not written"
            a += n
        res.append((ch.m.clone(), ch.logL.clone(), ch.pred.clone(), ch.counters.clone(), torch.cat(outs), int(ch.n_evals[0])))
    ok = all(torch.equal(x, y) for x, y in zip(res[0][:5], res[1][:5])) and res[0][5] == res[1][5]
    if not (all(torch.equal(x, y) for x, y in zip(res[2][:5], res[1][:5])) and res[2][5] == res[1][5]):
        ok = False
        print("   (the static entry point differs)")
    tag = "nl %d%s P %d B %d %s %s %s search %d cand %d rot %d beta %d calls %s" % (nl, "" if nb is None else " (mixed)", P, B, itype, wave, noise[0], srch, cands, rotate, beta is not None, calls)
    if not ok:
        bad += 1
        names = ["m", "logL", "pred", "counters", "flags+history"]
        what = []
        for nm, x, y in zip(names, res[0][:5], res[1][:5]):
            if not torch.equal(x, y):
                d = (x != y)
                rows = d.reshape(d.shape[0], -1).any(dim=1).nonzero().flatten().tolist() if d.dim() > 1 else d.nonzero().flatten().tolist()
                what.append("%s (%d entries, first rows %s)" % (nm, int(d.sum()), rows[:6]))
        what.append("lit %d vs %d" % (res[0][5], res[1][5]))
        print("MISMATCH cfg %d (%s): %s" % (r, tag, "; ".join(what)), flush=True)
    else:
        print("cfg %d ok (%s)  (%.0f s)" % (r, tag, time.time() - t0), flush=True)
L.check_waits()
print("disagreements:", bad)
