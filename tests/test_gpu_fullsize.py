"""The headline configuration at full size (6 layers, 40 periods, 1024 chains), checked through
properties that need no CPU oracle run of that size: every search algorithm and every way of
running the chains must give the same curves, states, counters and literal evaluation count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NL, P, B = 6, 40, 1024


def _setup(seed):
    import torch
    from mcmc_amd.engine import DeviceProblem
    rng = np.random.default_rng(seed)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * NL + [[0.1, 2.0, 1.9]] * (NL + 1))
    prob0 = DeviceProblem(NL, 1.75, periods)
    cand = rng.uniform(bounds[:, 0], bounds[:, 1], size=(3 * B, 2 * NL + 1))
    order = np.argsort(cand[:, :NL], axis=1, kind="stable")
    cand[:, :NL] = np.take_along_axis(cand[:, :NL], order, axis=1)
    cand[:, NL:2 * NL] = np.take_along_axis(cand[:, NL:2 * NL], order, axis=1)
    return rng, periods, bounds, prob0, torch.tensor(cand, device=prob0.device)


def test_forward_both_searches_agree_at_full_size():
    from mcmc_amd import _lib as L, engine as E
    _, _, _, prob0, cand = _setup(11)
    out = {}
    import torch
    for alg in (2, 1):
        E.set_default_tuning(search=alg)
        ne = torch.zeros(2, dtype=torch.int64, device=prob0.device)
        vel, st = prob0.forward(cand, n_evals=ne)
        out[alg] = (vel.clone(), st.clone(), int(ne[0]))
    E.set_default_tuning(search=0)
    assert out[1][2] == out[2][2] > 0                    # same sequential-algorithm evaluation count
    assert bool((out[1][1] == out[2][1]).all())
    assert bool((out[1][0] == out[2][0]).all())          # bit-identical curves, failures included
    ok = out[2][1] == 0
    assert int(ok.sum()) > B and int((~ok).sum()) > 0    # both outcomes occur among prior draws
    v = out[2][0][ok]
    assert bool((v > 0).all()) and bool((v < 2.0).all())


def test_chains_every_way_of_running_agrees_at_full_size():
    import torch
    from mcmc_amd import _lib as L, engine as E
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng, periods, bounds, prob0, cand = _setup(12)
    _, st = prob0.forward(cand)
    starts = cand[st == 0][:B].cpu().numpy()
    assert starts.shape[0] == B
    obs = prob0.forward(torch.tensor(starts[:1], device=prob0.device))[0][0].cpu().numpy() \
        + 0.025 * rng.standard_normal(P)
    prob = DeviceProblem(NL, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
    dev, NP, S = prob.device, 2 * NL + 1, 240   # two calls of 120 steps: the second is two launches
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    results = []
    # scheduled two/four/one candidates; plain two candidates; one candidate with either search
    for cands, search, sched in ((2, 2, True), (2, 2, False), (1, 2, False), (1, 1, False)):
        E.set_default_tuning(candidates=cands)
        E.set_default_tuning(search=search)
        ch = DeviceChains(prob, starts, 0.05)
        if not sched:
            ch.sched = None
        assert int((ch.init_likelihood() != 0).sum()) == 0
        acc = []
        for a in range(0, S, 120):
            out = ch.steps(idx[a:a + 120], cauchy[a:a + 120], u_acc[a:a + 120])
            acc.append(out["accepted"])
        results.append((ch.m.clone(), ch.logL.clone(), ch.pred.clone(), ch.counters.clone(),
                        torch.cat(acc), int(ch.n_evals[0])))
    E.set_default_tuning(candidates=0)
    E.set_default_tuning(search=0)
    ref = results[0]
    for other in results[1:]:
        for x, y in zip(ref[:5], other[:5]):
            assert torch.equal(x, y)
        assert ref[5] == other[5]
    cnt = ref[3].cpu().numpy()
    assert (cnt[:, 0] + cnt[:, 1] == S).all()            # every step accepted or rejected
    assert (cnt[:, 2] + cnt[:, 4] <= cnt[:, 1]).all()    # invalid proposals are rejections
    assert 0.1 < cnt[:, 0].sum() / (B * S) < 0.6


def test_more_chains_than_the_gpu_holds_go_through_the_many_chains_build():
    """With more chains than the GPU holds at two waves per SIMD the engine sends the steps
    through the build of the library with three waves per SIMD (mcmc_amd/build.py): the same
    chains bit for bit as the default build gives, both where that build still holds all chains
    at once (1280) and where its launches are a queue (1792)."""
    import torch
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    assert L.lib_many() is not None, "build first: python -m mcmc_amd.build"
    rng = np.random.default_rng(2024)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * NL + [[0.1, 2.0, 1.9]] * (NL + 1))
    prob0 = DeviceProblem(NL, 1.75, periods)
    for Bm, S in ((1280, 60), (1792, 30)):
        starts = np.empty((0, 2 * NL + 1))
        while starts.shape[0] < Bm:
            cand = rng.uniform(bounds[:, 0], bounds[:, 1], size=(2 * Bm, 2 * NL + 1))
            order = np.argsort(cand[:, :NL], axis=1, kind="stable")
            cand[:, :NL] = np.take_along_axis(cand[:, :NL], order, axis=1)
            cand[:, NL:2 * NL] = np.take_along_axis(cand[:, NL:2 * NL], order, axis=1)
            _, st = prob0.forward(torch.tensor(cand, device=prob0.device))
            starts = np.concatenate([starts, cand[st.cpu().numpy() == 0]])
        starts = starts[:Bm]
        obs = prob0.forward(torch.tensor(starts[:1], device=prob0.device))[0][0].cpu().numpy() \
            + 0.025 * rng.standard_normal(P)
        prob = DeviceProblem(NL, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
        dev, NP = prob.device, 2 * NL + 1
        idx = torch.tensor(rng.integers(0, NP, (S, Bm)).astype(np.int32), device=dev)
        cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, Bm)) - 0.5)), device=dev)
        u_acc = torch.tensor(rng.uniform(size=(S, Bm)), device=dev)
        res = []
        for many in (True, False):
            ch = DeviceChains(prob, starts, 0.05)
            assert ch.steps_lib is L.lib_many()
            if not many:
                ch.steps_lib = L.lib()
            assert int((ch.init_likelihood() != 0).sum()) == 0
            out1 = ch.steps(idx[:S // 2], cauchy[:S // 2], u_acc[:S // 2], hist_every=10)
            out2 = ch.steps(idx[S // 2:], cauchy[S // 2:], u_acc[S // 2:], hist_every=5)
            torch.cuda.synchronize()
            res.append((ch.m.clone(), ch.logL.clone(), ch.pred.clone(), ch.counters.clone(),
                        torch.cat([out1["accepted"], out2["accepted"]]),
                        torch.cat([out1["status"], out2["status"]]),
                        out1["hist_m"], out2["hist_logL"], int(ch.n_evals[0])))
        for x, y in zip(res[0][:-1], res[1][:-1]):
            assert torch.equal(x, y)
        assert res[0][-1] == res[1][-1] > 0
        assert int(res[0][3][:, 0].sum()) > 0
    L.check_waits()
