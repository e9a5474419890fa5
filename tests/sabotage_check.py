"""Run by tests/test_gpu_deferred_tails.py in a process of its own, with MCMC_AMD_LIB naming the
test build of the library in which every third interpolated root of the deferring period search
is wrong: the verification of the inferred halvings must catch each of them and the periods be
searched again, so curves, evaluation counts, decisions and states are still the oracle's.
The same build tells the candidates of the multi-candidate kernels to give their curves up without
reason (every wave that finishes asks the later ones to, and every third round they find the request
before they start) and runs every shape of the candidates' tree in turn (candidates that follow
acceptances as well as rejections, whatever the chains accept): the steps whose curves are missing
are run again and nothing else changes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def lvz_models(rng, B, nl):
    """Prior draws with unsorted velocities: low-velocity zones, curves that change branches."""
    d = np.sort(rng.uniform(0.001, 0.15, (B, nl)), axis=1)
    v = rng.uniform(0.1, 2.0, (B, nl + 1))
    return np.concatenate([d, v], axis=1)


def main():
    import torch
    import oracle
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    assert L.LIB_PATH.endswith("libmcmc_amd_sabotage.so"), L.LIB_PATH
    rng = np.random.default_rng(77)
    nl, P, B = 6, 40, 256
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    params = lvz_models(rng, B, nl)
    ost, ovel, olit = oracle.forward_batch(params, nl, 1.75, periods, nthreads=8)
    execs = {}
    for search in (1, 2):
        prob = DeviceProblem(nl, 1.75, periods, search=search)
        ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
        vel, st = prob.forward(torch.tensor(params, device=prob.device), n_evals=ne)
        vel, st = vel.cpu().numpy(), st.cpu().numpy()
        ok = ost == 0
        assert (st == ost).all() and ok.sum() > B // 2
        assert np.array_equal(vel[ok], ovel[ok])
        assert int(ne[0]) == olit, (search, int(ne[0]), olit)
        execs[search] = int(ne[1])
    # the sabotage is at work: the product build executes ~7.6 evaluations per sequential one
    assert execs[1] > 9.0 * olit, (execs, olit)

    # A 3-layer model whose root at the 13th period lies just under the ceiling (the largest
    # layer velocity) and whose 14th period has none.  The bracket of the 13th straddles the
    # ceiling, so an answer inferred from a wrong root can lie above it: "no root" must not be
    # said on an inferred answer (the deferring search once did, one period early).
    m13 = np.array([float.fromhex(v) for v in (
        "0x1.40edce5fc6e56p-8", "0x1.b1191cf2c4ff1p-8", "0x1.f8fc5e970f1d3p-6",
        "0x1.7c6752b8e9f87p+0", "0x1.e001947c78776p-1", "0x1.0c693717137adp+0",
        "0x1.a19149b61738ap-2")])
    for n_per, want_st in ((13, 0), (14, 3), (40, 3)):
        o_st, o_vel, o_lit = oracle.forward(m13, 3, 1.75, periods[:n_per])
        assert o_st == want_st
        for search in (1, 2):
            prob = DeviceProblem(3, 1.75, periods[:n_per], search=search)
            ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
            vel, st = prob.forward(torch.tensor(m13[None], device=prob.device), n_evals=ne)
            assert st.tolist() == [o_st] and int(ne[0]) == o_lit, (n_per, search, st.tolist(),
                                                                  int(ne[0]), o_lit)
            if o_st == 0:
                assert np.array_equal(vel[0].cpu().numpy(), o_vel)

    # MH steps with the wave-sequential and the predicted search
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    good = params[ost == 0][:64]
    Bc, S = len(good), 40
    obs = ovel[ost == 0][0] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    idx = rng.integers(0, 2 * nl + 1, (S, Bc)).astype(np.int32)
    cz = np.tan(np.pi * (rng.uniform(size=(S, Bc)) - 0.5))
    ua = rng.uniform(size=(S, Bc))
    _, v0, _ = oracle.forward_batch(good, nl, 1.75, periods, nthreads=8)
    logL0 = np.array([oracle.loglik(obs, v, "normal", noise) for v in v0])
    state = oracle.ChainState(good, logL0, v0, 0.05, bounds)
    acc = np.zeros((S, Bc), dtype=np.uint8)
    lit = 0
    oracle.set_threads(8)
    for s in range(S):
        acc[s], _, n = oracle.mh_step(state, idx[s], cz[s], ua[s], nl, 1.75, periods, obs,
                                      "normal", noise)
        lit += n
    oracle.set_threads(1)
    for search, candidates in ((1, 1), (2, 2), (2, 4)):
        prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds, search=search,
                             candidates=candidates)
        ch = DeviceChains(prob, good, 0.05)
        assert int((ch.init_likelihood() != 0).sum()) == 0
        ch.n_evals.zero_()
        dev = prob.device
        out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cz, device=dev),
                       torch.tensor(ua, device=dev))
        assert (out["accepted"].cpu().numpy() == acc).all(), (search, candidates)
        assert np.array_equal(ch.m.cpu().numpy(), state.m)
        assert np.array_equal(ch.pred.cpu().numpy(), state.pred)
        np.testing.assert_allclose(ch.logL.cpu().numpy(), state.logL, rtol=1e-9)
        assert int(ch.n_evals[0]) == lit, (search, candidates, int(ch.n_evals[0]), lit)
    # the unjustified requests were at work: curves were given up, and some of them were needed
    import ctypes
    # (and steps were settled by candidates that started from an earlier candidate's state:
    # the build runs every shape of the candidates' tree in turn)
    stats = (ctypes.c_ulonglong * 3)()
    assert L.lib().mhd_debug_aborts(stats) == 0
    assert stats[0] > 100 and stats[1] > 20 and stats[2] > 50, list(stats)
    print("sabotage check ok: executed / sequential evaluations %.1f (sequential search), %.1f "
          "(predicted); %d curves given up on request, %d of them needed and run again; %d steps "
          "settled by candidates that follow an acceptance"
          % (execs[1] / olit, execs[2] / olit, stats[0], stats[1], stats[2]))


if __name__ == "__main__":
    main()
