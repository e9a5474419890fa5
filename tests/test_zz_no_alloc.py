"""mhd_mh_steps_static, the entry point for callers that cannot allocate: same chains as
mhd_mh_steps, bit for bit, on two calls in a row that share the library's static device
scratch, the second with fewer chains and fewer steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NL, P = 6, 40


def _problem(rng, B):
    import torch
    from mcmc_amd.engine import DeviceProblem
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * NL + [[0.1, 2.0, 1.9]] * (NL + 1))
    prob0 = DeviceProblem(NL, 1.75, periods)
    starts = np.empty((0, 2 * NL + 1))
    while starts.shape[0] < B:
        cand = rng.uniform(bounds[:, 0], bounds[:, 1], size=(2 * B, 2 * NL + 1))
        order = np.argsort(cand[:, :NL], axis=1, kind="stable")
        cand[:, :NL] = np.take_along_axis(cand[:, :NL], order, axis=1)
        cand[:, NL:2 * NL] = np.take_along_axis(cand[:, NL:2 * NL], order, axis=1)
        _, st = prob0.forward(torch.tensor(cand, device=prob0.device))
        starts = np.concatenate([starts, cand[st.cpu().numpy() == 0]])
    starts = starts[:B]
    obs = prob0.forward(torch.tensor(starts[:1], device=prob0.device))[0][0].cpu().numpy() \
        + 0.025 * rng.standard_normal(P)
    prob = DeviceProblem(NL, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
    return prob, starts


def test_no_wait_between_waves_has_timed_out_in_the_tests_before():
    """The waves of a chain's candidates meet by counting arrivals in local memory, with a limit
    that keeps a defect from hanging the GPU; the library counts the waits that ran into it
    (mhd_wait_timeouts).  None may have, in everything the tests collected before this file ran
    in this process."""
    import torch
    from mcmc_amd import _lib as L
    torch.zeros(1, device="cuda")
    L.check_waits()


def test_accept_decisions_near_the_threshold_are_counted():
    """u <= exp(dl) is taken with the device's exponential where the reference takes it with
    numpy's; the two may differ by a unit or two in the last place, so a decision with u within
    16 units of exp(dl) is the only kind that could come out differently.  The library counts
    them (mhd_close_decisions): none in everything the tests before this one ran in this
    process, and every one of a batch of steps whose u is numpy's exp(dl) itself."""
    import torch
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceChains
    torch.zeros(1, device="cuda")
    assert L.close_decisions() == 0, "a parity test before this one had a decision within 16 ulps"
    rng = np.random.default_rng(123)
    B = 96
    prob, starts = _problem(rng, B)
    dev, NP = prob.device, 2 * NL + 1
    idx = torch.tensor(rng.integers(0, NP, (1, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(0.05 * np.tan(np.pi * (rng.uniform(size=(1, B)) - 0.5)), device=dev)
    # first with a tiny u: nearly everything that has a curve is accepted, and leaves its likelihood behind
    ch = DeviceChains(prob, starts, 0.05)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    l_old = ch.logL.cpu().numpy().copy()
    out = ch.steps(idx, cauchy, torch.full((1, B), 1e-300, dtype=torch.float64, device=dev))
    dl = ch.logL.cpu().numpy() - l_old
    moved = (out["accepted"][0].cpu().numpy() == 1) & (np.abs(dl) < 600.0)
    assert moved.sum() > B // 4 and L.close_decisions() == 0
    # then with u = exp(dl) by numpy, in both kernels that settle steps
    u = np.where(moved, np.exp(dl), 0.5)
    expect = 0
    for candidates in (2, 1):
        prob.candidates = candidates
        ch = DeviceChains(prob, starts, 0.05)
        assert int((ch.init_likelihood() != 0).sum()) == 0
        ch.steps(idx, cauchy, torch.tensor(u[None, :], device=dev))
        expect += int(moved.sum())
        assert L.close_decisions() == expect


def test_static_scratch_entry_point_matches_mhd_mh_steps():
    import torch
    from mcmc_amd.engine import DeviceChains
    rng = np.random.default_rng(77)
    # two calls in a row: 256 chains x 120 steps (a scheduled call of two launches), then 96
    # chains x 100 steps on the same static scratch
    for B, S in ((256, 120), (96, 100)):
        prob, starts = _problem(rng, B)
        dev, NP = prob.device, 2 * NL + 1
        idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
        cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
        u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
        res = []
        for static in (False, True):
            ch = DeviceChains(prob, starts, 0.05)
            if static:
                ch.sched = None           # the entry point must not need it
            assert int((ch.init_likelihood() != 0).sum()) == 0
            # two calls per case, so that the second finds the costs the first left
            out1 = ch.steps(idx[:S // 2], cauchy[:S // 2], u_acc[:S // 2], hist_every=10,
                            static_scratch=static)
            out2 = ch.steps(idx[S // 2:], cauchy[S // 2:], u_acc[S // 2:], hist_every=10,
                            static_scratch=static)
            torch.cuda.synchronize()
            res.append((ch.m.clone(), ch.logL.clone(), ch.pred.clone(), ch.counters.clone(),
                        ch.params_prev.clone(),
                        torch.cat([out1["accepted"], out2["accepted"]]),
                        torch.cat([out1["status"], out2["status"]]),
                        torch.cat([out1["hist_m"], out2["hist_m"]]),
                        torch.cat([out1["hist_logL"], out2["hist_logL"]]),
                        int(ch.n_evals[0])))
        for x, y in zip(res[0][:-1], res[1][:-1]):
            assert torch.equal(x, y)                      # integer and FP64 state: bit-exact
        assert res[0][-1] == res[1][-1] > 0               # same sequential-algorithm evaluations
        assert int(res[1][3][:, 0].sum()) > 0             # something was accepted


def test_static_scratch_entry_point_refuses_what_it_cannot_hold():
    import ctypes
    import torch
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceChains, _ptr, _stream
    rng = np.random.default_rng(5)
    prob, starts = _problem(rng, 8)
    B = 4097
    ch = DeviceChains(prob, np.tile(starts[:1], (B, 1)), 0.05)
    z = torch.zeros((1, B), dtype=torch.float64, device=prob.device)
    zi = torch.zeros((1, B), dtype=torch.int32, device=prob.device)
    s = ch.struct()
    rc = L.lib().mhd_mh_steps_static(prob.cref(), ctypes.byref(s), B, 1, _ptr(zi), _ptr(z), _ptr(z),
                                     1.0, 0, 0, None, None, 0, None, None, None, None, None,
                                     _stream(prob.device))
    assert rc == -22 and b"4096" in L.lib().mhd_last_error()


def test_static_reset_clears_the_scratch_for_whoever_comes_next():
    """mhd_static_reset: after it a call adopts its chains without clearing (the rule for
    callers under graph capture and for allocators that hand out an old address again); the
    chains are those of mhd_mh_steps whatever the scratch held before."""
    import torch
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceChains, _stream
    rng = np.random.default_rng(91)
    B, S = 128, 60
    prob, starts = _problem(rng, B)
    dev, NP = prob.device, 2 * NL + 1
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    ref = DeviceChains(prob, starts, 0.05)
    assert int((ref.init_likelihood() != 0).sum()) == 0
    ref.steps(idx, cauchy, u_acc)
    for reset in (False, True, True):
        ch = DeviceChains(prob, starts, 0.05)
        ch.sched = None
        assert int((ch.init_likelihood() != 0).sum()) == 0
        if reset:
            L.check(L.lib().mhd_static_reset(_stream(dev)))
        ch.steps(idx[:S // 2], cauchy[:S // 2], u_acc[:S // 2], static_scratch=True)
        ch.steps(idx[S // 2:], cauchy[S // 2:], u_acc[S // 2:], static_scratch=True)
        torch.cuda.synchronize()
        assert torch.equal(ch.m, ref.m) and torch.equal(ch.logL, ref.logL)
        assert torch.equal(ch.counters, ref.counters)


def test_static_entry_point_under_graph_capture():
    """The use the entry point is for: mhd_static_reset, then a call captured into a HIP graph
    (torch.cuda.graph) and replayed twice on fresh draws.  Every launch of a call is on the
    caller's stream and nothing is allocated, so the capture holds the whole call -- on the
    second replay a scheduled one of two launches, made from the costs the first left in the
    static scratch.  The chains are those of mhd_mh_steps called eagerly on the same draws."""
    import torch
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceChains
    rng = np.random.default_rng(123)
    B, S = 128, 60
    prob, starts = _problem(rng, B)
    dev, NP = prob.device, 2 * NL + 1
    draws = [(torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev),
              torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev),
              torch.tensor(rng.uniform(size=(S, B)), device=dev)) for _ in range(2)]
    eager = DeviceChains(prob, starts, 0.05)
    assert int((eager.init_likelihood() != 0).sum()) == 0
    want = []
    for idx, cz, ua in draws:
        out = eager.steps(idx, cz, ua)
        want.append((eager.m.clone(), eager.logL.clone(), eager.counters.clone(),
                     out["accepted"].clone()))
    ch = DeviceChains(prob, starts, 0.05)
    ch.sched = None
    assert int((ch.init_likelihood() != 0).sum()) == 0
    idx_s, cz_s, ua_s = (torch.empty_like(x) for x in draws[0])
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(stream):
        L.check(L.lib().mhd_static_reset(
            __import__("ctypes").c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            out = ch.steps(idx_s, cz_s, ua_s, static_scratch=True)
        for (idx, cz, ua), (m, logL, counters, acc) in zip(draws, want):
            idx_s.copy_(idx)
            cz_s.copy_(cz)
            ua_s.copy_(ua)
            graph.replay()
            stream.synchronize()
            assert torch.equal(ch.m, m) and torch.equal(ch.counters, counters)
            assert torch.equal(ch.logL, logL) and torch.equal(out["accepted"], acc)
    torch.cuda.current_stream(dev).wait_stream(stream)
    assert int(ch.n_evals[0]) == int(eager.n_evals[0])
    L.check_waits()
