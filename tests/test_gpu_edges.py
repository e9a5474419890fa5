"""Edge cases of the C ABI on the GPU: the compiled maxima, the smallest problems, empty batches,
one chain -- against the CPU oracle where there is something to compare."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _draws(rng, n, n_layers):
    d = np.sort(rng.uniform(0.001, 0.15, (n, n_layers)), axis=1)
    v = rng.uniform(0.1, 2.0, (n, n_layers + 1))
    return np.concatenate([d, v], axis=1)


@pytest.mark.parametrize("n_layers,P", [(15, 128), (1, 1), (1, 128), (15, 1)])
def test_forward_at_the_compiled_limits(oracle, n_layers, P):
    """MHD_MAX_LAYERS finite layers and MHD_MAX_PERIODS periods, and one of each."""
    import torch
    from mcmc_amd.engine import DeviceProblem
    rng = np.random.default_rng(300 + n_layers + P)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy() if P > 1 else np.array([0.2])
    params = _draws(rng, 48, n_layers)
    prob = DeviceProblem(n_layers, 1.75, periods)
    ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
    vel, st = prob.forward(torch.tensor(params, device=prob.device), n_evals=ne)
    ost, ovel, one = oracle.forward_batch(params, n_layers, 1.75, periods, nthreads=8)
    vel, st = vel.cpu().numpy(), st.cpu().numpy()
    assert (st == ost).all()
    ok = ost == 0
    assert ok.sum() > 0
    assert np.array_equal(vel[ok], ovel[ok])          # the same lattice points, bit for bit
    assert int(ne[0]) == one


def test_sizes_beyond_the_limits_are_refused():
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceProblem
    with pytest.raises((L.MhdError, AssertionError, ValueError)):
        DeviceProblem(16, 1.75, np.linspace(0.05, 0.5, 8)).forward(
            __import__("torch").zeros((1, 33), dtype=__import__("torch").float64, device="cuda"))
    with pytest.raises((L.MhdError, AssertionError, ValueError)):
        DeviceProblem(3, 1.75, np.linspace(0.05, 0.5, 129)).forward(
            __import__("torch").zeros((1, 7), dtype=__import__("torch").float64, device="cuda"))


def test_empty_batches_do_nothing():
    import torch
    from mcmc_amd.engine import DeviceProblem
    prob = DeviceProblem(3, 1.75, np.linspace(0.05, 0.5, 8))
    vel, st = prob.forward(torch.zeros((0, 7), dtype=torch.float64, device=prob.device))
    assert vel.shape == (0, 8) and st.shape == (0,)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B", [1, 3])
def test_mh_steps_at_the_limits_match_the_oracle(oracle, B):
    """15 layers (31 parameters), 128 periods, one and three chains with the launch schedule's
    buffer present (too few chains for it to be used), a call of 0 steps in between."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(17 + B)
    nl, P, S = 15, 128, 24
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    starts = []
    while len(starts) < B:
        x = _draws(rng, 1, nl)[0]
        if oracle.forward(x, nl, 1.75, periods)[0] == 0:
            starts.append(x)
    starts = np.array(starts)
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds)
    ch = DeviceChains(prob, starts, 0.02)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    _, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods)
    ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])
    ost = oracle.ChainState(starts, ologL, ovel, 0.02, bounds)
    idx = rng.integers(0, NP, (S, B)).astype(np.int32)
    cauchy = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)) * 0.3
    u_acc = rng.uniform(size=(S, B))
    dev = prob.device
    t = lambda a: torch.tensor(a, device=dev)
    out1 = ch.steps(t(idx[:10]), t(cauchy[:10]), t(u_acc[:10]))
    out0 = ch.steps(t(idx[:0]), t(cauchy[:0]), t(u_acc[:0]))           # nothing to do
    assert out0["accepted"].shape == (0, B)
    out2 = ch.steps(t(idx[10:]), t(cauchy[10:]), t(u_acc[10:]))
    acc = np.zeros((S, B), dtype=np.uint8)
    stt = np.zeros((S, B), dtype=np.int32)
    for s in range(S):
        acc[s], stt[s], _ = oracle.mh_step(ost, idx[s], cauchy[s], u_acc[s], nl, 1.75, periods,
                                           obs, "normal", noise)
    got_acc = torch.cat([out1["accepted"], out2["accepted"]]).cpu().numpy()
    got_st = torch.cat([out1["status"], out2["status"]]).cpu().numpy()
    np.testing.assert_array_equal(got_acc, acc)
    np.testing.assert_array_equal(got_st, stt)
    np.testing.assert_array_equal(ch.m.cpu().numpy(), ost.m)
    np.testing.assert_allclose(ch.logL.cpu().numpy(), ost.logL, rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(ch.counters.cpu().numpy(), ost.counters)


def test_sample_prior_and_curve_history(oracle):
    """sample_prior steps (no forward model, logL_new = 1: model.py:234-236) against the oracle,
    and the per-step records: the last record of every history equals the final state."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(23)
    nl, P, B, S = 3, 12, 80, 30
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    starts = []
    while len(starts) < B:
        x = _draws(rng, 1, nl)[0]
        if oracle.forward(x, nl, 1.75, periods)[0] == 0:
            starts.append(x)
    starts = np.array(starts)
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds)
    dev = prob.device
    idx = rng.integers(0, NP, (S, B)).astype(np.int32)
    cauchy = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
    u_acc = rng.uniform(size=(S, B))
    t = lambda a: torch.tensor(a, device=dev)
    _, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods)
    ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])

    # sample_prior
    ch = DeviceChains(prob, starts, 0.05)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    ost = oracle.ChainState(starts, ologL, ovel, 0.05, bounds)
    out = ch.steps(t(idx), t(cauchy), t(u_acc), sample_prior=True)
    acc = np.zeros((S, B), dtype=np.uint8)
    for s in range(S):
        acc[s], _, _ = oracle.mh_step(ost, idx[s], cauchy[s], u_acc[s], nl, 1.75, periods, obs,
                                      "normal", noise, sample_prior=True)
    np.testing.assert_array_equal(out["accepted"].cpu().numpy(), acc)
    np.testing.assert_array_equal(ch.m.cpu().numpy(), ost.m)
    np.testing.assert_array_equal(ch.counters.cpu().numpy(), ost.counters)
    assert acc.sum() > 0

    # histories of a normal run
    ch = DeviceChains(prob, starts, 0.05)
    ch.init_likelihood()
    out = ch.steps(t(idx), t(cauchy), t(u_acc), hist_every=5, want_pred_hist=True)
    assert out["hist_m"].shape == (S // 5, B, NP)
    assert torch.equal(out["hist_m"][-1], ch.m)
    assert torch.equal(out["hist_logL"][-1], ch.logL)
    assert torch.equal(out["hist_pred"][-1], ch.pred)
    # a record in the middle: the state after 15 steps of the same draws
    ch2 = DeviceChains(prob, starts, 0.05)
    ch2.init_likelihood()
    ch2.steps(t(idx[:15]), t(cauchy[:15]), t(u_acc[:15]))
    assert torch.equal(out["hist_m"][2], ch2.m)
    assert torch.equal(out["hist_pred"][2], ch2.pred)
