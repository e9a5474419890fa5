"""Rayleigh-wave ellipticity on the device (the datum kinds "phase+ellipticity" and
"ellipticity") against the oracle: forward batches, Metropolis-Hastings steps on the joint
dispersion + ellipticity likelihood, and the EllipticityParams drop-in through Inversion.

The reference holds no values for this datum (README.md:4 names it, EllipticityParams at
src/inversion/model_params.py:212-213 is empty): parity with disba is unpinned; the oracle's
values are checked against theory in tests/test_oracle_ellipticity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 40


def _params(rng, nl, B):
    d = np.sort(rng.uniform(0.001, 0.15, (B, nl)), axis=1)
    v = np.sort(rng.uniform(0.1, 2.0, (B, nl + 1)), axis=1)
    return np.concatenate([d, v], axis=1)


@pytest.mark.parametrize("search", [2, 1])
@pytest.mark.parametrize("nl,mode", [(6, 0), (3, 0), (12, 0), (4, 1)])
def test_forward_ellipticity(oracle, nl, mode, search):
    import torch
    from mcmc_amd.engine import DeviceProblem
    rng = np.random.default_rng(100 * nl + mode)
    B = 128
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    params = _params(rng, nl, B)
    ost, ovel, oev = oracle.forward_batch(params, nl, 1.75, periods, mode=mode, itype=3, nthreads=8)
    ok = ost == 0
    assert ok.sum() > B // 2
    for itype, code in (("phase+ellipticity", 3), ("ellipticity", 4)):
        prob = DeviceProblem(nl, 1.75, periods, mode=mode, itype=itype, search=search)
        ev = torch.zeros(2, dtype=torch.int64, device=prob.device)
        vel, st = prob.forward(torch.tensor(params, device=prob.device), n_evals=ev)
        vel, st = vel.cpu().numpy(), st.cpu().numpy()
        assert (st == ost).all() and int(ev.cpu().numpy()[0]) == oev
        if code == 3:
            assert vel.shape == (B, 2 * P)
            assert np.array_equal(vel[ok, :P], ovel[ok, :P])           # phase: lattice points
            ell = vel[ok, P:]
        else:
            assert vel.shape == (B, P)
            ell = vel[ok]
        np.testing.assert_allclose(ell, ovel[ok, P:], rtol=1e-8, atol=1e-12)
        # no root (a higher mode past its cut-off): 0 in both halves
        assert ((ovel[ok, :P] == 0.0) == (ell == 0.0)).all()
    assert np.isfinite(ovel[ok]).all() and (np.abs(ovel[ok, P:]) > 0).any()


def test_ellipticity_is_not_defined_for_love_waves():
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceProblem
    import torch
    periods = np.flip(1 / np.logspace(0.3, 1.3, 8)).copy()
    prob = DeviceProblem(2, 1.75, periods, wave="love", itype="phase+ellipticity")
    with pytest.raises(L.MhdError):
        prob.forward(torch.tensor(np.array([[0.02, 0.05, 0.3, 0.5, 1.0]]), device=prob.device))


@pytest.mark.parametrize("candidates", [2, 1])
def test_mh_steps_on_dispersion_and_ellipticity(oracle, candidates):
    """A 6-layer inversion of the phase-velocity curve and the ellipticity spectrum together (one
    likelihood over 2P data, each half with its own standard deviation): 80 steps of 64 chains
    against the oracle."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(5)
    nl, B, S = 6, 64, 80
    NPc = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    true_m = np.array([0.01, 0.02, 0.04, 0.06, 0.08, 0.10, 0.2, 0.3, 0.45, 0.6, 0.8, 1.0, 1.5])
    sd = np.concatenate([np.full(P, 0.025), np.full(P, 0.1)])
    obs = oracle.forward(true_m, nl, 1.75, periods, itype=3)[1] + sd * rng.standard_normal(2 * P)
    starts = np.tile(true_m, (B, 1)) * rng.uniform(0.9, 1.1, (B, NPc))
    starts = np.stack([oracle.sort_layers(s, nl) for s in starts])
    noise = {"std": sd}
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds,
                         itype="phase+ellipticity", candidates=candidates)
    ch = DeviceChains(prob, starts, 0.05)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    ost, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods, itype=3, nthreads=8)
    assert (ost == 0).all()
    ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])
    np.testing.assert_allclose(ch.logL.cpu().numpy(), ologL, rtol=1e-9)
    st = oracle.ChainState(starts, ologL, ovel, 0.05, bounds)
    idx = rng.integers(0, NPc, (S, B)).astype(np.int32)
    cz = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
    ua = rng.uniform(size=(S, B))
    dev = prob.device
    out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cz, device=dev),
                   torch.tensor(ua, device=dev), hist_every=40, want_pred_hist=True)
    acc = np.zeros((S, B), dtype=np.uint8)
    oracle.set_threads(8)
    for s in range(S):
        acc[s], _, _ = oracle.mh_step(st, idx[s], cz[s], ua[s], nl, 1.75, periods, obs, "normal",
                                      noise, itype=3)
    oracle.set_threads(1)
    assert (out["accepted"].cpu().numpy() == acc).all()
    np.testing.assert_array_equal(ch.m.cpu().numpy(), st.m)
    np.testing.assert_allclose(ch.logL.cpu().numpy(), st.logL, rtol=1e-9)
    np.testing.assert_allclose(ch.pred.cpu().numpy(), st.pred, rtol=1e-8, atol=1e-12)
    assert out["hist_pred"].shape == (2, B, 2 * P)
    assert acc.sum() > B


def test_mh_steps_on_ellipticity_alone(oracle):
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(6)
    nl, B, S = 3, 32, 60
    NPc = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, 20)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    obs = oracle.forward(true_m, nl, 1.75, periods, itype=4)[1] + 0.05 * rng.standard_normal(20)
    starts = np.tile(true_m, (B, 1)) * rng.uniform(0.9, 1.1, (B, NPc))
    starts = np.stack([oracle.sort_layers(s, nl) for s in starts])
    noise = {"std": 0.05}
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds, itype="ellipticity")
    ch = DeviceChains(prob, starts, 0.05)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    ost, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods, itype=4, nthreads=8)
    ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])
    st = oracle.ChainState(starts, ologL, ovel, 0.05, bounds)
    idx = rng.integers(0, NPc, (S, B)).astype(np.int32)
    cz = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
    ua = rng.uniform(size=(S, B))
    dev = prob.device
    out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cz, device=dev),
                   torch.tensor(ua, device=dev))
    for s in range(S):
        a, _, _ = oracle.mh_step(st, idx[s], cz[s], ua[s], nl, 1.75, periods, obs, "normal",
                                 noise, itype=4)
        assert (out["accepted"][s].cpu().numpy() == a).all(), s
    np.testing.assert_array_equal(ch.m.cpu().numpy(), st.m)
    np.testing.assert_allclose(ch.logL.cpu().numpy(), st.logL, rtol=1e-9)


def test_ellipticity_params_through_inversion(oracle, tmp_path):
    """EllipticityParams as the model of an Inversion: the joint datum, one chain, against the
    oracle driven with the same draws."""
    from mcmc_amd.inversion import Data, EllipticityParams, Inversion
    nl, P_ = 3, 20
    bounds = {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])}
    mp = EllipticityParams(nl, bounds, {"depth": 0.05, "vel_s": 0.05}, 1.75)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P_)).copy()
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    sd = np.concatenate([np.full(P_, 0.025), np.full(P_, 0.08)])
    rng = np.random.default_rng(11)
    obs = oracle.forward(true_m, nl, 1.75, periods, itype=3)[1] + sd * rng.standard_normal(2 * P_)
    # the drop-in's own forward model is the oracle's
    np.testing.assert_allclose(mp.forward_model(periods, true_m),
                               oracle.forward(true_m, nl, 1.75, periods, itype=3)[1], rtol=1e-8)
    noise = {"std": sd}
    data = Data(periods, obs, "normal", noise)
    data.model_true = true_m
    np.random.seed(3)
    inv = Inversion(data, mp, "normal", noise, n_burn=0, n_chunk=40, n_mcmc=40, n_thin=2,
                    n_chains=4, n_temps=1, set_starting_model=True, write_results=False)
    drawn = []
    orig = inv._draws

    def record(S):
        d = orig(S)
        drawn.append(d)
        return d
    inv._draws = record
    inv.random_walk(mp, "cauchy")
    starts = np.tile(true_m, (4, 1))
    _, vel, _ = oracle.forward_batch(starts, nl, 1.75, periods, itype=3)
    logL0 = np.array([oracle.loglik(obs, v, "normal", noise) for v in vel])
    st = oracle.ChainState(starts, logL0, vel, 0.05, mp.assemble_param_bounds())
    for idx, cz, xi in drawn:
        for s in range(idx.shape[0]):
            oracle.mh_step(st, idx[s], cz[s], xi[s], nl, 1.75, periods, obs, "normal", noise,
                           itype=3)
    np.testing.assert_array_equal(inv.state.m.cpu().numpy(), st.m)
    np.testing.assert_allclose(inv.state.logL.cpu().numpy(), st.logL, rtol=1e-9)
    assert st.counters[:, 0].sum() > 0


def test_joint_inversion_with_burn_in_rotation(oracle):
    """The joint dispersion + ellipticity likelihood with covariance accumulation during burn-in
    and rotated-frame proposals, refreshed every n_cov_chunk steps, after it
    (src/inversion/model.py:183-193, :263-293), through Inversion.random_walk: 8 chains against a
    host simulation that drives the oracle's step and numpy's SVD on the same cadence."""
    from mcmc_amd.inversion import Data, EllipticityParams, Inversion
    nl, P_, NCH = 3, 20, 8
    n_burn, n_cov_chunk, n_mcmc = 30, 10, 90
    bounds = {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])}
    mp = EllipticityParams(nl, bounds, {"depth": 0.05, "vel_s": 0.05}, 1.75)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P_)).copy()
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    sd = np.concatenate([np.full(P_, 0.025), np.full(P_, 0.08)])
    rng = np.random.default_rng(12)
    obs = oracle.forward(true_m, nl, 1.75, periods, itype=3)[1] + sd * rng.standard_normal(2 * P_)
    noise = {"std": sd}
    data = Data(periods, obs, "normal", noise)
    data.model_true = true_m
    np.random.seed(5)
    inv = Inversion(data, mp, "normal", noise, n_burn=n_burn, n_chunk=45, n_mcmc=n_mcmc,
                    n_cov_chunk=n_cov_chunk, n_thin=1, n_chains=NCH, n_temps=1,
                    set_starting_model=True, write_results=False)
    drawn = []
    orig = inv._draws

    def record(S):
        d = orig(S)
        drawn.append(d)
        return d
    inv._draws = record
    inv.random_walk(mp, "cauchy", rotate_params=True)
    idx, cz, xi = (np.concatenate(x) for x in zip(*drawn))
    assert idx.shape[0] == n_mcmc

    starts = np.tile(true_m, (NCH, 1))
    _, vel, _ = oracle.forward_batch(starts, nl, 1.75, periods, itype=3)
    logL0 = np.array([oracle.loglik(obs, v, "normal", noise) for v in vel])
    st = oracle.ChainState(starts, logL0, vel, 0.05, mp.assemble_param_bounds())
    refreshed = 0
    for s in range(n_mcmc):
        rot = 1 | (0 if s < n_burn else 2)
        if s >= n_burn + n_cov_chunk + 2 and s % n_cov_chunk == 0:
            oracle.cov_accumulate(st)
            U, sv, _ = np.linalg.svd(st.cov_sum / np.maximum(st.n_cov, 1)[:, None, None])
            st.rot[:] = U
            st.width[:] = np.sqrt(sv)
            rot = 2
            refreshed += 1
        oracle.mh_step(st, idx[s], cz[s], xi[s], nl, 1.75, periods, obs, "normal", noise,
                       itype=3, rotate=rot)
    assert refreshed == 4
    np.testing.assert_array_equal(inv.state.n_cov.cpu().numpy(), st.n_cov)
    np.testing.assert_array_equal(inv.state.counters.cpu().numpy(), st.counters)
    np.testing.assert_allclose(inv.state.m.cpu().numpy(), st.m, rtol=1e-8)
    np.testing.assert_allclose(inv.state.logL.cpu().numpy(), st.logL, rtol=1e-7)
    np.testing.assert_allclose(inv.state.cov_sum.cpu().numpy(), st.cov_sum, rtol=1e-7, atol=1e-16)
    assert st.counters[:, 0].sum() > NCH


def test_forward_ellipticity_of_mixed_layer_counts(oracle):
    """The joint datum for models of 2..12 layers padded to 12 in one batch."""
    import torch
    from mcmc_amd.engine import DeviceProblem
    rng = np.random.default_rng(44)
    NLMAX, B = 12, 192
    NPmax = 2 * NLMAX + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    nb = rng.integers(2, NLMAX + 1, B).astype(np.int32)
    rows = np.full((B, NPmax), 0.777)
    compact = []
    for b in range(B):
        n = int(nb[b])
        d = np.sort(rng.uniform(0.001, 0.15, n))
        v = np.sort(rng.uniform(0.1, 2.0, n + 1))
        rows[b, :n] = d
        rows[b, NLMAX:NLMAX + n + 1] = v
        compact.append(np.concatenate([d, v]))
    prob = DeviceProblem(NLMAX, 1.75, periods, itype="phase+ellipticity")
    vel, st = prob.forward(torch.tensor(rows, device=prob.device), n_layers_b=nb)
    vel, st = vel.cpu().numpy(), st.cpu().numpy()
    n_ok = 0
    for b in range(B):
        ost, ovel, _ = oracle.forward(compact[b], int(nb[b]), 1.75, periods, itype=3)
        assert st[b] == ost, b
        if ost == 0:
            n_ok += 1
            assert np.array_equal(vel[b, :P], ovel[:P]), b
            np.testing.assert_allclose(vel[b, P:], ovel[P:], rtol=1e-8, atol=1e-12)
    assert n_ok > B // 2
