"""Batches that mix layer counts (rows padded to the largest model) and the joint phase + group
datum, against the oracle.  Neither has a call site or a vector in the reference: parity with
disba is unpinned, these tests pin the HIP path to the CPU restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NLMAX, P = 12, 40
NP = 2 * NLMAX + 1


def _padded_models(rng, B, lo_layers=2):
    """Prior draws with 2..12 layers each, in padded rows; the unused slots hold a marker."""
    nb = rng.integers(lo_layers, NLMAX + 1, B).astype(np.int32)
    rows = np.full((B, NP), 0.777)
    compact = []
    for b in range(B):
        n = int(nb[b])
        d = np.sort(rng.uniform(0.001, 0.15, n))
        v = rng.uniform(0.1, 2.0, n + 1)
        rows[b, :n] = d
        rows[b, NLMAX:NLMAX + n + 1] = v
        compact.append(np.concatenate([d, v]))
    return nb, rows, compact


def _slots(n):
    """Slots of the padded row that a model of n layers uses, in the order of its compact
    parameter vector."""
    return np.concatenate([np.arange(n), NLMAX + np.arange(n + 1)])


@pytest.mark.parametrize("search", [2, 1])
@pytest.mark.parametrize("wave,mode", [("rayleigh", 0), ("love", 0), ("love", 1), ("rayleigh", 2)])
def test_forward_of_mixed_layer_counts(oracle, search, wave, mode):
    """Models of 2..12 layers padded to 12 in one batch: Rayleigh and Love waves, fundamental and
    higher modes."""
    import torch
    from mcmc_amd.engine import DeviceProblem
    rng = np.random.default_rng(12)
    B = 384 if mode == 0 else 192
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    nb, rows, compact = _padded_models(rng, B)
    prob = DeviceProblem(NLMAX, 1.75, periods, search=search, wave=wave, mode=mode)
    ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
    vel, st = prob.forward(torch.tensor(rows, device=prob.device), n_evals=ne, n_layers_b=nb)
    vel, st = vel.cpu().numpy(), st.cpu().numpy()
    lit = 0
    n_ok = 0
    for b in range(B):
        ost, ovel, one = oracle.forward(compact[b], int(nb[b]), 1.75, periods,
                                        wave=int(wave == "love"), mode=mode)
        lit += one
        assert st[b] == ost, b
        if ost == 0:
            n_ok += 1
            assert np.array_equal(vel[b], ovel), b
    assert n_ok > B // 3 and int(ne[0]) == lit
    assert set(nb.tolist()) == set(range(2, NLMAX + 1))


def test_mh_steps_of_mixed_layer_counts_love_first_overtone(oracle):
    """Chains of 2..12 layers on Love-wave first-overtone data (curves that end at the mode's
    cut-off period carry zeros): 40 steps against the oracle run on each compact vector."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(21)
    B, S = 48, 40
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * NLMAX + [[0.1, 2.0, 1.9]] * (NLMAX + 1))
    kind = dict(wave=1, mode=1)
    nb_l, rows_l, comp_l = [], [], []
    while len(rows_l) < B:
        nb, rows, compact = _padded_models(rng, 1)
        if oracle.forward(compact[0], int(nb[0]), 1.75, periods, **kind)[0] == 0:
            nb_l.append(int(nb[0]))
            rows_l.append(rows[0])
            comp_l.append(compact[0])
    nb, rows = np.array(nb_l, dtype=np.int32), np.array(rows_l)
    obs = oracle.forward(comp_l[0], nb_l[0], 1.75, periods, **kind)[1] \
        + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    prob = DeviceProblem(NLMAX, 1.75, periods, obs, "normal", noise, bounds, wave="love", mode=1)
    ch = DeviceChains(prob, rows, 0.05, n_layers_b=nb)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    j = rng.integers(0, 2 ** 31, (S, B))
    idx_c = (j % (2 * nb[None, :] + 1)).astype(np.int32)       # index into the compact vector
    idx = np.where(idx_c < nb[None, :], idx_c, NLMAX + idx_c - nb[None, :]).astype(np.int32)
    cz = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
    ua = rng.uniform(size=(S, B))
    dev = prob.device
    out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cz, device=dev),
                   torch.tensor(ua, device=dev))
    acc = out["accepted"].cpu().numpy()
    m, logL = ch.m.cpu().numpy(), ch.logL.cpu().numpy()
    n_acc = 0
    for b in range(B):
        n = nb_l[b]
        sl = _slots(n)
        v0 = oracle.forward(comp_l[b], n, 1.75, periods, **kind)[1]
        st = oracle.ChainState(comp_l[b][None, :], [oracle.loglik(obs, v0, "normal", noise)],
                               v0[None, :], 0.05, bounds[sl])
        for s in range(S):
            a, _, _ = oracle.mh_step(st, idx_c[s, b:b + 1], cz[s, b:b + 1], ua[s, b:b + 1], n,
                                     1.75, periods, obs, "normal", noise, **kind)
            assert a[0] == acc[s, b], (b, s)
        np.testing.assert_array_equal(m[b, sl], st.m[0])
        assert logL[b] == pytest.approx(st.logL[0], rel=1e-9)
        n_acc += int(st.counters[0, 0])
    assert n_acc > 0


@pytest.mark.parametrize("candidates", [2, 1, 4])
def test_mh_steps_of_mixed_layer_counts(oracle, candidates):
    """60 MH steps of chains with 2..12 layers in one launch: states, decisions and counters of
    every chain equal those of the oracle run on its compact parameter vector."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(31)
    B, S = 96, 60
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * NLMAX + [[0.1, 2.0, 1.9]] * (NLMAX + 1))
    # starts whose forward model succeeds
    nb_l, rows_l, comp_l = [], [], []
    while len(nb_l) < B:
        nb, rows, compact = _padded_models(rng, B)
        for b in range(B):
            if oracle.forward(compact[b], int(nb[b]), 1.75, periods)[0] == 0 and len(nb_l) < B:
                nb_l.append(nb[b]); rows_l.append(rows[b]); comp_l.append(compact[b])
    nb, rows = np.array(nb_l, dtype=np.int32), np.array(rows_l)
    true = comp_l[0]
    obs = oracle.forward(true, int(nb[0]), 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    prob = DeviceProblem(NLMAX, 1.75, periods, obs, "normal", noise, bounds, candidates=candidates)
    ch = DeviceChains(prob, rows, 0.05, n_layers_b=nb)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    # the perturbed parameter: a draw among the chain's own parameters, mapped to its slot
    idx_c = np.stack([rng.integers(0, 2 * nb + 1) for _ in range(S)]).astype(np.int32)
    idx = np.empty_like(idx_c)
    for b in range(B):
        idx[:, b] = _slots(int(nb[b]))[idx_c[:, b]]
    cz = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
    ua = rng.uniform(size=(S, B))
    dev = prob.device
    out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cz, device=dev),
                   torch.tensor(ua, device=dev))
    acc = out["accepted"].cpu().numpy()
    m, logL, cnt = ch.m.cpu().numpy(), ch.logL.cpu().numpy(), ch.counters.cpu().numpy()
    n_acc = 0
    for b in range(B):
        n = int(nb[b])
        sl = _slots(n)
        _, v0, _ = oracle.forward(comp_l[b], n, 1.75, periods)
        st = oracle.ChainState(comp_l[b][None, :], [oracle.loglik(obs, v0, "normal", noise)],
                               v0[None, :], 0.05, bounds[sl])
        for s in range(S):
            a, _, _ = oracle.mh_step(st, idx_c[s, b:b + 1], cz[s, b:b + 1], ua[s, b:b + 1], n,
                                     1.75, periods, obs, "normal", noise)
            assert a[0] == acc[s, b], (b, s)
        np.testing.assert_array_equal(m[b, sl], st.m[0])
        unused = np.setdiff1d(np.arange(NP), sl)
        assert (m[b, unused] == 0.777).all()            # padding never changes
        assert logL[b] == pytest.approx(st.logL[0], rel=1e-9)
        np.testing.assert_array_equal(cnt[b], st.counters[0])
        n_acc += int(st.counters[0, 0])
    assert n_acc > B


@pytest.mark.parametrize("search", [2, 1])
def test_forward_phase_and_group_datum(oracle, search):
    import torch
    from mcmc_amd.engine import DeviceProblem
    rng = np.random.default_rng(8)
    nl, B = 6, 128
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    d = np.sort(rng.uniform(0.001, 0.15, (B, nl)), axis=1)
    v = np.sort(rng.uniform(0.1, 2.0, (B, nl + 1)), axis=1)
    params = np.concatenate([d, v], axis=1)
    prob = DeviceProblem(nl, 1.75, periods, itype="phase+group", search=search)
    vel, st = prob.forward(torch.tensor(params, device=prob.device))
    vel, st = vel.cpu().numpy(), st.cpu().numpy()
    ost, ovel, _ = oracle.forward_batch(params, nl, 1.75, periods, itype=2, nthreads=8)
    assert vel.shape == (B, 2 * P) and (st == ost).all()
    ok = ost == 0
    assert ok.sum() > B // 2
    assert np.array_equal(vel[ok, :P], ovel[ok, :P])                    # phase: lattice points
    np.testing.assert_allclose(vel[ok, P:], ovel[ok, P:], rtol=1e-8)     # group: a quotient
    # the halves are the single curves
    ph = oracle.forward_batch(params, nl, 1.75, periods, itype=0, nthreads=8)[1]
    assert np.array_equal(ovel[ok, :P], ph[ok])


@pytest.mark.parametrize("candidates", [2, 1])
def test_mh_steps_on_phase_and_group_datum(oracle, candidates):
    """A 6-layer inversion of phase and group velocity together (one likelihood over 2P data):
    80 steps of 64 chains against the oracle."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(3)
    nl, B, S = 6, 64, 80
    NPc = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    true_m = np.array([0.01, 0.02, 0.04, 0.06, 0.08, 0.10, 0.2, 0.3, 0.45, 0.6, 0.8, 1.0, 1.5])
    obs = oracle.forward(true_m, nl, 1.75, periods, itype=2)[1] + 0.025 * rng.standard_normal(2 * P)
    starts = np.tile(true_m, (B, 1)) * rng.uniform(0.9, 1.1, (B, NPc))
    starts = np.stack([oracle.sort_layers(s, nl) for s in starts])
    noise = {"std": np.concatenate([np.full(P, 0.025), np.full(P, 0.05)])}  # per datum
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds, itype="phase+group",
                         candidates=candidates)
    ch = DeviceChains(prob, starts, 0.05)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    ost, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods, itype=2, nthreads=8)
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
                                      noise, itype=2)
    oracle.set_threads(1)
    assert (out["accepted"].cpu().numpy() == acc).all()
    np.testing.assert_array_equal(ch.m.cpu().numpy(), st.m)
    np.testing.assert_allclose(ch.logL.cpu().numpy(), st.logL, rtol=1e-9)
    np.testing.assert_allclose(ch.pred.cpu().numpy(), st.pred, rtol=1e-8)
    assert out["hist_pred"].shape == (2, B, 2 * P)
    np.testing.assert_array_equal(out["hist_pred"][-1].cpu().numpy(), ch.pred.cpu().numpy())
    assert acc.sum() > B


def test_inversion_with_layer_counts(oracle):
    """Inversion.random_walk with chains of 2..6 layers: starts are prior draws of each chain's
    own layer count, only its own parameters are perturbed, padding stays untouched."""
    from mcmc_amd.inversion import Data, DispersionCurveParams, Inversion
    nlmax, P_ = 6, 20
    bounds = {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])}
    mp = DispersionCurveParams(nlmax, bounds, {"depth": 0.05, "vel_s": 0.05}, 1.75)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P_)).copy()
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    obs = oracle.forward(true_m, 3, 1.75, periods)[1]
    noise = {"std": 0.025}
    data = Data(periods, obs, "normal", noise)
    counts = np.array([2, 3, 4, 5, 6, 3, 3, 2], dtype=np.int32)
    np.random.seed(4)
    inv = Inversion(data, mp, "normal", noise, n_burn=0, n_chunk=50, n_mcmc=50, n_chains=1,
                    n_temps=None, layer_counts=None, write_results=False)
    assert inv.layer_counts is None
    np.random.seed(4)
    inv = Inversion(data, mp, "normal", noise, n_burn=0, n_chunk=25, n_mcmc=50, n_thin=2,
                    n_chains=8, n_temps=1, layer_counts=counts, write_results=False)
    start = inv.state.m.cpu().numpy().copy()
    inv.random_walk(mp, "cauchy")
    end = inv.state.m.cpu().numpy()
    cnt = inv.state.counters.cpu().numpy()
    assert (cnt[:, 0] + cnt[:, 1] == 100).all() and cnt[:, 0].sum() > 0
    moved = 0
    for b, n in enumerate(counts):
        used = np.concatenate([np.arange(n), nlmax + np.arange(n + 1)])
        unused = np.setdiff1d(np.arange(2 * nlmax + 1), used)
        assert np.array_equal(start[b, unused], end[b, unused])
        moved += not np.array_equal(start[b, used], end[b, used])
        st, vel, _ = oracle.forward(end[b, used], int(n), 1.75, periods)
        assert st == 0 and np.array_equal(vel, inv.state.pred[b].cpu().numpy())
    assert moved >= len(counts) // 2
