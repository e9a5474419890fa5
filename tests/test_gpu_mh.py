"""GPU parity of the Metropolis-Hastings step: against traces of the reference's own
Model.perturb_params (tests/golden/mh_trace_*.json) and against the CPU oracle, through the
C ABI.  Parameters and accept/reject decisions must be identical; logL within 1e-9."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[4, 2, 1], ids=["four", "two", "one"])
def candidates(request):
    """Every test runs with four, two (the default) and one candidate per chain in flight."""
    from mcmc_amd import _lib as L, engine as E
    E.set_default_tuning(candidates=request.param)
    yield request.param
    E.set_default_tuning(candidates=0)


def _noise(g):
    return {k: (np.array(v) if isinstance(v, list) else v) for k, v in g["noise_params"].items()}


@pytest.mark.parametrize("case", ["3layer_normal", "6layer_normal", "6layer_asymlaplace",
                                  "3layer_individual_T2", "asymlaplace_WH01", "3layer_uniform"])
def test_reference_trace(case):
    """Traces of the reference's own Model.perturb_params; 3layer_uniform with its "uniform"
    proposal (src/inversion/model.py:205-207), the others with the Cauchy one."""
    import torch
    from mcmc_amd import _lib as L
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    g = load_golden(f"mh_trace_{case}.json")
    nl = g["n_layers"]
    prob = DeviceProblem(nl, g["vpvs"], g["periods"], g["obs"], g["noise_dist"], _noise(g),
                         g["bounds"])
    ch = DeviceChains(prob, np.array([g["start"]]), g["width0"],
                      individual_acceptance=g["individual_acceptance"])
    assert int(ch.init_likelihood()[0]) == 0
    tol = 1e-13 if g["noise_dist"] == "asym-laplace" else 0.0
    assert ch.logL.item() == pytest.approx(g["start_logL"], rel=tol, abs=0)
    S = len(g["steps"])
    dev = prob.device
    idx = torch.tensor(np.array(g["idx"], dtype=np.int32)[:, None], device=dev)
    uniform = g.get("proposal", "cauchy") == "uniform"
    draw = np.array(g["u_prop"])
    cauchy = torch.tensor((draw if uniform else np.tan(np.pi * (draw - 0.5)))[:, None], device=dev)
    u_acc = torch.tensor(np.array(g["u_acc"])[:, None], device=dev)
    out = ch.steps(idx, cauchy, u_acc, T=g["T"], hist_every=1,
                   rotate=L.PROPOSE_UNIFORM if uniform else 0)
    hist_m = out["hist_m"].cpu().numpy()[:, 0]
    hist_l = out["hist_logL"].cpu().numpy()[:, 0]
    status = out["status"].cpu().numpy()[:, 0]
    want_m = np.array([r["m"] for r in g["steps"]])
    want_l = np.array([r["logL"] for r in g["steps"]])
    np.testing.assert_array_equal(hist_m, want_m)            # bit exact
    np.testing.assert_allclose(hist_l, want_l, rtol=1e-9)    # north-star tolerance
    if g["noise_dist"] == "normal":
        np.testing.assert_array_equal(hist_l, want_l)
    # the reference draws the accept uniform only for valid proposals
    assert [2 if s == 0 else 1 for s in status] == [r["n_uniform_draws"] for r in g["steps"]]
    last = g["steps"][-1]
    cnt = ch.counters.cpu().numpy()[0]
    if g["individual_acceptance"]:
        ci = ch.counters_ind.cpu().numpy()[0]
        np.testing.assert_array_equal(ci[:, 0], last["n_acc"])
        np.testing.assert_array_equal(ci[:, 1], last["n_rej"])
        np.testing.assert_array_equal(ci[:, 2], last["n_bounds_err"])
    else:
        assert cnt[0] == last["n_acc"] and cnt[1] == last["n_rej"]
        assert cnt[2] == last["n_bounds_err"] and cnt[4] == last["n_fm_err"]
    np.testing.assert_array_equal(ch.pred.cpu().numpy()[0], g["final_pred"])


@pytest.mark.parametrize("proposal", ["cauchy", "uniform"])
@pytest.mark.parametrize("n_layers,P,B,S", [(3, 20, 64, 60), (6, 40, 256, 40)])
def test_batched_steps_vs_oracle(oracle, n_layers, P, B, S, proposal):
    """Many chains from prior draws (includes forward failures and bounds rejections), with
    either proposal of the reference."""
    import torch
    from mcmc_amd import _lib as L
    flag = L.PROPOSE_UNIFORM if proposal == "uniform" else 0
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(42 + n_layers)
    nl, NP = n_layers, 2 * n_layers + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    starts = []
    while len(starts) < B:
        x = oracle.sort_layers(rng.uniform(bounds[:, 0], bounds[:, 1]), nl)
        if oracle.forward(x, nl, 1.75, periods)[0] == 0:
            starts.append(x)
    starts = np.array(starts)
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    beta = rng.choice([1.0, 0.7, 0.3], B)
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds)
    ch = DeviceChains(prob, starts, 0.05, beta=beta)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    _, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods)
    ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])
    ost = oracle.ChainState(starts, ologL, ovel, 0.05, bounds)
    idx = rng.integers(0, NP, (S, B)).astype(np.int32)
    cauchy = rng.uniform(size=(S, B))
    if not flag:
        cauchy = np.tan(np.pi * (cauchy - 0.5))
    u_acc = rng.uniform(size=(S, B))
    dev = prob.device
    out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cauchy, device=dev),
                   torch.tensor(u_acc, device=dev), rotate=flag)
    acc = np.zeros((S, B), dtype=np.uint8)
    stat = np.zeros((S, B), dtype=np.int32)
    lit = 0
    for s in range(S):
        acc[s], stat[s], ne = oracle.mh_step(ost, idx[s], cauchy[s], u_acc[s], nl, 1.75, periods,
                                             obs, "normal", noise, beta=beta, rotate=flag)
        lit += ne
    np.testing.assert_array_equal(out["accepted"].cpu().numpy(), acc)
    np.testing.assert_array_equal(out["status"].cpu().numpy(), stat)
    np.testing.assert_array_equal(ch.m.cpu().numpy(), ost.m)
    np.testing.assert_allclose(ch.logL.cpu().numpy(), ost.logL, rtol=1e-9)
    np.testing.assert_allclose(ch.pred.cpu().numpy(), ost.pred, rtol=1e-8)
    np.testing.assert_array_equal(ch.counters.cpu().numpy(), ost.counters)
    assert int(ch.n_evals[0]) == lit
    assert (stat == 1).any()   # bounds rejections were exercised


def test_chains_that_accept_most_proposals_vs_oracle(oracle, candidates):
    """Hot chains (beta down to 1e-4) and wide noise: nearly every valid proposal is accepted, so
    the kernels run candidates that follow acceptances (several accepted steps settle in one
    round).  Three calls in a row on one schedule: the share of accepted proposals that steers the
    choice of candidates is carried from call to call.  Decisions, states, counters, histories
    and the literal evaluation count are the oracle's."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(2024)
    nl, P, B, S = 6, 40, 128, 30
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    starts = []
    while len(starts) < B:
        x = oracle.sort_layers(rng.uniform(bounds[:, 0], bounds[:, 1]), nl)
        if oracle.forward(x, nl, 1.75, periods)[0] == 0:
            starts.append(x)
    starts = np.array(starts)
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.2}
    beta = rng.choice([1.0, 0.1, 1.0e-2, 1.0e-4], B)
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds)
    ch = DeviceChains(prob, starts, 0.02, beta=beta)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    _, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods, nthreads=8)
    ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])
    ost = oracle.ChainState(starts, ologL, ovel, 0.02, bounds)
    dev = prob.device
    lit = n_acc = 0
    oracle.set_threads(8)
    for call in range(3):
        idx = rng.integers(0, NP, (S, B)).astype(np.int32)
        cauchy = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
        u_acc = rng.uniform(size=(S, B))
        out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cauchy, device=dev),
                       torch.tensor(u_acc, device=dev), hist_every=1)
        acc = np.zeros((S, B), dtype=np.uint8)
        stat = np.zeros((S, B), dtype=np.int32)
        hist = np.zeros((S, B, NP))
        for s in range(S):
            acc[s], stat[s], ne = oracle.mh_step(ost, idx[s], cauchy[s], u_acc[s], nl, 1.75,
                                                 periods, obs, "normal", noise, beta=beta)
            hist[s] = ost.m
            lit += ne
        n_acc += int(acc.sum())
        np.testing.assert_array_equal(out["accepted"].cpu().numpy(), acc)
        np.testing.assert_array_equal(out["status"].cpu().numpy(), stat)
        np.testing.assert_array_equal(out["hist_m"].cpu().numpy(), hist)
        np.testing.assert_array_equal(ch.m.cpu().numpy(), ost.m)
        np.testing.assert_allclose(ch.logL.cpu().numpy(), ost.logL, rtol=1e-9)
        np.testing.assert_array_equal(ch.pred.cpu().numpy(), ost.pred)
        np.testing.assert_array_equal(ch.counters.cpu().numpy(), ost.counters)
    oracle.set_threads(1)
    assert int(ch.n_evals[0]) == lit
    valid = int((ost.counters[:, 0] + ost.counters[:, 1] - ost.counters[:, 2]).sum())
    assert n_acc > 0.6 * valid, (n_acc, valid)   # the chains did accept most valid proposals


def test_steps_of_chains_in_low_velocity_zone_models_vs_oracle(oracle):
    """Chains that start at models with strong low-velocity zones: their curves jump downwards
    between periods, so the curve of the current state is a poor guide for the candidate's and
    the predicted search has to repair and resynchronise.  Decisions and states must still be
    those of the sequential algorithm."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(1660)
    nl, NP, P, B, S = 6, 13, 40, 128, 100
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    base = np.array([0.03, 0.0339, 0.0494, 0.0627, 0.0643, 0.1304,
                     0.1677, 0.526, 1.2108, 0.3184, 1.7798, 0.1555, 1.0089])
    starts = []
    while len(starts) < B:
        x = base.copy()
        x[:nl] *= 1.0 + 0.05 * rng.standard_normal(nl)
        x[nl:] *= 1.0 + 0.10 * rng.standard_normal(nl + 1)
        x = oracle.sort_layers(np.clip(x, bounds[:, 0], bounds[:, 1]), nl)
        if oracle.forward(x, nl, 1.75, periods)[0] == 0:
            starts.append(x)
    starts = np.array(starts)
    _, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods)
    assert ((np.diff(ovel, axis=1) < 0).any(axis=1)).sum() > B // 2
    # data near the family, so that a good share of the candidates is accepted
    obs = ovel[0] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.1}
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds)
    ch = DeviceChains(prob, starts, 0.02)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])
    ost = oracle.ChainState(starts, ologL, ovel, 0.02, bounds)
    idx = rng.integers(0, NP, (S, B)).astype(np.int32)
    cauchy = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
    u_acc = rng.uniform(size=(S, B))
    dev = prob.device
    out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cauchy, device=dev),
                   torch.tensor(u_acc, device=dev))
    acc = np.zeros((S, B), dtype=np.uint8)
    stat = np.zeros((S, B), dtype=np.int32)
    lit = 0
    for s in range(S):
        acc[s], stat[s], ne = oracle.mh_step(ost, idx[s], cauchy[s], u_acc[s], nl, 1.75, periods,
                                             obs, "normal", noise)
        lit += ne
    np.testing.assert_array_equal(out["accepted"].cpu().numpy(), acc)
    np.testing.assert_array_equal(out["status"].cpu().numpy(), stat)
    np.testing.assert_array_equal(ch.m.cpu().numpy(), ost.m)
    np.testing.assert_allclose(ch.logL.cpu().numpy(), ost.logL, rtol=1e-9)
    np.testing.assert_array_equal(ch.pred.cpu().numpy(), ost.pred)
    assert int(ch.n_evals[0]) == lit
    assert 0.1 < acc.mean() < 0.9


def test_scheduled_launches_do_not_change_the_chains(oracle, candidates):
    """With the scratch of mhd_chains_t.sched the library measures what each chain costs and, from
    the second call on, runs the expensive chains with four candidates in flight and the cheap
    ones with one.  States, counters and the literal evaluation count must be those of the same
    calls without it."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    if candidates != 2:
        pytest.skip("scheduling applies to the default number of candidates")
    rng = np.random.default_rng(5)
    nl, P, B, S = 6, 24, 192, 90
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    prob0 = DeviceProblem(nl, 1.75, periods)
    starts = np.empty((0, NP))
    while starts.shape[0] < B:
        cand = np.array([oracle.sort_layers(x, nl)
                         for x in rng.uniform(bounds[:, 0], bounds[:, 1], size=(2 * B, NP))])
        _, st = prob0.forward(torch.tensor(cand, device=prob0.device))
        starts = np.concatenate([starts, cand[st.cpu().numpy() == 0]])
    starts = starts[:B]
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
    dev = prob.device
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    results, n4_seen = [], 0
    for scheduled in (True, False):
        ch = DeviceChains(prob, starts, 0.05)
        if not scheduled:
            ch.sched = None
        assert int((ch.init_likelihood() != 0).sum()) == 0
        acc = []
        for a in range(0, S, 30):
            out = ch.steps(idx[a:a + 30], cauchy[a:a + 30], u_acc[a:a + 30], hist_every=10)
            acc.append(torch.cat([out["accepted"].flatten().to(torch.float64),
                                  out["hist_logL"].flatten()]))
            if scheduled:
                n4_seen = max(n4_seen, int(ch.sched[4 * B + 16:4 * B + 20].view(torch.int32)[0]))
                # whatever these chains really cost: a few are booked as six times dearer than they
                # were, so that the next call's schedule has chains to give four candidates to
                cost = ch.sched[:4 * B].view(torch.float32)
                assert bool((cost > 0).all())
                cost[:6] *= 6.0
        results.append((ch.m.clone(), ch.logL.clone(), ch.pred.clone(), ch.counters.clone(),
                        torch.cat(acc), int(ch.n_evals[0])))
    a, b = results
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    assert a[5] == b[5]
    assert n4_seen > 0   # the schedule really gave some chains four candidates


def test_scheduled_launch_of_more_chains_than_fit_at_once(oracle, candidates):
    """1100 chains are more than a GPU of 256 compute units holds at once with two waves each:
    such a call is a queue of one-chain workgroups in the order of the schedule (a workgroup of
    two chains would keep its place until both are done).  Same chains as without the schedule."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    if candidates != 2:
        pytest.skip("scheduling applies to the default number of candidates")
    rng = np.random.default_rng(19)
    nl, P, B, S = 3, 12, 1100, 120
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    prob0 = DeviceProblem(nl, 1.75, periods)
    starts = np.empty((0, NP))
    while starts.shape[0] < B:
        cand = np.array([oracle.sort_layers(x, nl)
                         for x in rng.uniform(bounds[:, 0], bounds[:, 1], size=(B, NP))])
        _, st = prob0.forward(torch.tensor(cand, device=prob0.device))
        starts = np.concatenate([starts, cand[st.cpu().numpy() == 0]])
    starts = starts[:B]
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
    dev = prob.device
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    results = []
    for scheduled in (True, False):
        ch = DeviceChains(prob, starts, 0.05)
        if not scheduled:
            ch.sched = None
        assert int((ch.init_likelihood() != 0).sum()) == 0
        acc = []
        for a in range(0, S, 60):   # the second call knows the costs and is made of two launches
            acc.append(ch.steps(idx[a:a + 60], cauchy[a:a + 60], u_acc[a:a + 60])["accepted"])
        results.append((ch.m.clone(), ch.logL.clone(), ch.pred.clone(), ch.counters.clone(),
                        torch.cat(acc), int(ch.n_evals[0])))
    a, b = results
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    assert a[5] == b[5]


def test_scheduled_launches_of_an_odd_number_of_chains(oracle, candidates):
    """67 chains: the last workgroup of two-candidate chains holds one chain only, whose other two
    waves have nothing of their own and run two more of its candidates from its second round on;
    in the calls of 60 steps (two launches) chains that are done hand their waves on as well.
    Same chains as without the schedule."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    if candidates != 2:
        pytest.skip("scheduling applies to the default number of candidates")
    rng = np.random.default_rng(23)
    nl, P, B, S = 4, 16, 67, 150
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    prob0 = DeviceProblem(nl, 1.75, periods)
    starts = np.empty((0, NP))
    while starts.shape[0] < B:
        cand = np.array([oracle.sort_layers(x, nl)
                         for x in rng.uniform(bounds[:, 0], bounds[:, 1], size=(2 * B, NP))])
        _, st = prob0.forward(torch.tensor(cand, device=prob0.device))
        starts = np.concatenate([starts, cand[st.cpu().numpy() == 0]])
    starts = starts[:B]
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
    dev = prob.device
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    results = []
    for scheduled in (True, False):
        ch = DeviceChains(prob, starts, 0.05)
        if not scheduled:
            ch.sched = None
        assert int((ch.init_likelihood() != 0).sum()) == 0
        acc = []
        for a, b in ((0, 10), (10, 30), (30, 90), (90, 150)):   # 10, 20, 60 and 60 steps
            out = ch.steps(idx[a:b], cauchy[a:b], u_acc[a:b], hist_every=5)
            acc.append(torch.cat([out["accepted"].flatten().to(torch.float64),
                                  out["status"].flatten().to(torch.float64),
                                  out["hist_logL"].flatten(), out["hist_m"].flatten()]))
        results.append((ch.m.clone(), ch.logL.clone(), ch.pred.clone(), ch.counters.clone(),
                        torch.cat(acc), int(ch.n_evals[0])))
    a, b = results
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    assert a[5] == b[5]


def test_scheduled_rotated_launches_do_not_change_the_chains(oracle, candidates):
    """The deadline of a scheduled call cuts chains between two steps of rotated-frame proposals
    with covariance accumulation: the second launch must go on exactly there (cov_sum, n_cov,
    params_prev, per-parameter counters and the history are booked per step)."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    if candidates != 2:
        pytest.skip("scheduling applies to the default number of candidates")
    rng = np.random.default_rng(11)
    nl, P, B, S = 4, 20, 128, 240   # calls of 120 steps: long enough to be made of two launches
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    prob0 = DeviceProblem(nl, 1.75, periods)
    starts = np.empty((0, NP))
    while starts.shape[0] < B:
        cand = np.array([oracle.sort_layers(x, nl)
                         for x in rng.uniform(bounds[:, 0], bounds[:, 1], size=(2 * B, NP))])
        _, st = prob0.forward(torch.tensor(cand, device=prob0.device))
        starts = np.concatenate([starts, cand[st.cpu().numpy() == 0]])
    starts = starts[:B]
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", {"std": 0.025}, bounds)
    dev = prob.device
    rot = torch.tensor(np.stack([np.linalg.qr(rng.standard_normal((NP, NP)))[0]
                                 for _ in range(B)]), device=dev)
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    u_acc = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    results, cut = [], 0
    for scheduled in (True, False):
        ch = DeviceChains(prob, starts, 0.04)
        if not scheduled:
            ch.sched = None
        ch.rot.copy_(rot)
        assert int((ch.init_likelihood() != 0).sum()) == 0
        hist = []
        for a in range(0, S, 120):
            out = ch.steps(idx[a:a + 120], cauchy[a:a + 120], u_acc[a:a + 120], rotate=3,
                           hist_every=8)
            hist.append(torch.cat([out["accepted"].flatten().to(torch.float64),
                                   out["status"].flatten().to(torch.float64),
                                   out["hist_logL"].flatten(), out["hist_m"].flatten()]))
            if scheduled:  # chains the second launch of the call found unfinished
                cut = max(cut, int(ch.sched[4 * B + 20:4 * B + 24].view(torch.int32)[0]))
        results.append((ch.m.clone(), ch.logL.clone(), ch.pred.clone(), ch.counters.clone(),
                        ch.cov_sum.clone(), ch.n_cov.clone(), ch.params_prev.clone(),
                        torch.cat(hist), int(ch.n_evals[0])))
    a, b = results
    for x, y in zip(a[:8], b[:8]):
        assert torch.equal(x, y)
    assert a[8] == b[8]
    assert cut > 0   # the deadline really cut chains in the middle of a call


def test_batched_rotated_steps_vs_oracle(oracle):
    """Rotated-frame proposals with covariance accumulation on every step, many steps per launch
    (the settling order of the steps matters for cov_sum / params_prev)."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(7)
    nl, P, B, S = 3, 16, 48, 50
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    starts = []
    while len(starts) < B:
        x = oracle.sort_layers(rng.uniform(bounds[:, 0], bounds[:, 1]), nl)
        if oracle.forward(x, nl, 1.75, periods)[0] == 0:
            starts.append(x)
    starts = np.array(starts)
    obs = oracle.forward(starts[0], nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    rot = np.stack([np.linalg.qr(rng.standard_normal((NP, NP)))[0] for _ in range(B)])
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", noise, bounds)
    ch = DeviceChains(prob, starts, 0.03)
    assert int((ch.init_likelihood() != 0).sum()) == 0
    dev = prob.device
    ch.rot.copy_(torch.tensor(rot, device=dev))
    _, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods)
    ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])
    ost = oracle.ChainState(starts, ologL, ovel, 0.03, bounds)
    ost.rot[:] = rot
    idx = rng.integers(0, NP, (S, B)).astype(np.int32)
    cauchy = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
    u_acc = rng.uniform(size=(S, B))
    out = ch.steps(torch.tensor(idx, device=dev), torch.tensor(cauchy, device=dev),
                   torch.tensor(u_acc, device=dev), rotate=3)
    acc = np.zeros((S, B), dtype=np.uint8)
    for s in range(S):
        acc[s], _, _ = oracle.mh_step(ost, idx[s], cauchy[s], u_acc[s], nl, 1.75, periods, obs,
                                      "normal", noise, rotate=3)
    np.testing.assert_array_equal(out["accepted"].cpu().numpy(), acc)
    np.testing.assert_allclose(ch.m.cpu().numpy(), ost.m, rtol=1e-12)
    np.testing.assert_array_equal(ch.n_cov.cpu().numpy(), ost.n_cov)
    np.testing.assert_allclose(ch.cov_sum.cpu().numpy(), ost.cov_sum, rtol=1e-9, atol=1e-18)
    np.testing.assert_allclose(ch.params_prev.cpu().numpy(), ost.params_prev, rtol=1e-12)
    np.testing.assert_array_equal(ch.counters.cpu().numpy(), ost.counters)
    assert acc.sum() > B and (ost.n_cov > 0).all()


def test_rotation_trace(oracle):
    """Covariance accumulation + rotated-frame proposals against the reference trace (the
    reference's matvec is BLAS, so parameters agree to rounding, not bit for bit)."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    g = load_golden("mh_trace_3layer_rotate.json")
    nl = g["n_layers"]
    prob = DeviceProblem(nl, g["vpvs"], g["periods"], g["obs"], g["noise_dist"], _noise(g),
                         g["bounds"])
    ch = DeviceChains(prob, np.array([g["start"]]), g["width0"])
    assert int(ch.init_likelihood()[0]) == 0
    dev = prob.device
    cauchy = np.tan(np.pi * (np.array(g["u_prop"]) - 0.5))
    for s, rec in enumerate(g["steps"]):
        n_step, thin = divmod(s, g["n_thin"])
        burn = n_step < g["n_burn"]
        upd = (n_step >= g["n_burn"] + g["n_cov_chunk"] + 2 and n_step % g["n_cov_chunk"] == 0
               and thin == 0)
        rot = 1
        if upd:
            ch.cov_accumulate()
            cov = (ch.cov_sum[0] / ch.n_cov[0]).cpu().numpy()
            U, sv, _ = np.linalg.svd(cov)
            ch.rot[0] = torch.tensor(U, device=dev)
            ch.width[0] = torch.tensor(np.sqrt(sv), device=dev)
            rot = 0
        if not burn:
            rot |= 2
        ch.steps(torch.tensor([[g["idx"][s]]], dtype=torch.int32, device=dev),
                 torch.tensor([[cauchy[s]]], dtype=torch.float64, device=dev),
                 torch.tensor([[g["u_acc"][s]]], dtype=torch.float64, device=dev), rotate=rot)
        assert int(ch.n_cov[0]) == rec["n_cov"], s
        np.testing.assert_allclose(ch.m[0].cpu().numpy(), rec["m"], rtol=1e-9, err_msg=str(s))
        assert int(ch.counters[0, 0]) == rec["n_acc"]
    np.testing.assert_allclose(ch.cov_sum[0].cpu().numpy(), g["final_cov_sum"], rtol=1e-9,
                               atol=1e-18)


@pytest.mark.parametrize("search", [2, 1])
def test_every_recorded_curve_is_the_forward_of_its_parameters(candidates, search):
    """The curves a chain keeps after each step are bit for bit those of mhd_forward_batch on
    the parameters it keeps, whichever search and however many candidates made them: 1024 chains
    of 6 layers x 40 periods over calls of 48 and 60 steps.  Step 59 of the second call takes
    chain 956 to a model whose last period has a halving abscissa 5e-9 (relative) from a root of
    the period equation, where the value is rounding alone: the one- and the two-abscissa
    evaluations of the equation have to round alike for the paths to agree on its sign."""
    import torch
    from mcmc_amd import engine as E
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    rng = np.random.default_rng(9123)
    # (the draws are taken in the order of scratch/soak_sched.py, whose configuration 123 this is)
    nl = int(rng.integers(2, 8))
    P = int(rng.choice([8, 16, 24, 40]))
    B = int(rng.choice([64, 65, 67, 96, 129, 200, 255, 256, 383, 512, 700, 1024, 1100]))
    itype = str(rng.choice(["phase", "phase", "group", "phase+ellipticity"]))
    rotate = int(rng.choice([0, 0, 3]))
    calls = [int(x) for x in rng.choice([3, 7, 12, 20, 30, 47, 48, 60, 90, 130],
                                        size=int(rng.integers(3, 6)))]
    assert (nl, P, B, itype, rotate, calls) == (5, 40, 1024, "phase", 0, [48, 60, 12, 48, 3])
    NP = 2 * nl + 1
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
    prob0 = DeviceProblem(nl, 1.75, periods)
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
    obs = true + 0.03 * rng.standard_normal(P)
    prob = DeviceProblem(nl, 1.75, periods, obs, "normal", {"std": 0.03}, bounds)
    dev = prob.device
    rng.uniform()
    S = sum(calls)
    idx = torch.tensor(rng.integers(0, NP, (S, B)).astype(np.int32), device=dev)
    cz = torch.tensor(np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5)), device=dev)
    ua = torch.tensor(rng.uniform(size=(S, B)), device=dev)
    calls = calls[:2]
    reached = False
    E.set_default_tuning(search=search)
    try:
        for scheduled in (True, False):
            ch = DeviceChains(prob, starts, 0.05)
            if not scheduled:
                ch.sched = None
            assert int(ch.init_likelihood().abs().sum()) == 0
            a = 0
            for n in calls:
                o = ch.steps(idx[a:a + n], cz[a:a + n], ua[a:a + n], hist_every=1,
                             want_pred_hist=True)
                a += n
                E.set_default_tuning(search=3 - search, candidates=0)   # the other search
                pr, st = prob.forward(o["hist_m"].reshape(n * B, NP))
                E.set_default_tuning(search=search, candidates=candidates)
                assert int(st.abs().sum()) == 0
                bad = (pr != o["hist_pred"].reshape(n * B, P)).any(1).reshape(n, B).nonzero()
                assert bad.shape[0] == 0, f"(step, chain) {bad[:5].tolist()}"
                if n == 60:
                    reached = reached or o["hist_m"][59, 956, 4].item() == float.fromhex(
                        "0x1.1a5a21d769b5ep-3")
    finally:
        E.set_default_tuning(search=0)
    assert reached, "the chains no longer pass through the model this test is about"
