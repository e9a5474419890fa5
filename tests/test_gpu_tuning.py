"""Acceptance-rate step-size tuning (src/inversion/model.py:538-556) in the batched path, and the
per-step records of Inversion.random_walk: against traces of the reference's own
Model.stepsize_tuning (tests/golden/mh_trace_3layer_tune*.json, made by tools/make_golden_mh.py)
and against the CPU oracle driven with the same draws."""
import glob

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["3layer_tune", "3layer_tune_individual"])
def test_reference_trace_with_stepsize_tuning(case):
    """mhd_mh_steps + mhd_stepsize_tuning reproduce the reference's states and proposal widths
    bit for bit: one rate per chain (every width scaled alike) and one per parameter."""
    import torch
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    g = load_golden(f"mh_trace_{case}.json")
    prob = DeviceProblem(g["n_layers"], g["vpvs"], g["periods"], g["obs"], g["noise_dist"],
                         g["noise_params"], g["bounds"])
    ch = DeviceChains(prob, np.array([g["start"]]), g["width0"],
                      individual_acceptance=g["individual_acceptance"])
    assert int(ch.init_likelihood()[0]) == 0
    dev = prob.device
    S, every = len(g["steps"]), g["tune_every"]
    idx = torch.tensor(np.array(g["idx"], dtype=np.int32)[:, None], device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (np.array(g["u_prop"]) - 0.5))[:, None], device=dev)
    u_acc = torch.tensor(np.array(g["u_acc"])[:, None], device=dev)
    want_m = np.array([r["m"] for r in g["steps"]])
    want_w = np.array([r["width"] for r in g["steps"]])
    changed = 0
    for a in range(0, S, every):
        b = min(S, a + every)
        out = ch.steps(idx[a:b], cauchy[a:b], u_acc[a:b], T=g["T"], hist_every=1)
        np.testing.assert_array_equal(out["hist_m"].cpu().numpy()[:, 0], want_m[a:b])
        before = ch.width.cpu().numpy()[0].copy()
        np.testing.assert_array_equal(before, want_w[b - 2] if b - a > 1 else before)
        if b - a == every:
            ch.stepsize_tuning()
        after = ch.width.cpu().numpy()[0]
        np.testing.assert_array_equal(after, want_w[b - 1])
        changed += int((after != before).any())
    assert changed >= 3
    last = g["steps"][-1]
    if g["individual_acceptance"]:
        np.testing.assert_array_equal(ch.counters_ind.cpu().numpy()[0][:, 0], last["n_acc"])
    else:
        assert ch.counters.cpu().numpy()[0][0] == last["n_acc"]


@pytest.mark.parametrize("individual", [False, True], ids=["one_rate", "per_parameter"])
@pytest.mark.parametrize("sink", ["async", "sync"])
def test_inversion_tuning_and_per_step_records_vs_oracle(oracle, tmp_path, individual, sink):
    """Inversion(tune_stepsize=4): 16 chains, 24 outer steps of 2 MH steps, tuning after every
    fourth outer step of the 12-step burn-in.  States, widths and the records in the result
    files (model_params, logL, acc_rate, bounds_err, fm_err, proposal_width after every outer
    step) equal those of the oracle driven with the same draws."""
    from mcmc_amd.inversion import Data, DispersionCurveParams, Inversion
    from mcmc_amd.inversion.inversion import rates_from_counters
    nl, P, B, n_mcmc, nt = 3, 20, 16, 24, 2
    NP = 2 * nl + 1
    bounds = {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])}
    mp = DispersionCurveParams(nl, bounds, {"depth": 0.05, "vel_s": 0.05}, 1.75)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    rng = np.random.default_rng(8)
    obs = oracle.forward(true_m, nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    data = Data(periods, obs, "normal", noise)
    np.random.seed(21)
    inv = Inversion(data, mp, "normal", noise, n_burn=12, n_chunk=8, n_mcmc=n_mcmc, n_thin=nt,
                    n_chains=B, n_temps=1, individual_acceptance=individual, tune_stepsize=4,
                    out_filename="tune", out_dir=str(tmp_path) + "/", async_sink=sink == "async")
    start = inv.state.m.cpu().numpy().copy()
    drawn = []
    orig = inv._draws

    def record(S):
        d = orig(S)
        drawn.append(d)
        return d
    inv._draws = record
    inv.random_walk(mp, "cauchy")
    idx = np.concatenate([d[0] for d in drawn])
    cz = np.concatenate([d[1] for d in drawn])
    xi = np.concatenate([d[2] for d in drawn])
    assert idx.shape == (n_mcmc * nt, B)

    _, vel, _ = oracle.forward_batch(start, nl, 1.75, periods)
    logL0 = np.array([oracle.loglik(obs, v, "normal", noise) for v in vel])
    st = oracle.ChainState(start, logL0, vel, 0.05, mp.assemble_param_bounds())
    cind = np.zeros((B, NP, 5), dtype=np.int64)
    hist = {k: [] for k in ("m", "logL", "cnt", "width")}
    rows = np.arange(B)
    for step in range(n_mcmc):
        for s in range(step * nt, (step + 1) * nt):
            acc, status, _ = oracle.mh_step(st, idx[s], cz[s], xi[s], nl, 1.75, periods, obs,
                                            "normal", noise)
            cind[rows, idx[s], 0] += acc
            cind[rows, idx[s], 1] += 1 - acc
            cind[rows, idx[s], 2] += status == 1
            cind[rows, idx[s], 4] += status >= 3
        if step < 12 and (step + 1) % 4 == 0:
            if individual:
                st.width[:] = oracle.stepsize_tuning(st.width, cind[:, :, 0], cind[:, :, 1])
            else:
                st.width[:] = oracle.stepsize_tuning(st.width, st.counters[:, 0],
                                                     st.counters[:, 1])
        hist["m"].append(st.m.copy())
        hist["logL"].append(st.logL.copy())
        hist["cnt"].append(cind.copy() if individual else st.counters.copy())
        hist["width"].append(st.width.copy())
    np.testing.assert_array_equal(inv.state.m.cpu().numpy(), st.m)
    np.testing.assert_array_equal(inv.state.width.cpu().numpy(), st.width)
    assert (st.width != 0.05).any()
    np.testing.assert_array_equal(inv.state.counters.cpu().numpy(), st.counters)
    files = sorted(glob.glob(str(tmp_path) + "/results-tune-*.npz"))
    assert len(files) == 3
    got = {k: np.concatenate([np.load(f)[k] for f in files])
           for k in ("model_params", "logL", "acc_rate", "bounds_err", "physics_err", "fm_err",
                     "proposal_width")}
    np.testing.assert_array_equal(got["model_params"], np.array(hist["m"]))
    np.testing.assert_allclose(got["logL"], np.array(hist["logL"]), rtol=1e-9)
    np.testing.assert_array_equal(got["proposal_width"], np.array(hist["width"]))
    want = rates_from_counters(np.array(hist["cnt"]))
    for k, v in want.items():
        assert got[k].shape == ((n_mcmc, B, NP) if individual else (n_mcmc, B))
        np.testing.assert_array_equal(got[k], v)
    assert got["acc_rate"].max() > 0 and got["bounds_err"].max() > 0
    f0 = np.load(files[0])
    assert f0["cov_mat"].shape == (B, NP, NP) and (f0["chain"] == np.arange(B)).all()


def test_per_step_covariance_records_and_out_filename_defaults(oracle, tmp_path, monkeypatch):
    """record_diagnostics="full" in a run with rotate_params: cov_mat after every step equals
    cov_sum / n_cov of the oracle's covariance accumulation; and an empty or missing
    out_filename makes up a name from the clock, as the reference does
    (src/inversion/inversion.py:83-86), while write_results=False writes nothing."""
    from mcmc_amd.inversion import Data, DispersionCurveParams, Inversion
    nl, P, B, n_mcmc = 3, 20, 8, 12
    NP = 2 * nl + 1
    bounds = {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])}
    mp = DispersionCurveParams(nl, bounds, {"depth": 0.05, "vel_s": 0.05}, 1.75)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    obs = oracle.forward(true_m, nl, 1.75, periods)[1]
    noise = {"std": 0.025}
    data = Data(periods, obs, "normal", noise)
    data.model_true = true_m
    np.random.seed(3)
    import types
    import mcmc_amd.inversion.inversion as inv_mod
    monkeypatch.setattr(inv_mod, "time", types.SimpleNamespace(time=lambda: 1234567.8))
    for name in ("", None):
        inv = Inversion(data, mp, "normal", noise, n_burn=100, n_chunk=n_mcmc, n_mcmc=n_mcmc,
                        n_cov_chunk=50, n_chains=B, n_temps=1, set_starting_model=True,
                        out_filename=name, out_dir=str(tmp_path / repr(name)) + "/",
                        record_diagnostics="full")
        assert inv.out_filename == "1234567"
    drawn = []
    orig = inv._draws

    def record(S):
        d = orig(S)
        drawn.append(d)
        return d
    inv._draws = record
    inv.random_walk(mp, "cauchy", rotate_params=True)
    idx, cz, xi = (np.concatenate([d[k] for d in drawn]) for k in range(3))
    starts = np.tile(true_m, (B, 1))
    _, vel, _ = oracle.forward_batch(starts, nl, 1.75, periods)
    logL0 = np.array([oracle.loglik(obs, v, "normal", noise) for v in vel])
    st = oracle.ChainState(starts, logL0, vel, 0.05, mp.assemble_param_bounds())
    covs = []
    for s in range(n_mcmc):
        oracle.mh_step(st, idx[s], cz[s], xi[s], nl, 1.75, periods, obs, "normal", noise,
                       rotate=1)
        covs.append(st.cov_sum / np.maximum(st.n_cov, 1)[:, None, None])
    f = np.load(glob.glob(str(tmp_path / repr(None)) + "/results-1234567-*.npz")[0])
    assert f["cov_mat"].shape == (n_mcmc, B, NP, NP)
    np.testing.assert_allclose(f["cov_mat"], np.array(covs), rtol=1e-12, atol=1e-300)
    assert np.abs(f["cov_mat"][-1]).max() > 0
    inv = Inversion(data, mp, "normal", noise, n_burn=0, n_chunk=4, n_mcmc=4, n_chains=2,
                    n_temps=1, set_starting_model=True, write_results=False,
                    out_dir=str(tmp_path / "none") + "/")
    inv.random_walk(mp, "cauchy")
    assert inv.out_filename is None and not (tmp_path / "none").exists()
