"""GPU tests of the host mirror of the reference API (mcmc_amd.inversion)."""
import glob
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _mp(n_layers, **kw):
    from mcmc_amd.inversion import DispersionCurveParams
    bounds = {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])}
    return DispersionCurveParams(n_layers, bounds, {"depth": 0.05, "vel_s": 0.05}, 1.75, **kw)


def test_forward_model_drop_in(oracle):
    from mcmc_amd.inversion import DispersionError
    mp = _mp(3)
    periods = np.flip(1 / np.logspace(0.3, 1.3, 20))
    m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    np.testing.assert_array_equal(mp.forward_model(periods, m),
                                  oracle.forward(m, 3, 1.75, periods)[1])
    # a prior draw for which the sequential search leaves its range
    rng = np.random.default_rng(103)
    for _ in range(2000):
        x = np.concatenate([np.sort(rng.uniform(0.001, 0.15, 3)), rng.uniform(0.1, 2.0, 4)])
        if oracle.forward(x, 3, 1.75, periods)[0] != 0:
            with pytest.raises(DispersionError):
                mp.forward_model(periods, x)
            break
    else:
        pytest.skip("no failing draw found")


@pytest.mark.parametrize("case", ["3layer_normal", "3layer_individual_T2", "3layer_rotate",
                                  "3layer_uniform", "3layer_uniform_rotate"])
def test_model_perturb_params_drop_in(case):
    """Model.perturb_params driven exactly like the reference's: np.random is patched to serve
    the trace's draws in the order the reference consumes them.  Both proposal distributions
    of the reference."""
    from mcmc_amd.inversion import Data, Model
    g = load_golden(f"mh_trace_{case}.json")
    nl = g["n_layers"]
    mp = _mp(nl)
    data = Data(np.array(g["periods"]), np.array(g["obs"]), g["noise_dist"], g["noise_params"])
    model = Model(mp, g["noise_dist"], g["noise_params"], g["n_cov_chunk"],
                  g["individual_acceptance"])
    mp.model_params = np.array(g["start"])
    model.data_pred = mp.forward_model(data.periods, mp.model_params)
    model.logL = Model.get_likelihood(data, model.data_pred, g["noise_dist"], g["noise_params"])
    assert model.logL == g["start_logL"]

    class Feeder:
        s = 0
        n = 0

        def randint(self, n):
            return int(g["idx"][self.s])

        def uniform(self, *a, **k):
            self.n += 1
            return float(g["u_prop"][self.s]) if self.n == 1 else float(g["u_acc"][self.s])

    f = Feeder()
    saved = (np.random.randint, np.random.uniform, np.random.get_state, np.random.set_state)
    np.random.randint, np.random.uniform = f.randint, f.uniform
    np.random.get_state = lambda: f.n
    np.random.set_state = lambda n: setattr(f, "n", n)
    try:
        for s, rec in enumerate(g["steps"]):
            n_step, thin = divmod(s, g["n_thin"])
            model.perturb_params(data, g.get("proposal", "cauchy"), n_step, g["n_burn"], 500,
                                 thin, T=g["T"],
                                 rotate_params=g["rotate"])
            assert f.n == rec["n_uniform_draws"], s
            f.s += 1
            f.n = 0
            if g["rotate"]:
                np.testing.assert_allclose(mp.model_params, rec["m"], rtol=1e-9)
                assert model.n_cov == rec["n_cov"]
            else:
                np.testing.assert_array_equal(mp.model_params, rec["m"])
                assert model.logL == rec["logL"]
            ar = model.acceptance_rate
            np.testing.assert_array_equal(np.asarray(ar["n_acc"]), rec["n_acc"])
            np.testing.assert_array_equal(np.asarray(ar["n_bounds_err"]), rec["n_bounds_err"])
            np.testing.assert_allclose(np.asarray(ar["acc_rate"]), rec["acc_rate"], rtol=1e-15)
    finally:
        np.random.randint, np.random.uniform, np.random.get_state, np.random.set_state = saved


@pytest.mark.parametrize("proposal", ["cauchy", "uniform"])
def test_inversion_random_walk_matches_oracle(oracle, tmp_path, proposal):
    """Inversion.random_walk on supplied draws: chains, thinning, chunked output; with either
    proposal distribution of the reference (any other name is refused)."""
    from mcmc_amd.inversion import Data, Inversion
    flag = 4 if proposal == "uniform" else 0
    nl, P = 3, 20
    mp = _mp(nl)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    rng = np.random.default_rng(0)
    obs = oracle.forward(true_m, nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(P)
    noise = {"std": 0.025}
    data = Data(periods, obs, "normal", noise)
    np.random.seed(5)
    inv = Inversion(data, mp, "normal", noise, n_burn=0, n_chunk=20, n_mcmc=40, n_thin=3,
                    n_chains=1, out_filename="t1", out_dir=str(tmp_path) + "/")
    start = inv.state.m.cpu().numpy().copy()
    drawn = []
    orig = inv._draws

    def record(S):
        d = orig(S)
        drawn.append(d)
        return d
    inv._draws = record
    with pytest.raises(ValueError):
        inv.random_walk(mp, "normal")
    inv.random_walk(mp, proposal)
    idx = np.concatenate([d[0] for d in drawn])
    cz = np.concatenate([d[1] for d in drawn])
    xi = np.concatenate([d[2] for d in drawn])
    assert idx.shape == (120, 1)
    assert (cz >= 0).all() == (proposal == "uniform")   # the draw itself, not its tangent
    _, vel, _ = oracle.forward_batch(start, nl, 1.75, periods)
    st = oracle.ChainState(start, [oracle.loglik(obs, vel[0], "normal", noise)], vel, 0.05,
                           mp.assemble_param_bounds())
    hist = []
    for s in range(120):
        oracle.mh_step(st, idx[s], cz[s], xi[s], nl, 1.75, periods, obs, "normal", noise,
                       rotate=flag)
        if s % 3 == 2:
            hist.append(st.m[0].copy())
    np.testing.assert_array_equal(inv.state.m.cpu().numpy(), st.m)
    files = sorted(glob.glob(str(tmp_path) + "/results-t1-*.npz"))
    assert len(files) == 2 and os.path.exists(str(tmp_path) + "/input-t1.npz")
    got = np.concatenate([np.load(f)["model_params"][:, 0] for f in files])
    np.testing.assert_array_equal(got, np.array(hist))
    with pytest.raises(PermissionError):
        inv.define_input_dataset(data, mp, "normal", noise)


def test_inversion_tempering_and_rotation_run(oracle, tmp_path):
    """Working parallel tempering (betas are a permutation of the ladder, beta==1 chains are
    recorded) and the rotation schedule on several chains."""
    from mcmc_amd.inversion import Data, Inversion
    nl, P = 3, 20
    mp = _mp(nl)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    obs = oracle.forward(true_m, nl, 1.75, periods)[1]
    noise = {"std": 0.025}
    data = Data(periods, obs, "normal", noise)
    np.random.seed(1)
    inv = Inversion(data, mp, "normal", noise, n_burn=5, n_chunk=10, n_mcmc=30, n_cov_chunk=5,
                    n_thin=2, n_chains=8, beta_spacing_factor=1.3, out_filename="pt",
                    out_dir=str(tmp_path) + "/")
    ladder = np.sort(inv.betas.copy())
    assert (ladder[-2:] == 1).all()
    inv.random_walk(mp, "cauchy", rotate_params=True)
    np.testing.assert_array_equal(np.sort(inv.betas), ladder)
    assert inv.swap_prop > 0
    np.testing.assert_array_equal(inv.state.beta.cpu().numpy(), inv.betas)
    assert int(inv.state.n_cov.min()) > 0
    rot = inv.state.rot.cpu().numpy()
    for R in rot:                      # refreshed rotation matrices are orthogonal
        np.testing.assert_allclose(R @ R.T, np.eye(7), atol=1e-10)
    f = np.load(sorted(glob.glob(str(tmp_path) + "/results-pt-*.npz"))[0])
    assert f["model_params"].shape[0] == 10 and f["model_params"].shape[2] == 7
    chains = inv.chains
    assert len(chains) == 8 and chains[0].data_pred.shape == (P,)


def test_the_binding_shown_to_maintainers_works(oracle):
    """The ctypes stub printed in INTEGRATION.md (the reference-side binding a maintainer would
    add) is executed as it stands, only the library's path made absolute."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    start = text.index("# src/inversion/_mhd.py")
    code = text[start:text.index("```", start)]
    code = code.replace('"mcmc_amd/lib/libmcmc_amd.so"',
                        repr(os.path.join(root, "mcmc_amd", "lib", "libmcmc_amd.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    nl = 3
    periods = np.flip(1 / np.logspace(0.3, 1.3, 20)).copy()
    rng = np.random.default_rng(3)
    params = np.concatenate([np.sort(rng.uniform(0.001, 0.15, (16, nl)), axis=1),
                             np.sort(rng.uniform(0.1, 2.0, (16, nl + 1)), axis=1)], axis=1)
    vel, st = ns["forward_batch"](params, nl, 1.75, periods)
    ost, ovel, _ = oracle.forward_batch(params, nl, 1.75, periods)
    assert (st == ost).all() and (ost == 0).any()
    assert np.array_equal(vel[ost == 0], ovel[ost == 0])


def test_device_spellings_and_geometric_ladder(oracle):
    """DeviceProblem takes "cuda", torch.device("cuda") and "cuda:0" alike (the index is made
    explicit before mhd_init sees it); a tempered run on the non-default geometric ladder keeps
    its betas a permutation of that ladder and exchanges with the cold chains."""
    import torch
    from mcmc_amd.engine import DeviceProblem
    from mcmc_amd.inversion import Data, Inversion
    periods = np.flip(1 / np.logspace(0.3, 1.3, 20)).copy()
    for dev in ("cuda", torch.device("cuda"), "cuda:0", None):
        prob = DeviceProblem(3, 1.75, periods, device=dev)
        assert prob.device == torch.device("cuda", torch.cuda.current_device())
    with pytest.raises(Exception):
        DeviceProblem(3, 1.75, periods, device="cpu")
    mp = _mp(3)
    true_m = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])
    obs = oracle.forward(true_m, 3, 1.75, periods)[1]
    noise = {"std": 0.025}
    data = Data(periods, obs, "normal", noise)
    data.model_true = true_m
    np.random.seed(2)
    inv = Inversion(data, mp, "normal", noise, n_burn=0, n_chunk=20, n_mcmc=20, n_thin=2,
                    n_chains=4, n_temps=16, beta_spacing_factor=1.15, set_starting_model=True,
                    beta_ladder="geometric", write_results=False)
    ladder = inv.get_betas(1.15, 16)
    np.testing.assert_allclose(ladder[:6], [1, 1, 1, 1, 1 / 1.15, 1 / 1.15 ** 2], rtol=1e-15)
    inv.random_walk(mp, "cauchy")
    got = inv.betas.reshape(4, 16)
    np.testing.assert_array_equal(np.sort(got, axis=1)[:, ::-1], np.tile(ladder, (4, 1)))
    assert inv.swap_acc > 0
