"""The CPU oracle against every known-answer vector the reference holds.

Forward model: the four 100-point disba output curves embedded at
src/plotting/plot_dispersion_curve.py:1509-1530, 1601-1925 and the three layer
tables at :1533-1548 (tests/golden/disba_curves.json).  The literals are
8-digit prints computed from 8-digit prints of the parameters, so they cannot
pin 1e-8 by themselves; the gates are the ones the data supports:
  * <= 2e-6 relative on all 400 values (the search stops at 1e-6 relative),
  * every printed value lies on the dc/2**16 lattice anchored at the curve's
    first value (sequential seeding + pure halving),
  * per curve, >= 50 % of values agree to within 0.2 lattice units and every
    disagreement is a whole number of lattice steps (a halving decision flipped
    by the rounding of the printed inputs).
"""
import numpy as np
import pytest

from conftest import load_golden

DC = 0.005
UNIT = DC / 65536.0


def test_layer_tables(oracle):
    g = load_golden("disba_curves.json")
    # tables 0..2 belong to m1..m3 (same vs values)
    for tab, cur in zip(g["layer_tables"], g["curves"][:3]):
        d, a, b, rho = oracle.layer_table(cur["params"], 2, 1.75)
        np.testing.assert_allclose(d, tab["thickness"], atol=6e-9)
        np.testing.assert_allclose(a, tab["vp"], atol=6e-9)
        np.testing.assert_allclose(b, tab["vs"], atol=6e-9)
        np.testing.assert_allclose(rho, tab["rho"], atol=6e-9)


@pytest.mark.parametrize("ic", range(4))
def test_embedded_disba_curves(oracle, ic):
    g = load_golden("disba_curves.json")
    cur = g["curves"][ic]
    periods = np.flip(1 / np.logspace(0, 1.1, 100)).copy()
    st, vel, ne = oracle.forward(cur["params"], g["n_layers"], g["vpvs"], periods)
    ref = np.array(cur["velocity"])
    assert st == 0
    rel = np.abs(vel - ref) / ref
    assert rel.max() <= 2e-6
    # printed values on the lattice
    lat = (ref - ref[0]) / UNIT
    assert np.abs(lat - np.round(lat)).max() < 0.2
    # our values exactly on the lattice anchored at our own first value
    mine = (vel - vel[0]) / UNIT
    assert np.abs(mine - np.round(mine)).max() < 1e-6
    d = (vel - ref) / UNIT
    near = np.abs(d) < 0.2
    assert near.mean() >= 0.5
    assert np.abs(d - np.round(d)).max() < 0.2
    # literal algorithm: about 19 secular evaluations per period
    assert 17.0 < ne / 100.0 < 21.0


def test_numpy_sum_order(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 5, 7, 8, 9, 20, 40, 60, 100, 127, 128, 129, 200, 256, 1000):
        for _ in range(200):
            a = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4)
            assert oracle.npsum(a) == np.sum(a), n


def test_misc_helpers(oracle):
    g = load_golden("mh_misc.json")
    s = g["sort_layers"]
    np.testing.assert_array_equal(oracle.sort_layers(s["in"], 2), s["out"])
    vp = g["vel_p"]
    d, a, b, rho = oracle.layer_table([0.01, 0.02] + vp["vs"], 2, 1.75)
    np.testing.assert_array_equal(a, vp["vp"])
    # numpy's vectorised float64 power may differ from libm pow() in the last bit
    np.testing.assert_allclose(rho, vp["rho"], rtol=4e-16)
    for rec in g["likelihood"]:
        obs, pred = rec["obs"], rec["pred"]
        assert oracle.loglik(obs, pred, "normal", {"std": 0.025}) == rec["normal_scalar"]
        assert oracle.loglik(obs, pred, "normal", {"std": np.array(rec["std_vec"])}) == \
            rec["normal_vector"]
        al = oracle.loglik(obs, pred, "asym-laplace",
                           {"lambd_scale": 1.3, "lambd": np.array(rec["lambd"]), "kappa": 0.75})
        assert al == pytest.approx(rec["asym_laplace_scalar_kappa"], rel=1e-13)
        al = oracle.loglik(obs, pred, "asym-laplace",
                           {"lambd_scale": 1.3, "lambd": np.array(rec["lambd"]),
                            "kappa": np.array(rec["kappa"]), "lambd_max": 60.0})
        assert al == pytest.approx(rec["asym_laplace_vector_kappa_clipped"], rel=1e-13)


def _replay(oracle, g):
    """Drive orc_mh_step over a golden trace; yields per-step comparisons."""
    nl, vpvs = g["n_layers"], g["vpvs"]
    periods, obs = np.array(g["periods"]), np.array(g["obs"])
    NP = 2 * nl + 1
    noise = {k: (np.array(v) if isinstance(v, list) else v) for k, v in g["noise_params"].items()}
    st = oracle.ChainState(np.array([g["start"]]), [g["start_logL"]], np.array([g["start_pred"]]),
                           g["width0"], g["bounds"])
    uniform = g.get("proposal", "cauchy") == "uniform"
    # the "uniform" proposal (src/inversion/model.py:205-207) takes the draw as it is
    cauchy = np.array(g["u_prop"]) if uniform else np.tan(np.pi * (np.array(g["u_prop"]) - 0.5))
    ind = g["individual_acceptance"]
    cnt_ind = np.zeros((4, NP))
    n_thin = g["n_thin"]
    for s, rec in enumerate(g["steps"]):
        n_step, thin = divmod(s, n_thin)
        rot = 0
        if g["rotate"]:
            burn = n_step < g["n_burn"]
            upd = (n_step >= g["n_burn"] + g["n_cov_chunk"] + 2
                   and n_step % g["n_cov_chunk"] == 0 and thin == 0)
            if upd:
                oracle.cov_accumulate(st)
                cov = st.cov_sum[0] / st.n_cov[0]
                U, sv, _ = np.linalg.svd(cov)
                st.rot[0] = U
                st.width[0] = np.sqrt(sv)
                rot = 0
            else:
                rot = 1
            if not burn:
                rot |= 2
        acc, status, _ = oracle.mh_step(st, [g["idx"][s]], [cauchy[s]], [g["u_acc"][s]], nl, vpvs,
                                        periods, obs, g["noise_dist"], noise, T=g["T"],
                                        rotate=rot | (4 if uniform else 0))
        k = g["idx"][s]
        cnt_ind[0, k] += acc[0]
        cnt_ind[1, k] += 1 - acc[0]
        cnt_ind[2, k] += status[0] == 1
        cnt_ind[3, k] += status[0] >= 3
        # Model.stepsize_tuning after every tune_every-th step (the traces that have it)
        if g.get("tune_every") and (s + 1) % g["tune_every"] == 0:
            if ind:
                st.width[:] = oracle.stepsize_tuning(st.width, cnt_ind[0][None], cnt_ind[1][None])
            else:
                st.width[:] = oracle.stepsize_tuning(st.width, st.counters[:, 0], st.counters[:, 1])
        yield s, rec, st, acc[0], status[0], (cnt_ind if ind else None)


@pytest.mark.parametrize("case", ["3layer_normal", "6layer_normal", "6layer_asymlaplace",
                                  "3layer_individual_T2", "asymlaplace_WH01", "3layer_tune",
                                  "3layer_tune_individual", "3layer_uniform"])
def test_mh_trace_bit_exact(oracle, case):
    """State after every step equals the reference's own Model.perturb_params, bit for bit; in
    the two traces with Model.stepsize_tuning (one rate per chain, one per parameter) the
    proposal widths as well.  3layer_uniform: the reference's "uniform" proposal."""
    g = load_golden(f"mh_trace_{case}.json")
    n = 0
    widths = set()
    for s, rec, st, acc, status, cnt_ind in _replay(oracle, g):
        np.testing.assert_array_equal(st.m[0], rec["m"], err_msg=f"step {s}")
        if g.get("tune_every"):
            np.testing.assert_array_equal(st.width[0], rec["width"], err_msg=f"step {s}")
            widths.update(rec["width"])
        if "asymlaplace" in case:
            assert st.logL[0] == pytest.approx(rec["logL"], rel=1e-13), s
        else:
            assert st.logL[0] == rec["logL"], s
        # the reference draws the accept uniform only for a valid proposal
        assert rec["n_uniform_draws"] == (2 if status == 0 else 1)
        if cnt_ind is None:
            assert st.counters[0, 0] == rec["n_acc"] and st.counters[0, 1] == rec["n_rej"]
            assert st.counters[0, 2] == rec["n_bounds_err"]
            assert st.counters[0, 4] == rec["n_fm_err"]
        else:
            np.testing.assert_array_equal(cnt_ind[0], rec["n_acc"])
            np.testing.assert_array_equal(cnt_ind[1], rec["n_rej"])
            np.testing.assert_array_equal(cnt_ind[2], rec["n_bounds_err"])
        n += 1
    assert n == len(g["steps"])
    np.testing.assert_array_equal(st.pred[0], g["final_pred"])
    if g.get("tune_every"):
        assert max(widths) > 0.05                       # widened ...
        assert case != "3layer_tune_individual" or min(widths) < 0.05  # ... and narrowed


@pytest.mark.parametrize("case", ["3layer_rotate", "3layer_uniform_rotate"])
def test_mh_trace_rotation(oracle, case):
    """Covariance accumulation is bit exact; the rotated proposal (BLAS matvec in the
    reference) agrees to rounding, so the trace is compared at 1e-9.  Both proposals."""
    g = load_golden(f"mh_trace_{case}.json")
    for s, rec, st, acc, status, _ in _replay(oracle, g):
        assert st.n_cov[0] == rec["n_cov"], s
        np.testing.assert_allclose(st.m[0], rec["m"], rtol=1e-9, err_msg=f"step {s}")
        np.testing.assert_allclose(st.width[0], rec["width"], rtol=1e-9)
        assert st.logL[0] == pytest.approx(rec["logL"], rel=1e-7)
        assert st.counters[0, 0] == rec["n_acc"]
    np.testing.assert_allclose(st.cov_sum[0], g["final_cov_sum"], rtol=1e-9, atol=1e-18)
    np.testing.assert_allclose(np.abs(st.rot[0]), np.abs(g["final_rot"]), rtol=1e-6, atol=1e-9)
