#!/usr/bin/env python3
"""Generate golden traces of the reference's Metropolis-Hastings half.

Runs in the build container only (the reference never travels).  The reference
modules src/inversion/{model,model_params,inversion}.py are imported from
/root/reference with empty stand-in modules for the packages that are not
installed (disba, numba, xarray, h5py, dask); the only thing replaced is the
third-party forward call, which is routed to the CPU oracle
(oracle/surf96_oracle.c) so that the trace pins the MH logic itself:
normalise -> pick index -> Cauchy perturb -> denormalise -> sort_layers ->
bounds -> forward -> likelihood -> accept -> counters / covariance / rotation.

np.random.randint / np.random.uniform are wrapped so that step s consumes
idx[s], u_prop[s] and (only when the proposal was valid, as the reference
does) u_acc[s].

Output: tests/golden/mh_trace_<case>.json  (inputs + per-step expected state).
"""
import ctypes
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class DispersionError(Exception):
        pass

    class TypingError(Exception):
        pass

    mod("disba", PhaseDispersion=None)
    mod("disba._exception", DispersionError=DispersionError)
    mod("numba")
    mod("numba.core")
    mod("numba.core.errors", TypingError=TypingError)
    mod("xarray")
    mod("h5py")
    mod("dask")
    mod("dask.dataframe")
    mod("dask.distributed", Client=None)
    return DispersionError


DispersionError = _stub_modules()
sys.path.insert(0, "/root/reference/src")
from inversion.model import Model  # noqa: E402
from inversion.model_params import DispersionCurveParams  # noqa: E402
from inversion.inversion import Inversion  # noqa: E402

_lib = ctypes.CDLL(os.path.join(ROOT, "oracle/_build/libsurf96_oracle.so"))
_dp = ctypes.POINTER(ctypes.c_double)
_lib.orc_forward.argtypes = [_dp, ctypes.c_int, ctypes.c_double, _dp, ctypes.c_int, ctypes.c_int,
                             ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, _dp,
                             ctypes.POINTER(ctypes.c_longlong)]


def oracle_forward(n_layers, vpvs):
    def fwd(periods, m):
        periods = np.ascontiguousarray(periods, dtype=np.float64)
        m = np.ascontiguousarray(m, dtype=np.float64)
        vel = np.zeros(len(periods))
        st = _lib.orc_forward(m.ctypes.data_as(_dp), n_layers, vpvs, periods.ctypes.data_as(_dp),
                              len(periods), 0, 0, 0, 0.005, 0.025, vel.ctypes.data_as(_dp), None)
        if st != 0:
            raise DispersionError("no root")
        return vel
    return fwd


class Data:
    def __init__(self, periods, data_obs):
        self.periods = periods
        self.data_obs = data_obs
        self.n_data = len(periods)


class DrawFeeder:
    """Serves idx[s], u_prop[s], u_acc[s] in the order the reference asks."""

    def __init__(self, idx, u_prop, u_acc):
        self.idx, self.u_prop, self.u_acc = idx, u_prop, u_acc
        self.s = 0
        self.n_uniform = 0

    def randint(self, n):
        return int(self.idx[self.s])

    def uniform(self, *a, **k):
        assert not a and not k
        self.n_uniform += 1
        return float(self.u_prop[self.s]) if self.n_uniform == 1 else float(self.u_acc[self.s])

    def next_step(self):
        used = self.n_uniform
        self.s += 1
        self.n_uniform = 0
        return used


def run_case(name, n_layers, bounds, periods, true_m, noise_dist, noise_params, n_steps, seed,
             rotate=False, n_burn=0, n_cov_chunk=200, n_thin=1, individual=False, start=None,
             T=1, tune_every=0, proposal="cauchy"):
    if ONLY and name not in ONLY:
        return
    vpvs = 1.75
    rng = np.random.default_rng(seed)
    mp = DispersionCurveParams(n_layers, bounds, {"depth": 0.05, "vel_s": 0.05}, vpvs)
    fwd = oracle_forward(n_layers, vpvs)
    mp.forward_model = fwd
    true_curve = fwd(periods, np.array(true_m))
    obs = true_curve + 0.025 * rng.standard_normal(len(periods))
    data = Data(periods, obs)
    model = Model(mp, noise_dist, noise_params, n_cov_chunk, individual)
    NP = mp.n_model_params
    b = model.param_bounds
    # start: a valid prior draw
    if start is None:
        while True:
            start = rng.uniform(b[:, 0], b[:, 1])
            start = mp.sort_layers(start.copy())
            try:
                pred = fwd(periods, start)
                break
            except DispersionError:
                pass
    else:
        start = np.array(start, dtype=float)
        pred = fwd(periods, start)
    mp.model_params = start.copy()
    model.logL = Model.get_likelihood(data, pred, noise_dist, noise_params)
    model.data_pred = pred

    total = n_steps * n_thin
    idx = rng.integers(0, NP, size=total)
    u_prop = rng.uniform(size=total)
    u_acc = rng.uniform(size=total)
    feeder = DrawFeeder(idx, u_prop, u_acc)
    saved = (np.random.randint, np.random.uniform)
    np.random.randint, np.random.uniform = feeder.randint, feeder.uniform
    steps = []
    try:
        for s in range(n_steps):
            for t in range(n_thin):
                model.perturb_params(data, proposal, s, n_burn, 500, t, T=T, rotate_params=rotate)
                used = feeder.next_step()
                # acceptance-rate step-size tuning (src/inversion/model.py:538-556; the reference's
                # own call, after every step, is commented out at :257): here after every
                # tune_every-th step
                if tune_every and (s * n_thin + t + 1) % tune_every == 0:
                    model.stepsize_tuning(s)
                ar = model.acceptance_rate
                rec = {
                    "m": mp.model_params.tolist(),
                    "logL": float(model.logL),
                    "n_uniform_draws": used,
                    "n_acc": np.asarray(ar["n_acc"]).tolist(),
                    "n_rej": np.asarray(ar["n_rej"]).tolist(),
                    "n_bounds_err": np.asarray(ar["n_bounds_err"]).tolist(),
                    "n_fm_err": np.asarray(ar["n_fm_err"]).tolist(),
                    "acc_rate": np.asarray(ar["acc_rate"]).tolist(),
                }
                if tune_every:
                    rec["width"] = np.asarray(model.proposal_width).tolist()
                if rotate:
                    rec["n_cov"] = int(model.n_cov)
                    rec["width"] = np.asarray(model.proposal_width).tolist()
                    rec["rot_trace"] = float(np.trace(model.rotation_matrix))
                steps.append(rec)
    finally:
        np.random.randint, np.random.uniform = saved
    out = {
        "case": name,
        "reference": "src/inversion/model.py:146-257 driven step by step; forward = CPU oracle",
        "n_layers": n_layers,
        "vpvs": vpvs,
        "bounds": b.tolist(),
        "width0": model_params_width(mp),
        "periods": periods.tolist(),
        "obs": obs.tolist(),
        "noise_dist": noise_dist,
        "noise_params": {k: (np.asarray(v).tolist()) for k, v in noise_params.items()},
        "start": start.tolist(),
        "start_pred": pred.tolist(),
        "start_logL": float(Model.get_likelihood(data, pred, noise_dist, noise_params)),
        "n_burn": n_burn,
        "n_cov_chunk": n_cov_chunk,
        "n_thin": n_thin,
        "rotate": bool(rotate),
        "individual_acceptance": bool(individual),
        "T": T,
        "tune_every": tune_every,
        "proposal": proposal,
        "idx": idx.tolist(),
        "u_prop": u_prop.tolist(),
        "u_acc": u_acc.tolist(),
        "steps": steps,
        "final_pred": np.asarray(model.data_pred).tolist(),
    }
    if rotate:
        out["final_cov_sum"] = model.cov_mat_sum.tolist()
        out["final_rot"] = model.rotation_matrix.tolist()
    path = os.path.join(ROOT, "tests/golden", f"mh_trace_{name}.json")
    json.dump(out, open(path, "w"))
    acc = steps[-1]["n_acc"]
    print(name, "steps", total, "n_acc", acc, "n_bounds_err", steps[-1]["n_bounds_err"],
          "n_fm_err", steps[-1]["n_fm_err"], "->", path)


def model_params_width(mp):
    return np.asarray(mp.assemble_proposal_width(), dtype=float).tolist()


def misc_fixture():
    """Spot values of the stateless helpers."""
    out = {}
    mp = DispersionCurveParams(
        2, {"depth": np.array([0.001, 0.10]),
            "vel_s": np.array([[0.1, 0.7], [0.1, 0.7], [1.0, 2.0]])},
        {"depth": 0.05, "vel_s": 0.05}, 1.75)
    out["assemble_param_bounds_2layer"] = mp.assemble_param_bounds().tolist()
    out["assemble_proposal_width_2layer"] = np.asarray(mp.assemble_proposal_width()).tolist()
    out["sort_layers"] = {"in": [0.063, 0.010, 0.4, 0.2, 1.7],
                          "out": mp.sort_layers(np.array([0.063, 0.010, 0.4, 0.2, 1.7])).tolist()}
    vs = np.array([0.20183109, 0.63001643, 1.64596118])
    vp = mp.get_vel_p(vs)
    out["vel_p"] = {"vs": vs.tolist(), "vp": vp.tolist(), "rho": mp.get_density(vp).tolist()}
    rng = np.random.default_rng(7)
    lik = []
    for n in (5, 8, 20, 40, 60, 100, 128, 200):
        obs = rng.uniform(0.2, 0.9, n)
        pred = obs + 0.02 * rng.standard_normal(n)
        d = Data(np.arange(n), obs)
        std_vec = rng.uniform(0.01, 0.05, n)
        lam = rng.uniform(20.0, 80.0, n)
        kap = rng.uniform(0.5, 1.5, n)
        lik.append({
            "obs": obs.tolist(), "pred": pred.tolist(), "std_vec": std_vec.tolist(),
            "lambd": lam.tolist(), "kappa": kap.tolist(),
            "normal_scalar": float(Model.get_likelihood(d, pred, "normal", {"std": 0.025})),
            "normal_vector": float(Model.get_likelihood(d, pred, "normal", {"std": std_vec})),
            "asym_laplace_scalar_kappa": float(Model.get_likelihood(
                d, pred, "asym-laplace", {"lambd_scale": 1.3, "lambd": lam.copy(), "kappa": 0.75})),
            "asym_laplace_vector_kappa_clipped": float(Model.get_likelihood(
                d, pred, "asym-laplace",
                {"lambd_scale": 1.3, "lambd": lam.copy(), "kappa": kap, "lambd_max": 60.0})),
        })
    out["likelihood"] = lik
    # get_betas as the reference computes it (first ceil(n/4) entries are 0)
    betas = {}
    for n in (1, 2, 4, 8, 64):
        inv = Inversion.__new__(Inversion)
        inv.n_chains = n
        betas[str(n)] = np.asarray(inv.get_betas(1.15), dtype=float).tolist()
    out["get_betas_factor_1.15"] = betas
    path = os.path.join(ROOT, "tests/golden/mh_misc.json")
    json.dump(out, open(path, "w"))
    print("->", path)


ONLY = set(sys.argv[1:])  # case names to (re)generate; all when none is given


def main():
    p20 = np.flip(1 / np.logspace(0.3, 1.3, 20)).copy()
    p40 = np.flip(1 / np.logspace(0.3, 1.3, 40)).copy()
    b3 = {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])}
    run_case("3layer_normal", 3, b3, p20, [0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0],
             "normal", {"std": 0.025}, n_steps=400, seed=1)
    run_case("6layer_normal", 6, b3, p40,
             [0.01, 0.02, 0.04, 0.06, 0.08, 0.10, 0.2, 0.3, 0.45, 0.6, 0.8, 1.0, 1.5],
             "normal", {"std": 0.025}, n_steps=400, seed=2,
             start=[0.012, 0.022, 0.041, 0.063, 0.085, 0.11, 0.21, 0.31, 0.44, 0.62, 0.79, 1.02,
                    1.45])
    lam = np.linspace(30.0, 70.0, 40)
    run_case("6layer_asymlaplace", 6, b3, p40,
             [0.01, 0.02, 0.04, 0.06, 0.08, 0.10, 0.2, 0.3, 0.45, 0.6, 0.8, 1.0, 1.5],
             "asym-laplace", {"lambd_scale": 1.0, "lambd": lam, "kappa": 0.8, "lambd_max": 60.0},
             n_steps=300, seed=3,
             start=[0.012, 0.022, 0.041, 0.063, 0.085, 0.11, 0.21, 0.31, 0.44, 0.62, 0.79, 1.02,
                    1.45])
    run_case("3layer_individual_T2", 3, b3, p20, [0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0],
             "normal", {"std": 0.025}, n_steps=200, seed=4, individual=True, T=2.0)
    run_case("3layer_rotate", 3, b3, p20, [0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0],
             "normal", {"std": 0.025}, n_steps=60, seed=5, rotate=True, n_burn=10, n_cov_chunk=5,
             n_thin=2)
    # Model.stepsize_tuning every 25 steps: one rate for the chain, and one per parameter
    run_case("3layer_tune", 3, b3, p20, [0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0],
             "normal", {"std": 0.025}, n_steps=300, seed=11, tune_every=25)
    run_case("3layer_tune_individual", 3, b3, p20, [0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0],
             "normal", {"std": 0.025}, n_steps=400, seed=12, individual=True, tune_every=40)
    # the reference's "uniform" proposal as it stands (src/inversion/model.py:205-207: physical
    # lower bound + U(0,1) in normalised coordinates), plain and in the rotated frame
    run_case("3layer_uniform", 3, b3, p20, [0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0],
             "normal", {"std": 0.025}, n_steps=400, seed=21, proposal="uniform")
    run_case("3layer_uniform_rotate", 3, b3, p20, [0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0],
             "normal", {"std": 0.025}, n_steps=60, seed=22, rotate=True, n_burn=10, n_cov_chunk=5,
             n_thin=2, proposal="uniform")
    # the reference's one working configuration (src/main.py:380-407 -> :261-347): two layers,
    # asymmetric-Laplace likelihood, the 60 periods and per-period lambda scaling of site WH01
    # (flipped, clipped at 60: src/main.py:313-317), lambd = 1, kappa = 0.75, bounds of
    # src/main.py:61-64, started at the true model (src/main.py:152-153)
    import pandas as pd
    df = pd.read_csv("/root/reference/data/spread/WH01-asym-laplace-params.csv")
    lam_scale = np.flip(df["scaled_lambd"].values).copy()
    lam_scale[lam_scale > 60] = 60
    b2 = {"depth": np.array([0.001, 0.10]),
          "vel_s": np.array([[0.100, 0.700], [0.100, 0.700], [1.000, 2.000]])}
    wh01_true = [0.010, 0.063, 0.200, 0.400, 1.700]
    run_case("asymlaplace_WH01", 2, b2, np.flip(1 / df["freqs"].values).copy(), wh01_true,
             "asym-laplace", {"lambd_scale": lam_scale, "lambd": 1.0, "kappa": 0.75},
             n_steps=400, seed=6, start=wh01_true)
    if not ONLY:
        misc_fixture()


if __name__ == "__main__":
    main()
