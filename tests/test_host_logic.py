"""CPU tests: host logic, ABI surface, and the world_size-2 sharded tempering exchange over
gloo through the sampler's own collective (tests/test_zy_two_ranks.py runs the whole sharded
sampler with two ranks on a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    """libmcmc_amd.so loads without a GPU and exports what include/mcmc_amd.h declares."""
    from mcmc_amd import _lib as L
    assert os.path.exists(L.LIB_PATH), "build first: python -m mcmc_amd.build"
    assert os.path.exists(L.MANY_LIB_PATH), "build first: python -m mcmc_amd.build"
    header = open(os.path.join(ROOT, "include", "mcmc_amd.h")).read()
    declared = set(re.findall(r"\b(mhd_[a-z_]+)\s*\(", header))
    assert declared == set(L.EXPORTS)
    # (the build for many chains per GPU has the same ABI)
    for path in (L.LIB_PATH, L.MANY_LIB_PATH):
        lib = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(lib, name), (path, name)
        lib.mhd_abi_version.restype = ctypes.c_int
        assert lib.mhd_abi_version() == L.ABI_VERSION
    # struct layouts agree with the header's field order and sizes
    assert ctypes.sizeof(L.Problem) == 8 * 4 + 3 * 8 + 6 * 8 + 2 * 4
    assert ctypes.sizeof(L.Chains) == 12 * 8
    assert ctypes.sizeof(L.Diag) == 3 * 8


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before anything touches a device."""
    from mcmc_amd import _lib as L
    lib = L.lib()
    p = L.Problem(n_layers=99, n_periods=10, dc=0.005)
    rc = lib.mhd_forward_batch(ctypes.byref(p), None, 1, None, None, None, None)
    assert rc == -22 and b"n_layers" in lib.mhd_last_error()
    p = L.Problem(n_layers=3, n_periods=4000, dc=0.005)
    assert lib.mhd_forward_batch(ctypes.byref(p), None, 1, None, None, None, None) == -22
    # NULL arrays: an error for a batch, nothing to do for an empty one
    p = L.Problem(n_layers=3, n_periods=8, dc=0.005, dt=0.025, vpvs=1.75, periods=1)
    assert lib.mhd_forward_batch(ctypes.byref(p), None, 1, None, None, None, None) == -22
    assert lib.mhd_forward_batch(ctypes.byref(p), None, 0, None, None, None, None) == 0
    assert lib.mhd_forward_batch(ctypes.byref(p), None, -1, None, None, None, None) == -22
    with pytest.raises(L.MhdError):
        L.check(-22)


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mcmc_amd.engine import DeviceProblem
    from mcmc_amd._lib import MhdError
    with pytest.raises(MhdError):
        DeviceProblem(3, 1.75, np.linspace(0.05, 0.5, 20))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mcmc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "surf96_oracle" not in txt, f


def test_parameterisation_helpers_match_reference():
    from mcmc_amd.inversion.model_params import DispersionCurveParams
    g = load_golden("mh_misc.json")
    mp = DispersionCurveParams(
        2, {"depth": np.array([0.001, 0.10]),
            "vel_s": np.array([[0.1, 0.7], [0.1, 0.7], [1.0, 2.0]])},
        {"depth": 0.05, "vel_s": 0.05}, 1.75)
    np.testing.assert_array_equal(mp.assemble_param_bounds(), g["assemble_param_bounds_2layer"])
    np.testing.assert_array_equal(mp.assemble_proposal_width(),
                                  g["assemble_proposal_width_2layer"])
    s = g["sort_layers"]
    np.testing.assert_array_equal(mp.sort_layers(np.array(s["in"])), s["out"])
    v = g["vel_p"]
    np.testing.assert_array_equal(mp.get_vel_p(np.array(v["vs"])), v["vp"])
    np.testing.assert_array_equal(mp.get_density(np.array(v["vp"])), v["rho"])
    assert mp.n_model_params == 5 and mp.depth_inds.sum() == 2 and mp.vel_s_inds.sum() == 3


def test_get_betas_reference_and_fixed():
    from mcmc_amd.inversion.inversion import Inversion
    g = load_golden("mh_misc.json")["get_betas_factor_1.15"]
    for n, want in g.items():
        inv = Inversion.__new__(Inversion)
        inv.n_chains = int(n)
        inv.reference_betas = True
        np.testing.assert_allclose(inv.get_betas(1.15), want, rtol=0, atol=0)
        inv.reference_betas = False
        fixed = inv.get_betas(1.15)
        want = np.array(want)
        np.testing.assert_array_equal(fixed[want > 0], want[want > 0])
        assert (fixed[want == 0] == 1).all()


def test_geometric_ladder_option():
    """beta_ladder="geometric": the cold quarter at 1, then 1/f, 1/f^2, ... without the
    reference's jump; the default stays the reference's formula."""
    from mcmc_amd.inversion.inversion import Inversion
    inv = Inversion.__new__(Inversion)
    inv.n_chains, inv.reference_betas, inv.beta_ladder = 8, False, "geometric"
    b = inv.get_betas(1.15)
    np.testing.assert_allclose(b, [1, 1] + [1.15 ** -k for k in range(1, 7)], rtol=1e-15)
    inv.beta_ladder = "reference"
    np.testing.assert_allclose(inv.get_betas(1.15), [1, 1] + [1.15 ** -k for k in range(2, 8)],
                               rtol=1e-15)


def test_stepsize_tuning_argument_validation_without_gpu():
    from mcmc_amd import _lib as L
    lib = L.lib()
    p = L.Problem(n_layers=3, n_periods=8, dc=0.005, dt=0.025, vpvs=1.75, periods=1)
    assert lib.mhd_stepsize_tuning(ctypes.byref(p), None, 4, None) == -22
    ch = L.Chains()
    assert lib.mhd_stepsize_tuning(ctypes.byref(p), ctypes.byref(ch), 4, None) == -22
    ch = L.Chains(width=8, counters=8)
    assert lib.mhd_stepsize_tuning(ctypes.byref(p), ctypes.byref(ch), 0, None) == 0
    assert lib.mhd_stepsize_tuning(ctypes.byref(p), ctypes.byref(ch), -1, None) == -22
    assert lib.mhd_sched_min_steps() == 48
    assert lib.mhd_static_reset(None) == -22     # mhd_init() has not been called: no device


def test_tempering_sweep_rule(oracle):
    betas = np.array([1.0, 1.0, 0.5, 0.25])
    logL = np.array([-10.0, -12.0, -3.0, -50.0])
    # pair (0,1): equal betas -> skipped.  pair (1,2): logratio=(0.5-1)(-12+3)=4.5 -> accept.
    # pair (2,3) now holds betas (1.0, 0.25): logratio=(0.25-1)(-3+50) < 0, xi=0.9 -> reject.
    b, acc, prop = oracle.tempering_sweep(betas, logL, np.array([0.3, 0.99, 0.9]))
    np.testing.assert_array_equal(b, [1.0, 0.5, 1.0, 0.25])
    assert (acc, prop) == (1, 2)
    # the same ladder twice, once per layout: exchanges stay inside a ladder
    from mcmc_amd.inversion.inversion import ladder_layout
    xi = np.array([[0.3, 0.99, 0.9], [0.3, 0.99, 0.9]])
    for layout in ("ladder", "rung"):
        rs, ls = ladder_layout(4, 2, layout)
        pos = (np.arange(2)[:, None] * ls + np.arange(4)[None, :] * rs).ravel()
        bb, ll = np.empty(8), np.empty(8)
        bb[pos], ll[pos] = np.tile(betas, 2), np.tile(logL, 2)
        out, acc, prop = oracle.tempering_sweep(bb, ll, xi, 4, 2, rs, ls)
        np.testing.assert_array_equal(out[pos], np.tile([1.0, 0.5, 1.0, 0.25], 2))
        assert (acc, prop) == (2, 4)


def test_replica_betas_per_ladder():
    """64 temperatures x 128 ladders: every ladder holds the ladder of get_betas(64), the
    coldest quarter at beta = 1, nothing underflows; 1024 replicas per rank of 8."""
    from mcmc_amd.inversion.inversion import Inversion, ladder_layout
    for layout in ("ladder", "rung"):
        inv = Inversion.__new__(Inversion)
        inv.n_chains, inv.reference_betas = 128, False
        inv.n_temps, inv.n_ladders, inv.n_replicas = 64, 128, 64 * 128
        inv.rung_stride, inv.ladder_stride = ladder_layout(64, 128, layout)
        b = inv.replica_betas(1.15)
        ladder = inv.get_betas(1.15, 64)
        assert ladder.min() > 1e-4 and (ladder[:16] == 1).all() and (np.diff(ladder[15:]) < 0).all()
        grid = b[np.arange(128)[:, None] * inv.ladder_stride + np.arange(64)[None, :] * inv.rung_stride]
        assert (grid == ladder[None, :]).all()
        assert b.size == 8192 and b.size // 8 == 1024


def _rank_main(rank, world, port, ret, tmp):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from mcmc_amd.inversion.inversion import Inversion, all_gather_blocks, ladder_layout
    n_temps, n_ladders = 4, 6
    n = n_temps * n_ladders
    rng = np.random.RandomState(3)           # same stream on every rank
    all_logL = rng.normal(-100, 20, (5, n))  # 5 sweeps
    out = {}
    for layout in ("ladder", "rung"):
        rs, ls = ladder_layout(n_temps, n_ladders, layout)
        inv = Inversion.__new__(Inversion)
        inv.n_chains, inv.reference_betas = n_ladders, False
        inv.n_temps, inv.n_ladders, inv.n_replicas = n_temps, n_ladders, n
        inv.rung_stride, inv.ladder_stride = rs, ls
        betas = inv.replica_betas(1.3)
        swap_rng = np.random.RandomState(0)
        per = n // world
        full = torch.zeros(n, dtype=torch.float64)
        for sweep in range(5):
            local = torch.tensor(all_logL[sweep, rank * per:(rank + 1) * per])
            # the sampler's own collective (Inversion.gather_logL calls it on device tensors)
            all_gather_blocks(full, local)
            betas, _, _ = oracle.tempering_sweep(betas, full.numpy(),
                                                 swap_rng.rand(n_ladders, n_temps - 1),
                                                 n_temps, n_ladders, rs, ls)
        out[layout] = betas.tolist()
    ret[rank] = out
    # every rank writes the records of its own chains: no two ranks name the same file
    inv = Inversion.__new__(Inversion)
    inv.rank, inv.world, inv.out_dir, inv.out_filename = rank, world, tmp + "/", "run"
    inv._lo, inv.n_local, inv._writer = 2 * rank, 2, None
    inv._write_chunk(0, {"model_params": [torch.full((2, 3), float(rank))]})
    # a missing name is made up from rank 0's clock, the same on every rank
    inv.out_filename = None
    ret[f"name{rank}"] = inv._clock_name()
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_tempering_two_ranks_gloo(oracle, tmp_path):
    """world_size 2 over gloo, 4 temperatures x 6 ladders in both layouts: both ranks reach the
    betas of the single-process sweeps; and both write their chunk of results."""
    import torch.multiprocessing as mp
    from mcmc_amd.inversion.inversion import Inversion, ladder_layout
    world, port = 2, 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_main, args=(world, port, ret, str(tmp_path)), nprocs=world, join=True)
    n_temps, n_ladders = 4, 6
    n = n_temps * n_ladders
    rng = np.random.RandomState(3)
    all_logL = rng.normal(-100, 20, (5, n))
    for layout in ("ladder", "rung"):
        rs, ls = ladder_layout(n_temps, n_ladders, layout)
        inv = Inversion.__new__(Inversion)
        inv.n_chains, inv.reference_betas = n_ladders, False
        inv.n_temps, inv.n_ladders, inv.n_replicas = n_temps, n_ladders, n
        inv.rung_stride, inv.ladder_stride = rs, ls
        start = inv.replica_betas(1.3)
        betas = start.copy()
        swap_rng = np.random.RandomState(0)
        for sweep in range(5):
            betas, _, _ = oracle.tempering_sweep(betas, all_logL[sweep],
                                                 swap_rng.rand(n_ladders, n_temps - 1), n_temps,
                                                 n_ladders, rs, ls)
        assert ret[0][layout] == ret[1][layout] == betas.tolist()
        assert (betas != start).any()
        pos = np.arange(n_ladders)[:, None] * ls + np.arange(n_temps)[None, :] * rs
        np.testing.assert_array_equal(np.sort(betas[pos], axis=1), np.sort(start[pos], axis=1))
    for r in range(world):
        f = np.load(str(tmp_path) + f"/results-run-r{r}-00000.npz")
        assert (f["model_params"] == float(r)).all() and f["model_params"].shape == (2, 3)
        assert f["chain"].tolist() == [2 * r, 2 * r + 1] and int(f["rank"]) == r
    assert ret["name0"] == ret["name1"] and ret["name0"].isdigit()


def test_rates_follow_the_reference_rule():
    """acc_rate and the error ratios of a record: counts over n_acc + n_rej, 0 while nothing was
    rejected (src/inversion/model.py:483, :520)."""
    from mcmc_amd.inversion.inversion import rates_from_counters
    cnt = np.array([[3, 0, 1, 0, 0], [3, 1, 1, 0, 2], [0, 4, 4, 0, 0]])
    r = rates_from_counters(cnt)
    np.testing.assert_array_equal(r["acc_rate"], [0.0, 0.75, 0.0])
    np.testing.assert_array_equal(r["bounds_err"], [0.0, 0.25, 1.0])
    np.testing.assert_array_equal(r["fm_err"], [0.0, 0.5, 0.0])
    np.testing.assert_array_equal(r["physics_err"], [0.0, 0.0, 0.0])


def test_stepsize_tuning_of_the_compatibility_model_matches_reference():
    """Model.stepsize_tuning on the host (one chain) against the reference's trace: the widths
    after each tuning, given the counters the trace holds at that step."""
    from mcmc_amd.inversion.model import Model
    for case in ("3layer_tune", "3layer_tune_individual"):
        g = load_golden(f"mh_trace_{case}.json")
        m = Model.__new__(Model)
        m.proposal_width = np.array(g["width0"])
        every = g["tune_every"]
        for s in range(every - 1, len(g["steps"]), every):
            rec = g["steps"][s]
            m.acceptance_rate = {"acc_rate": np.asarray(rec["acc_rate"])}
            m.stepsize_tuning(s)
            np.testing.assert_array_equal(m.proposal_width, rec["width"])


def test_elementary_functions_of_the_kernels_are_accurate(tmp_path):
    """mcmc_amd/csrc/mhd_math.hpp (sincos, exp, reciprocal square root of the secular-function
    kernels) compiled for the host against long-double libm over two million arguments: exp and
    the reciprocal square root below one ulp, sin and cos (arguments up to 1e5) below 1.6 ulp
    where they are not near a zero and to 2e-16 absolute everywhere."""
    import subprocess
    exe = str(tmp_path / "math_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "host", "math_check.cpp")])
    out = subprocess.check_output([exe], text=True).split()
    got = {out[i]: float(out[i + 1]) for i in range(0, len(out), 2)}
    assert set(got) >= {"sin_ulp", "cos_ulp", "sincos_abs", "exp_ulp", "rsqrt_ulp"}, got
    for k, lim in (("sin_ulp", 1.6), ("cos_ulp", 1.6), ("exp_ulp", 1.0), ("rsqrt_ulp", 1.0)):
        assert got[k] < lim, got
    assert got["sincos_abs"] < 2.5e-16, got
