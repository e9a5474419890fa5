"""CPU tests: host logic, ABI surface, and the world_size-2 sharded tempering exchange (gloo)."""
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
    lib = ctypes.CDLL(L.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "mcmc_amd.h")).read()
    declared = set(re.findall(r"\b(mhd_[a-z_]+)\s*\(", header))
    assert declared == set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    lib.mhd_abi_version.restype = ctypes.c_int
    assert lib.mhd_abi_version() == L.ABI_VERSION
    # struct layouts agree with the header's field order and sizes
    assert ctypes.sizeof(L.Problem) == 8 * 4 + 3 * 8 + 5 * 8 + 2 * 4
    assert ctypes.sizeof(L.Chains) == 12 * 8


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


def test_tempering_sweep_rule():
    from mcmc_amd.inversion.inversion import tempering_sweep
    betas = np.array([1.0, 1.0, 0.5, 0.25])
    logL = np.array([-10.0, -12.0, -3.0, -50.0])
    # pair (0,1): equal betas -> skipped.  pair (1,2): logratio=(0.5-1)(-12+3)=4.5 -> accept.
    # pair (2,3) now holds betas (1.0, 0.25): logratio=(0.25-1)(-3+50) < 0, xi=0.9 -> reject.
    b, acc, prop = tempering_sweep(betas, logL, np.array([0.3, 0.99, 0.9]))
    np.testing.assert_array_equal(b, [1.0, 0.5, 1.0, 0.25])
    assert (acc, prop) == (1, 2)


def _rank_main(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcmc_amd.inversion.inversion import gather_chain_values, tempering_sweep
    n = 16
    rng = np.random.RandomState(3)           # same stream on every rank
    all_logL = rng.normal(-100, 20, (5, n))  # 5 sweeps
    betas = 1.0 / 1.3 ** np.arange(n)
    swap_rng = np.random.RandomState(0)
    per = n // world
    for sweep in range(5):
        local = torch.tensor(all_logL[sweep, rank * per:(rank + 1) * per])
        full = gather_chain_values(local, n, world)
        betas, _, _ = tempering_sweep(betas, full, swap_rng.rand(n - 1))
    ret[rank] = betas.tolist()
    dist.destroy_process_group()


def test_sharded_tempering_two_ranks_gloo():
    """world_size 2 over gloo: both ranks reach the betas of the single-process sweep."""
    import torch.multiprocessing as mp
    from mcmc_amd.inversion.inversion import tempering_sweep
    world, port = 2, 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_main, args=(world, port, ret), nprocs=world, join=True)
    n = 16
    rng = np.random.RandomState(3)
    all_logL = rng.normal(-100, 20, (5, n))
    betas = 1.0 / 1.3 ** np.arange(n)
    swap_rng = np.random.RandomState(0)
    for sweep in range(5):
        betas, _, _ = tempering_sweep(betas, all_logL[sweep], swap_rng.rand(n - 1))
    assert ret[0] == ret[1] == betas.tolist()
    assert sorted(ret[0]) == sorted((1.0 / 1.3 ** np.arange(n)).tolist())  # a permutation
