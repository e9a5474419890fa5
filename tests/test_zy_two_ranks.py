"""The sharded tempered sampler with world size 2: two ranks, each a process of its own, run the
product's Inversion.random_walk -- all_gather_blocks (dist.all_gather_into_tensor on the device
tensors) + mhd_tempering_sweep after every outer step -- on their halves of 4 temperatures x 6
ladders.  Both share the one GPU of the test box; gloo stands in for RCCL, which refuses two
ranks on one device.  Betas, states, swap counters and the per-rank result files must be those
of a single-process host simulation (oracle.mh_step + oracle.tempering_sweep, same draws)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "helpers"))


@pytest.mark.parametrize("layout", ["ladder", "rung"])
def test_two_ranks_run_the_products_all_gather_and_sweep(oracle, tmp_path, layout):
    import two_rank_worker as W
    from mcmc_amd.inversion.inversion import ladder_layout
    NT, NLAD, nl = W.N_TEMPS, W.N_LADDERS, W.NL
    R = NT * NLAD
    mp, periods = W.problem()
    rng = np.random.default_rng(2)
    obs = oracle.forward(W.TRUE_M, nl, 1.75, periods)[1] + 0.025 * rng.standard_normal(W.P)
    np.save(tmp_path / "obs.npy", obs)
    world, port = 2, str(29600 + os.getpid() % 1500)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "helpers", "two_rank_worker.py"),
                               str(r), str(world), port, str(tmp_path), layout],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=100)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{outs[r][-3000:]}"

    # the same run on the host, in one process
    noise = {"std": 0.025}
    rs, ls = ladder_layout(NT, NLAD, layout)
    ladder = np.ones(NT)
    nf = int(np.ceil(NT / 4))
    ladder[nf:] = 1.0 / 1.3 ** np.arange(nf, NT)
    betas = np.empty(R)
    for l in range(NLAD):
        betas[l * ls + np.arange(NT) * rs] = ladder
    start_betas = betas.copy()
    starts = np.tile(W.TRUE_M, (R, 1))
    _, vel, _ = oracle.forward_batch(starts, nl, 1.75, periods)
    logL0 = np.array([oracle.loglik(obs, v, "normal", noise) for v in vel])
    st = oracle.ChainState(starts, logL0, vel, 0.05, mp.assemble_param_bounds())
    draw_rng, swap_rng = np.random.RandomState(77), np.random.RandomState(4)
    acc = prop = 0
    hist_m, hist_b = [], []
    for chunk in range(W.N_MCMC // W.N_CHUNK):
        idx, cz, xi = W.full_draws(draw_rng, W.N_CHUNK * W.N_THIN, R)
        u_swap = swap_rng.rand(W.N_CHUNK, NLAD, NT - 1)
        for j in range(W.N_CHUNK):
            for s in range(j * W.N_THIN, (j + 1) * W.N_THIN):
                oracle.mh_step(st, idx[s], cz[s], xi[s], nl, 1.75, periods, obs, "normal", noise,
                               beta=betas)
            betas, a, p = oracle.tempering_sweep(betas, st.logL, u_swap[j], NT, NLAD, rs, ls)
            acc, prop = acc + a, prop + p
            hist_m.append(st.m.copy())
            hist_b.append(betas.copy())
    assert (betas != start_betas).any() and acc > 0
    per = R // world
    for r in range(world):
        f = np.load(tmp_path / f"rank{r}.npz")
        assert int(f["lo"]) == r * per
        np.testing.assert_array_equal(f["betas"], betas)            # all replicas, every rank
        np.testing.assert_array_equal(f["m"], st.m[r * per:(r + 1) * per])
        np.testing.assert_allclose(f["logL"], st.logL[r * per:(r + 1) * per], rtol=1e-9)
        # what the last all-gather left on this rank: every replica's logL before the last sweep
        np.testing.assert_allclose(f["logL_all"], st.logL, rtol=1e-9)
        assert f["swaps"].tolist() == [acc, prop]
        got_m = np.concatenate([np.load(tmp_path / f"results-pt2-r{r}-{c:05d}.npz")["model_params"]
                                for c in range(W.N_MCMC // W.N_CHUNK)])
        got_b = np.concatenate([np.load(tmp_path / f"results-pt2-r{r}-{c:05d}.npz")["beta"]
                                for c in range(W.N_MCMC // W.N_CHUNK)])
        np.testing.assert_array_equal(got_m, np.array(hist_m)[:, r * per:(r + 1) * per])
        np.testing.assert_array_equal(got_b, np.array(hist_b)[:, r * per:(r + 1) * per])
