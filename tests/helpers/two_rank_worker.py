"""One rank of the two-rank tempering test (tests/test_zy_two_ranks.py), started as a process
of its own before anything touches the GPU:

    python two_rank_worker.py RANK WORLD PORT OUT_DIR LAYOUT

Builds a real Inversion(n_temps=4, n_chains=6) on this rank's block of the 24 replicas (both
ranks share GPU 0; gloo stands in for RCCL, which refuses two ranks on one device), runs
random_walk -- the product's own all-gather of logL and device sweep after every outer step --
on draws that every rank generates alike and takes its columns of, and leaves its betas, states
and swap counters in OUT_DIR/rank<r>.npz (the result chunks go to OUT_DIR as well)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_TEMPS, N_LADDERS, NL, P = 4, 6, 3, 20
N_MCMC, N_THIN, N_CHUNK = 10, 3, 5
TRUE_M = np.array([0.02, 0.04, 0.07, 0.2, 0.4, 0.6, 1.0])


def problem():
    from mcmc_amd.inversion import DispersionCurveParams
    bounds = {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])}
    mp = DispersionCurveParams(NL, bounds, {"depth": 0.05, "vel_s": 0.05}, 1.75)
    periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
    return mp, periods


def full_draws(rng, S, R):
    """The draws of S MH steps for all R replicas (every rank makes the same arrays)."""
    NP = 2 * NL + 1
    idx = rng.randint(NP, size=(S, R)).astype(np.int32)
    cz = np.tan(np.pi * (rng.uniform(size=(S, R)) - 0.5))
    xi = rng.uniform(size=(S, R))
    return idx, cz, xi


def main():
    rank, world, port, out_dir, layout = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3],
                                          sys.argv[4], sys.argv[5])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from mcmc_amd.inversion import Data, Inversion
    mp, periods = problem()
    obs = np.load(os.path.join(out_dir, "obs.npy"))
    noise = {"std": 0.025}
    data = Data(periods, obs, "normal", noise)
    data.model_true = TRUE_M
    inv = Inversion(data, mp, "normal", noise, n_burn=0, n_chunk=N_CHUNK, n_mcmc=N_MCMC,
                    n_thin=N_THIN, n_chains=N_LADDERS, n_temps=N_TEMPS, beta_spacing_factor=1.3,
                    set_starting_model=True, out_filename="pt2", out_dir=out_dir + "/",
                    swap_seed=4, layout=layout)
    assert inv.world == world and inv.n_local == N_TEMPS * N_LADDERS // world
    rng = np.random.RandomState(77)
    lo, n = inv._lo, inv.n_local

    def my_draws(S):
        idx, cz, xi = full_draws(rng, S, inv.n_replicas)
        return (np.ascontiguousarray(idx[:, lo:lo + n]), np.ascontiguousarray(cz[:, lo:lo + n]),
                np.ascontiguousarray(xi[:, lo:lo + n]))
    inv._draws = my_draws
    inv.random_walk(mp, "cauchy")
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), betas=inv.betas,
             m=inv.state.m.cpu().numpy(), logL=inv.state.logL.cpu().numpy(),
             logL_all=inv.logL_all.cpu().numpy(), swaps=np.array([inv.swap_acc, inv.swap_prop]),
             lo=lo)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
