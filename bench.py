#!/usr/bin/env python3
"""Headline benchmark: Metropolis-Hastings steps per second, 6-layer Rayleigh
phase-velocity inversion, 1024 chains x 40 periods per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one Model.perturb_params for every chain of the batch (one
proposal, at most one forward curve, one likelihood, one accept test per
chain).  Chains are independent, so ranks share nothing on the data path
(weak scaling: 1024 chains per GPU); the only collective is the timing
barrier / max-reduction.  Inputs are synthetic and seeded: periods
np.flip(1/np.logspace(0.3, 1.3, 40)), vpvs 1.75, Gardner density, bounds
depth [0.001, 0.15] km and vs [0.1, 2.0] km/s, proposal width 0.05, Cauchy
proposal, Gaussian likelihood std 0.025, starts drawn from the prior until the
forward model succeeds, draws from default_rng(0).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_LAYERS, VPVS, N_PERIODS, CHAINS_PER_GPU = 6, 1.75, 40, 1024
TRUE_M = np.array([0.01, 0.02, 0.04, 0.06, 0.08, 0.10, 0.2, 0.3, 0.45, 0.6, 0.8, 1.0, 1.5])


def true_model(n_layers):
    """The 6-layer model of the headline configuration, or one of the same span with n_layers."""
    if n_layers == 6:
        return TRUE_M
    depth = np.linspace(0.01, 0.10, n_layers)
    vs = np.interp(np.linspace(0, 1, n_layers + 1), np.linspace(0, 1, 7), TRUE_M[6:])
    return np.concatenate([depth, vs])
STD = 0.025
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def flops_per_eval(rows, wave="rayleigh"):
    """Algorithmic flops of one secular evaluation (+,-,*,/,sqrt,sin,cos,exp = 1 each)."""
    return 31 + 172 * (rows - 1) if wave == "rayleigh" else 8 + 25 * (rows - 1)


def problem_inputs(n_layers=N_LAYERS):
    periods = np.flip(1 / np.logspace(0.3, 1.3, N_PERIODS)).copy()
    bounds = np.array([[0.001, 0.15, 0.149]] * n_layers + [[0.1, 2.0, 1.9]] * (n_layers + 1))
    return periods, bounds


def prior_starts(prob, bounds, n, rng):
    """Independent prior draws (model.py:94-105) kept when the forward model succeeds."""
    import torch
    out = np.empty((0, bounds.shape[0]))
    while out.shape[0] < n:
        cand = rng.uniform(bounds[:, 0], bounds[:, 1], size=(2 * n, bounds.shape[0]))
        nl = (bounds.shape[0] - 1) // 2
        order = np.argsort(cand[:, :nl], axis=1, kind="stable")
        cand[:, :nl] = np.take_along_axis(cand[:, :nl], order, axis=1)
        cand[:, nl:2 * nl] = np.take_along_axis(cand[:, nl:2 * nl], order, axis=1)
        _, st = prob.forward(torch.tensor(cand, device=prob.device))
        out = np.concatenate([out, cand[st.cpu().numpy() == 0]])
    return out[:n]


def cpu_baseline(periods, bounds, obs, starts, seed, budget_s=12.0):
    """The CPU oracle's MH loop on a bounded sample of the same workload."""
    import oracle
    nl, NP = N_LAYERS, 2 * N_LAYERS + 1
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    noise = {"std": STD}
    rng = np.random.default_rng(seed)
    res = {}
    for label, nt, B in (("1core", 1, 64), ("allcores", cores, min(len(starts), 16 * cores))):
        m = starts[:B]
        st, vel, _ = oracle.forward_batch(m, nl, VPVS, periods, nthreads=nt)
        logL = np.array([oracle.loglik(obs, v, "normal", noise) for v in vel])
        S = oracle.ChainState(m, logL, vel, 0.05, bounds)
        oracle.set_threads(nt)
        n_steps, t0 = 0, time.perf_counter()
        while True:
            idx = rng.integers(0, NP, B).astype(np.int32)
            cz = np.tan(np.pi * (rng.uniform(size=B) - 0.5))
            ua = rng.uniform(size=B)
            oracle.mh_step(S, idx, cz, ua, nl, VPVS, periods, obs, "normal", noise)
            n_steps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s / 2 or n_steps >= 2000:
                break
        res[label] = (B * n_steps / dt, nt, B, n_steps)
        oracle.set_threads(1)
    v, nt, B, n = res["allcores"]
    return {
        "value": v, "unit": "MH steps/s", "cores": nt, "kind": "port",
        "sample": f"{B} chains x {n} steps of the same 6-layer/40-period workload, "
                  f"OpenMP over chains; 1 core: {res['1core'][0]:.1f} steps/s "
                  f"({res['1core'][2]} chains x {res['1core'][3]} steps)",
        "value_1core": res["1core"][0],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--chunk", type=int, default=500,
                    help="MH steps per call of mhd_mh_steps: the reference's cadence of host work "
                         "(n_chunk = n_cov_chunk = 500, src/main.py:127-129)")
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--search", type=int, default=2,
                    help="2 = period-parallel root search (default), 1 = wave-sequential")
    ap.add_argument("--candidates", type=int, default=2,
                    help="2 = two candidates per chain in flight (default), 1 = one")
    # other configurations of the same path (the headline metric is the default of each)
    ap.add_argument("--layers", type=int, default=N_LAYERS)
    ap.add_argument("--itype", default="phase",
                    choices=["phase", "group", "phase+group", "phase+ellipticity", "ellipticity"])
    ap.add_argument("--wave", choices=["rayleigh", "love"], default="rayleigh")
    ap.add_argument("--rotation", choices=["off", "on"], default="off",
                    help="on: rotated-frame proposals and covariance accumulation every step")
    ap.add_argument("--tempering", action="store_true",
                    help="parallel tempering: the chains of every GPU are ladders of --temps "
                         "temperatures; after every --thin steps the ranks all-gather logL "
                         "(RCCL) and run the exchange sweep, inside the timed region")
    ap.add_argument("--temps", type=int, default=64)
    ap.add_argument("--thin", type=int, default=10, help="MH steps between two exchange sweeps")
    ap.add_argument("--ladder", choices=["reference", "geometric"], default="reference",
                    help="reference: 1/1.15**ind from the first tempered rung on (the reference's "
                         "formula); geometric: 1/1.15, 1/1.15**2, ... right after the cold quarter")
    args = ap.parse_args()
    headline = (args.layers, args.itype, args.wave, args.rotation, args.chains, args.tempering) == \
        (N_LAYERS, "phase", "rayleigh", "off", CHAINS_PER_GPU, False)
    if args.tempering:
        args.chunk = args.thin
        if args.chains % args.temps:
            raise SystemExit("--chains must be a multiple of --temps")

    if os.environ.get("MCMC_AMD_BENCH_WATCHDOG"):
        # debugging aid: after that many seconds every thread's stack goes to stderr and the
        # process ends (a hung collective otherwise says nothing)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MCMC_AMD_BENCH_WATCHDOG"]), exit=True)
    import torch
    import torch.distributed as dist
    from mcmc_amd.engine import DeviceChains, DeviceProblem
    from mcmc_amd import _lib as L, engine as E
    from mcmc_amd.inversion.inversion import all_gather_blocks
    E.set_default_tuning(search=args.search)
    E.set_default_tuning(candidates=args.candidates)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # (a rehearsal of the multi-process path on a box with fewer GPUs than ranks: the ranks share
    # the GPUs there are, and MCMC_AMD_DIST_BACKEND=gloo stands in for RCCL, which refuses two
    # ranks on one device; on a node with one GPU per rank neither changes anything)
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        backend = os.environ.get("MCMC_AMD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    nl = args.layers
    periods, bounds = problem_inputs(nl)
    B, NP, K, W = args.chains, 2 * nl + 1, args.steps, args.warmup
    kind = dict(wave=args.wave, itype=args.itype)
    # data: true curve + N(0, std^2); the true curve comes from the product path
    prob0 = DeviceProblem(nl, VPVS, periods, **kind)
    true_curve, st = prob0.forward(torch.tensor(true_model(nl)[None, :], device=prob0.device))
    assert int(st[0]) == 0
    np.random.seed(0)
    obs = true_curve[0].cpu().numpy()
    obs = obs + STD * np.random.randn(N_PERIODS if args.itype == "phase" else len(obs))
    prob = DeviceProblem(nl, VPVS, periods, obs, "normal", {"std": STD}, bounds, **kind)
    rng = np.random.default_rng(1000 + rank)
    starts = prior_starts(prob, bounds, B, rng)
    beta_all = logL_all = swap_counts = None
    if args.tempering:
        # ladders of --temps rungs, rungs of a ladder neighbours: every rank owns whole ladders
        # but gathers logL and sweeps all of them, as a run with any other layout must
        from mcmc_amd.engine import tempering_sweep
        n_ladders = world * B // args.temps
        nf = int(np.ceil(args.temps / 4))
        ladder = np.ones(args.temps)
        ladder[nf:] = 1.0 / 1.15 ** (np.arange(nf, args.temps) - (nf - 1 if args.ladder == "geometric" else 0))
        beta_all = torch.tensor(np.tile(ladder, n_ladders), device=prob.device)
        logL_all = torch.zeros(world * B, dtype=torch.float64, device=prob.device)
        swap_counts = torch.zeros(2, dtype=torch.int64, device=prob.device)
    chains = DeviceChains(prob, starts, 0.05,
                          beta=None if beta_all is None else beta_all[rank * B:(rank + 1) * B])
    assert int((chains.init_likelihood() != 0).sum()) == 0
    rotate = 3 if args.rotation == "on" else 0  # identity rotation matrices: same proposals

    drng = np.random.default_rng(rank)  # default_rng(0) on rank 0
    total = W + K
    dev = prob.device
    idx = torch.tensor(drng.integers(0, NP, (total, B)).astype(np.int32), device=dev)
    cauchy = torch.tensor(np.tan(np.pi * (drng.uniform(size=(total, B)) - 0.5)), device=dev)
    u_acc = torch.tensor(drng.uniform(size=(total, B)), device=dev)

    u_swap = None
    if args.tempering:
        n_sweeps = (total + args.thin - 1) // args.thin
        u_swap = torch.tensor(np.random.default_rng(99).uniform(
            size=(n_sweeps, world * B // args.temps, args.temps - 1)), device=dev)

    def run(lo, hi):
        n_launch = 0
        for a in range(lo, hi, args.chunk):
            b = min(hi, a + args.chunk)
            chains.steps(idx[a:b], cauchy[a:b], u_acc[a:b], rotate=rotate, want_flags=False)
            n_launch += 1
            if args.tempering:
                full = chains.logL
                if world > 1:
                    full = all_gather_blocks(logL_all, chains.logL)  # the sampler's collective
                tempering_sweep(full, beta_all, u_swap[a // args.thin], args.temps,
                                world * B // args.temps, 1, args.temps, swap_counts)
        return n_launch

    run(0, W)
    torch.cuda.synchronize()
    chains.n_evals.zero_()
    acc0 = chains.counters[:, 0].sum().item()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    n_launch = run(W, total)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    lit, ex = (int(x) for x in chains.n_evals.tolist())
    acc_rate = (chains.counters[:, 0].sum().item() - acc0) / (B * K)

    many_build = chains.steps_lib is not L.lib()
    wave_slots = torch.cuda.get_device_properties(dev).multi_processor_count * (12 if many_build else 8)
    if rank == 0:
        F = flops_per_eval(nl + 1, args.wave)
        launch_s = kernel_ms * 1e-3 / n_launch
        achieved = lit * F / n_launch / launch_s / 1e12
        line = {
            "metric": "MH steps/sec (whole node), 6-layer Rayleigh, 1024 chains x 40 periods"
                      if headline else
                      f"MH steps/sec (whole node), parallel tempering: {args.temps} temperatures x "
                      f"{world * B // args.temps} ladders, {B} replicas per GPU, exchange sweep "
                      f"(all-gather of logL + device sweep) every {args.thin} steps"
                      if args.tempering else
                      f"MH steps/sec (whole node), {nl}-layer {args.wave}, datum {args.itype}, "
                      f"{B} chains x 40 periods, rotation {args.rotation}",
            "value": world * B * K / elapsed,
            "unit": "MH steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: 6-layer Rayleigh fundamental-mode phase-velocity "
                            "inversion, 1024 independent chains x 40 periods per GPU"
                            if headline else
                            "configs[2]: parallel tempering, replicas sharded over the GPUs, "
                            "all-gather of logL per sweep" if args.tempering else
                            "variant of configs[1] (see metric)",
                "chains_per_gpu": B, "n_periods": N_PERIODS, "n_layers": nl,
                "wave": args.wave, "itype": args.itype,
                "proposal": "cauchy", "likelihood": "normal", "rotation": args.rotation,
                "steps_per_call": min(args.chunk, K), "calls_timed": n_launch,
                "search": args.search,
                "candidates_in_flight": args.candidates, "acceptance_rate": round(acc_rate, 4),
                "library": "libmcmc_amd_w3.so (three waves per SIMD, for more chains than the "
                           "GPU holds at two)" if many_build else "libmcmc_amd.so",
                **({"temperatures": args.temps, "steps_between_sweeps": args.thin,
                    "beta_ladder": args.ladder,
                    "exchanges_accepted_proposed": swap_counts.tolist()}
                   if args.tempering else {}),
            },
            "roofline": {
                "bound": "valu_fp64",
                "achieved": achieved,
                "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                # HBM bytes come from rocprofv3 --pmc passes, which this process cannot take of
                # itself: see profiles/ for the counters of this same command
                "traffic": None,
                "kernel": "one call of mhd_mh_steps = " + (
                    ("%d x (schedule_kernel, then mh_steps_multi_kernel<4>: workgroups of four "
                     "waves holding one, two or four chains)" if 2 * B <= wave_slots else
                     "%d x (schedule_kernel, then mh_steps_multi_kernel<2>: more chains than the "
                     "GPU holds at once, a queue of one-chain workgroups in the schedule's order)")
                    % (2 if min(args.chunk, K) >= L.lib().mhd_sched_min_steps() else 1)
                    if args.candidates == 2
                    else "mh_steps_multi_kernel<%d>" % args.candidates if args.candidates == 4
                    else "mh_steps_kernel") + "; launch_ms is the call, by HIP events on its stream",
                "launch_ms": launch_s * 1e3,
                "algorithmic_evals_per_launch": lit / n_launch,
                "executed_evals_per_launch": ex / n_launch,
                "flops_per_eval": F,
            },
        }
        if not args.no_cpu and world == 1 and headline:  # the CPU leg belongs to the one-GPU run
            line["cpu_baseline"] = cpu_baseline(periods, bounds, obs, starts, seed=7)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
