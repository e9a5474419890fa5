"""mhd_forward_batch by itself: curves per second of the headline problem's forward model (6 layers,
40 periods, prior draws, no chain and so no guide for the predictor), both searches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as Bn
from mcmc_amd.engine import DeviceProblem

nl = Bn.N_LAYERS
periods, bounds = Bn.problem_inputs(nl)
rng = np.random.default_rng(5)
for B in (1024, 8192, 65536):
    for search in (2, 1):
        prob = DeviceProblem(nl, Bn.VPVS, periods, search=search)
        cand = rng.uniform(bounds[:, 0], bounds[:, 1], size=(B, 2 * nl + 1))
        order = np.argsort(cand[:, :nl], axis=1, kind="stable")
        cand[:, :nl] = np.take_along_axis(cand[:, :nl], order, axis=1)
        cand[:, nl:2 * nl] = np.take_along_axis(cand[:, nl:2 * nl], order, axis=1)
        m = torch.tensor(cand, device=prob.device)
        prob.forward(m)
        torch.cuda.synchronize()
        prob.n_evals.zero_() if hasattr(prob, "n_evals") else None
        t0 = time.perf_counter()
        for _ in range(5):
            vel, st = prob.forward(m)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        ok = int((st == 0).sum())
        print("B %6d search %d: %.3f ms per batch, %.2f M curves/s (%d with a root at every period)" % (B, search, dt * 1e3, B / dt / 1e6, ok))
