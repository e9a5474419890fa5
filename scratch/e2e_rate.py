"""End-to-end rate of Inversion.random_walk (host draws from np.random, upload, launches, history
download) next to the kernel headline: 1024 chains, 6 layers, 40 periods, n_thin = 10,
n_chunk = 50 outer steps (500 MH steps per call)."""
import os, sys, time
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from mcmc_amd.inversion import Data, DispersionCurveParams, Inversion
from bench import TRUE_M, problem_inputs, STD
nl, P = 6, 40
periods, _ = problem_inputs(nl)
mp = DispersionCurveParams(nl, {"depth": np.array([0.001, 0.15]), "vel_s": np.array([0.1, 2.0])},
                           {"depth": 0.05, "vel_s": 0.05}, 1.75)
np.random.seed(0)
obs = mp.forward_model(periods, TRUE_M) + STD * np.random.randn(P)
data = Data(periods, obs, "normal", {"std": STD})
import tempfile
res = {}
for label, files, asink in (("no result files", False, True),
                            ("result files (.npz per chunk), written beside the sampling", True, True),
                            ("result files (.npz per chunk), synchronous sink", True, False)):
    for n_mcmc in (200, 600):
        np.random.seed(1)
        with tempfile.TemporaryDirectory() as tmp:
            inv = Inversion(data, mp, "normal", {"std": STD}, n_burn=0, n_chunk=50, n_mcmc=n_mcmc, n_thin=10,
                            n_chains=1024, n_temps=1, out_filename="e2e", write_results=bool(files), out_dir=tmp + "/", async_sink=asink)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            inv.random_walk(mp, "cauchy")
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[(label, n_mcmc)] = dt
        print("%s: 1024 chains x %d outer steps x n_thin 10 = %d MH steps in %.3f s -> %.0f MH steps/s end to end" % (label, n_mcmc, 1024 * n_mcmc * 10, dt, 1024 * n_mcmc * 10 / dt))
    d = res[(label, 600)] - res[(label, 200)]
    print("%s: outer steps 200..600 (past the start from prior draws): %.0f MH steps/s" % (label, 1024 * 400 * 10 / d))
