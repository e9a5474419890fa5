"""The sequential algorithm's evaluation count of one model whose curve has no root at some
period, by the oracle and by both device searches.  usage: count_case.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, oracle
import mcmc_amd.engine as E
from mcmc_amd.engine import DeviceProblem
m = np.array([float.fromhex(v) for v in ['0x1.40edce5fc6e56p-8', '0x1.b1191cf2c4ff1p-8', '0x1.f8fc5e970f1d3p-6', '0x1.7c6752b8e9f87p+0', '0x1.e001947c78776p-1', '0x1.0c693717137adp+0', '0x1.a19149b61738ap-2']])
periods = np.flip(1 / np.logspace(0.3, 1.3, 40)).copy()
prob = DeviceProblem(3, 1.75, periods)
ost, ovel, one = oracle.forward(m, 3, 1.75, periods)
print("oracle: status", ost, "evaluations", one)
for P in range(1, 41):
    o = oracle.forward(m, 3, 1.75, periods[:P])
    if o[0] != 0:
        print("first period without a root:", P - 1, "evaluations up to it", o[2], "before it", oracle.forward(m, 3, 1.75, periods[:P - 1])[2])
        break
for srch in (1, 2):
    E.set_default_tuning(search=srch)
    ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
    v, st = prob.forward(torch.tensor(m[None], device=prob.device), n_evals=ne)
    print("device forward search", srch, "status", st.tolist(), "evaluations", ne.tolist())
    for P in range(1, 41):
        pr = DeviceProblem(3, 1.75, periods[:P])
        ne = torch.zeros(2, dtype=torch.int64, device=prob.device)
        v, st = pr.forward(torch.tensor(m[None], device=prob.device), n_evals=ne)
        oc = oracle.forward(m, 3, 1.75, periods[:P])
        if int(ne[0]) != oc[2] or int(st[0]) != oc[0]:
            print("   first", P, "periods: device", st.tolist(), int(ne[0]), "oracle", oc[0], oc[2])
