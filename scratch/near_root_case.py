"""One model whose last period's answer the MH kernels and the forward kernel disagreed on."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import mcmc_amd.engine as E
from mcmc_amd.engine import DeviceProblem
import oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from secular_ld import secular_ld
m = np.array([float.fromhex(x) for x in ['0x1.5f04cea1fb866p-10', '0x1.32659f56c2269p-5', '0x1.0d676cad9591cp-4', '0x1.19a16ea03edc6p-3', '0x1.1a5a21d769b5ep-3', '0x1.8f46d93dca59bp+0', '0x1.3550d255a5d1ep-3', '0x1.b4fffc618613fp-1', '0x1.9a5ef93e7824bp+0', '0x1.c4c295cab7da6p+0', '0x1.a4f7eeefa6ee7p-1']])
P = 40
periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
prob = DeviceProblem(5, 1.75, periods)
mt = torch.tensor(m[None], device=prob.device)
for s in (1, 2):
    E.set_default_tuning(search=s)
    v, st = prob.forward(mt)
    print("forward search", s, "%.17g" % v[0, 39].item(), st.tolist())
a, b = 0.19655727085962943, 0.19655696568384817
lt = oracle.layer_table(m, 5, 1.75)
T = periods[39]
for k in range(-4, 5):
    c = 0.5 * (a + b) + k * (a - b) / 2
    dv = prob.secular(mt, torch.zeros(1, dtype=torch.int32, device=prob.device), torch.tensor([39], dtype=torch.int32, device=prob.device), torch.tensor([c], dtype=torch.float64, device=prob.device)).item()
    print("c %.17g device %.6e oracle %.6e long double %.6e" % (c, dv, oracle.secular(*lt, T, c), float(secular_ld(*lt, T, c))))
