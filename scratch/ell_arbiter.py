"""One configuration of scratch/soak_mh.py on the CPU only: the oracle's chain at the end, and the
ellipticity at one period in extended precision.  usage: ell_arbiter.py [cfg chain period kernels_value]
(cfg 28 chain 58 period 1: the kernels gave 0.343807246923192; cfg 34 chain 56 period 1: 0.60217120138619)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import oracle
sys.path.insert(0, os.path.join(ROOT, "scratch"))
import secular_ld as SL
LD = np.longdouble

r = int(sys.argv[1]) if len(sys.argv) > 1 else 28
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 58
KP = int(sys.argv[3]) if len(sys.argv) > 3 else 1
KV = float(sys.argv[4]) if len(sys.argv) > 4 else 0.343807246923192
rng = np.random.default_rng(7000 + r)
nl = int(rng.integers(2, 9)); NP = 2 * nl + 1
P = int(rng.choice([10, 20, 40]))
itype = int(rng.choice([0, 0, 1, 2, 3]))
wave = int(itype == 0 and rng.uniform() < 0.25)
ND = 2 * P if itype in (2, 3) else P
rotate = int(rng.choice([0, 0, 3])); cand = int(rng.choice([1, 2, 2, 4]))
sched = bool(rng.uniform() < 0.5) and cand == 2
B, S, ncall = 64, 40, 3
periods = np.flip(1 / np.logspace(0.3, 1.3, P)).copy()
bounds = np.array([[0.001, 0.15, 0.149]] * nl + [[0.1, 2.0, 1.9]] * (nl + 1))
true_m = np.concatenate([np.sort(rng.uniform(0.005, 0.14, nl)), np.sort(rng.uniform(0.15, 1.8, nl + 1))])
st0, tc, _ = oracle.forward(true_m, nl, 1.75, periods, wave=wave, itype=itype)
laplace = rng.uniform() < 0.3
assert not laplace and itype == 3
noise = {"std": rng.uniform(0.02, 0.06, ND)}
obs = tc + noise["std"] * rng.standard_normal(ND)
starts = np.tile(true_m, (B, 1)) * rng.uniform(0.9, 1.1, (B, NP))
starts = np.clip(np.stack([oracle.sort_layers(s, nl) for s in starts]), bounds[:, 0], bounds[:, 1])
beta = rng.uniform(0.2, 1.0, B) if rng.uniform() < 0.4 else None
assert rotate == 0
ost, ovel, _ = oracle.forward_batch(starts, nl, 1.75, periods, itype=3, nthreads=8)
ologL = np.array([oracle.loglik(obs, v, "normal", noise) for v in ovel])
st = oracle.ChainState(starts, ologL, ovel, 0.05, bounds)
oracle.set_threads(8)
for c in range(ncall):
    idx = rng.integers(0, NP, (S, B)).astype(np.int32)
    cz = np.tan(np.pi * (rng.uniform(size=(S, B)) - 0.5))
    ua = rng.uniform(size=(S, B))
    for s in range(S):
        oracle.mh_step(st, idx[s], cz[s], ua[s], nl, 1.75, periods, obs, "normal", noise, beta=beta, itype=3)
m = st.m[CH]
print("model", repr(m.tolist()))
print("kernels' value %.15g" % KV)
print("oracle pred phase[k] %.15g ell[k] %.15g" % (st.pred[CH, KP], st.pred[CH, P + KP]))
lt = oracle.layer_table(m, nl, 1.75)
c = st.pred[CH, KP]; T = periods[KP]
e = oracle.compound(*lt, T, c)
print("compound vector (double):", e)
# extended precision: the same recursion, all five components
compound_ld = SL.compound_ld
ev = compound_ld(*lt, T, c)
omega = 2 * np.arctan2(LD(0), LD(-1)) / LD(T); k = omega / LD(c)
print("compound vector (long double):", [float(x) for x in ev])
print("ellipticity long double: -k e2/e1 = %.15g, e3/(k e2) = %.15g" % (float(-k * ev[2] / ev[1]), float(ev[3] / (k * ev[2]))))
kd = 2 * np.pi / T / c
print("ellipticity double (oracle): -k e2/e1 = %.15g, e3/(k e2) = %.15g" % (-kd * e[2] / e[1], e[3] / (kd * e[2])))

# the root itself (extended precision, bisection around the answer) and the ellipticity there
f = lambda x: float(SL.compound_ld(*lt, T, x)[0])
lo, hi = c * (1 - 3e-6), c * (1 + 3e-6)
assert f(lo) * f(hi) < 0
for _ in range(80):
    mid = 0.5 * (lo + hi)
    if f(mid) * f(lo) > 0: lo = mid
    else: hi = mid
root = 0.5 * (lo + hi)
ev = SL.compound_ld(*lt, T, root)
k = omega / LD(root)
print("root %.15g (answer %.15g, %.2e relative away); there e0 = %.2e and the two ratios are %.12g and %.12g" %
      (root, c, c / root - 1, float(ev[0]), float(-k * ev[2] / ev[1]), float(ev[3] / (k * ev[2]))))
for x in (c * (1 - 1e-6), c * (1 - 3e-7), c, c * (1 + 3e-7), c * (1 + 1e-6)):
    ev = SL.compound_ld(*lt, T, x); k = omega / LD(x)
    print("  c %.12f: e0 %+.3e  -k e2/e1 %.6f  e3/(k e2) %.6f" % (x, float(ev[0]), float(-k * ev[2] / ev[1]), float(ev[3] / (k * ev[2]))))
