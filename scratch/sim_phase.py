import numpy as np
cost = np.load('/root/repo/gpurun_out/cost.npy') / 100.0   # us per step at one candidate
B, S = len(cost), 250
cons = {1: 1.0, 2: 1.65, 4: 2.35, 8: 2.9}
def classes(c, rem, budget=1990):
    # smallest T with sum J <= budget, J from {1,2,4,8}; chains with nothing left take no slot
    w = c * rem
    for T in np.arange(1.0, 400.0, 0.5) * 1000:
        J = np.where(rem <= 0, 0, np.where(w <= T, 1, np.where(w / 1.65 <= T, 2, np.where(w / 2.35 <= T, 4, np.where(w / 2.9 <= T, 8, 99)))))
        if J.max() < 99 and J.sum() <= budget: return J
    raise SystemExit("no fit")
def run(noise, deadlines, seed, overhead=0.05):
    rng = np.random.default_rng(seed)
    true = cost * np.exp(noise * rng.standard_normal(B))
    rem = np.full(B, float(S)); t = 0.0
    for D in list(deadlines) + [None]:
        J = classes(cost, rem)
        rate = np.array([cons.get(int(j), 0.0) for j in J]) / true     # steps per us
        need = np.where(rem > 0, rem / np.maximum(rate, 1e-12), 0.0)
        if D is None or need.max() <= D * 1000:
            t += need.max() / 1000 + overhead; rem[:] = 0; break
        done = np.minimum(rem, np.floor(rate * D * 1000))
        rem = rem - done; t += D + overhead
    return t
for noise in (0.09, 0.13):
    base = np.mean([run(noise, (), s) for s in range(5)])
    print("noise %.2f: one phase %.1f ms" % (noise, base))
    for dl in ((60,), (70,), (55, 20), (50, 20, 10), (65, 15), (45, 25, 12)):
        print("   deadlines", dl, "-> %.1f ms" % np.mean([run(noise, dl, s) for s in range(5)]))
