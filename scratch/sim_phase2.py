import numpy as np
cost = np.load('/root/repo/gpurun_out/cost.npy') / 100.0   # us per step at one candidate
B, S, SLOTS = len(cost), 250, 2048
cons = {0: 0.0, 1: 1.0, 2: 1.65, 4: 2.35}
def classes0(c):
    order = np.argsort(-c); J = np.full(B, 2)
    J[order[:30]] = 4; J[order[B - 106:]] = 1      # as the current schedule does
    return J
def classes_later(c, rem):
    w = c * rem; R = int((rem > 0).sum()); order = np.argsort(-w)
    n4 = min(R, max(0, (SLOTS - 2 * R) // 2))
    J = np.where(rem > 0, 2, 0); J[order[:n4]] = 4
    return J
def run(noise, alphas, seed, overhead=0.1):
    rng = np.random.default_rng(seed)
    true = cost * np.exp(noise * rng.standard_normal(B))
    rem = np.full(B, float(S)); t = 0.0
    for ph in range(len(alphas) + 1):
        J = classes0(cost) if ph == 0 else classes_later(cost, rem)
        cj = np.array([cons[int(j)] for j in J])
        pred = np.where(rem > 0, cost * rem / np.maximum(cj, 1e-9), 0.0)
        rate = cj / true
        need = np.where(rem > 0, rem / np.maximum(rate, 1e-12), 0.0)
        if ph == len(alphas):
            t += need.max() / 1000 + overhead; break
        D = alphas[ph] * (np.median(pred) if ph == 0 else pred[rem > 0].mean())
        if need.max() <= D:
            t += need.max() / 1000 + overhead; rem[:] = 0; break
        done = np.minimum(rem, np.ceil(rate * D))
        rem = rem - done; t += D / 1000 + 0.3 + overhead     # 0.3 ms: the round in flight at the deadline
        if not (rem > 0).any(): break
    return t
for noise in (0.09, 0.13):
    print("noise %.2f: one phase %.1f ms" % (noise, np.mean([run(noise, (), s) for s in range(5)])))
    for al in ((0.9,), (1.0,), (0.9, 1.0), (1.0, 1.0), (0.8, 1.0), (0.9, 1.5), (0.9, 1.0, 1.0), (1.0, 1.5, 1.5)):
        print("   alphas", al, "-> %.1f ms" % np.mean([run(noise, al, s) for s in range(5)]))
