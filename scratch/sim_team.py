import numpy as np
cost = np.load('/root/repo/gpurun_out/cost.npy') / 100.0   # us per step at one candidate
B, S = len(cost), 250
cons = {0: 0.0, 1: 1.0, 2: 1.65, 3: 2.07, 4: 2.35, 5: 2.55, 6: 2.70, 7: 2.82, 8: 2.90}
rng = np.random.default_rng(0)
def sim(noise):
    # the true cost of the next launch differs from the estimate
    true = cost * np.exp(noise * rng.standard_normal(B))
    order = np.argsort(-cost)
    # now: static classes as the current schedule would choose (approximation: 1,2,4 water-filling)
    def need(T):
        J = np.where(cost <= T, 1, np.where(cost / 1.65 <= T, 2, np.where(cost / 2.35 <= T, 4, 99)))
        return J
    T = next(T for T in np.arange(150, 900, 2.0) if need(T).sum() <= 1990)
    J = need(T)
    static = (true * S / np.array([cons[j] for j in J])).max()
    # teams: workgroup g holds one chain of each quartile, eight waves, two per chain at first
    Q = B // 4
    worst = 0.0
    for g in range(Q):
        ids = [order[g], order[Q + g], order[2 * Q + g], order[3 * Q + g]]
        rem = np.array([float(S)] * 4); Jt = np.array([2, 2, 2, 2]); c = true[ids]; t = 0.0
        while (rem > 0).any():
            rate = np.array([cons[int(j)] for j in Jt]) / c
            live = rem > 0
            dt = (rem[live] / rate[live]).min()
            t += dt
            rem = np.where(live, rem - rate * dt, 0.0)
            rem[rem < 1e-9] = 0.0
            # waves of finished chains move to the chain with the most time left
            free = int(Jt[rem == 0].sum()); Jt[rem == 0] = 0
            for _ in range(free):
                live = rem > 0
                if not live.any(): break
                left = np.where(live & (Jt < 8), rem / (np.array([cons[int(j)] for j in Jt]) / c + 1e-12), -1)
                k = int(np.argmax(left))
                if left[k] < 0: break
                Jt[k] += 1
        worst = max(worst, t)
    mean_work = (true * S / 1.65).mean()
    return static / 1000, worst / 1000, mean_work / 1000
for noise in (0.0, 0.08, 0.15):
    s_, t_, m_ = sim(noise)
    print("noise %.2f: static classes makespan %.1f ms, teams %.1f ms (mean chain at two candidates %.1f ms)" % (noise, s_, t_, m_))
