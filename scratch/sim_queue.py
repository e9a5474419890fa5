import numpy as np, heapq
cost = np.load('/root/repo/gpurun_out/cost.npy') / 100.0   # us per step at one candidate
B, S = len(cost), 250
cons = {1: 1.0, 2: 1.65, 4: 2.35}
SLOTS = 2048
def run(n4, n1, noise, seed):
    rng = np.random.default_rng(seed)
    true = cost * np.exp(noise * rng.standard_normal(B))
    order = np.argsort(-cost)
    items = [(i, 4) for i in order[:n4]] + [(i, 2) for i in order[n4:B - n1]] + [(i, 1) for i in order[B - n1:]]
    free = SLOTS; t = 0.0; running = []; end = 0.0
    for i, J in items:           # FIFO dispatch
        while free < J:
            tf, Jf = heapq.heappop(running); t = max(t, tf); free += Jf
        dur = true[i] * S / cons[J]
        heapq.heappush(running, (t + dur, J)); free -= J
        end = max(end, t + dur)
    return end / 1000
for noise in (0.0, 0.09, 0.13):
    print("noise", noise)
    for n4, n1 in ((30, 106), (30, 0), (64, 0), (128, 0), (192, 0), (256, 0), (384, 0), (128, 64), (256, 128)):
        r = [run(n4, n1, noise, s) for s in range(5)]
        print("  n4 %3d n1 %3d slots asked %4d: makespan %.1f ms" % (n4, n1, 4*n4 + 2*(B-n4-n1) + n1, np.mean(r)))
