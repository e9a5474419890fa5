import os, sys, ctypes, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch; torch.cuda.init(); torch.zeros(1,device="cuda")
import mcmc_amd._lib as L
L.LIB_PATH=os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
lib=L.lib()
sys.argv=['bench.py','--steps','500','--warmup','250','--no-cpu']+sys.argv[1:]
import runpy, io, contextlib
import mcmc_amd.engine as E
_orig=E.DeviceChains.__init__
_keep=[]
def _init(self,*a,**k):
    _orig(self,*a,**k); _keep.append(self)
E.DeviceChains.__init__=_init
import numpy as np
buf=io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT,'bench.py'), run_name='__main__')
d=json.loads(buf.getvalue())
print("value %.0f ms/step %.3f"%(d['value'], d['ms_per_step']))
tt=(ctypes.c_ulonglong*4096)()
lib.mhd_debug_times(tt)
a=np.array(list(tt),dtype=np.float64).reshape(2048,2)[:1024]
t0=a[:,0].min()
st=(a[:,0]-t0)/100.0; en=(a[:,1]-t0)/100.0
dur=en-st
print("start us: p50 %.0f p90 %.0f p99 %.0f max %.0f | end us: p50 %.0f p99 %.0f max %.0f | dur us/step p50 %.0f p99 %.0f max %.0f"%(*np.percentile(st,[50,90,99,100]), *np.percentile(en,[50,99,100]), *(np.percentile(dur,[50,99,100])/250)))
late=np.argsort(st)[-12:]
print("latest starters:", [(int(i), int(st[i]), int(en[i])) for i in late])
slow=np.argsort(en)[-8:]
print("latest finishers:", [(int(i), int(st[i]), int(en[i])) for i in slow])

ch=_keep[-1]; B=1024
raw=ch.sched.cpu().numpy()
cost=raw[:4*B].view(np.float32).astype(np.float64); n4=int(raw[4*B:4*B+4].view(np.int32)[0]); ns=int(raw[4*B+4:4*B+8].view(np.int32)[0]); order=raw[4*B+32:4*B+32+4*B].view(np.int32); print('header', raw[4*B:4*B+32].view(np.int32)); print('first-launch n4', int(raw[4*B+16:4*B+20].view(np.int32)[0]), 'chains unfinished at the last launch', int(raw[4*B+20:4*B+24].view(np.int32)[0]))
print("n4",n4,"nsplit",ns,"chains starting later than 1 ms:",int((st>1000).sum()))
# cost is ticks per step at one candidate; predicted us/step per class
cons={1:1.0,2:1.65,4:2.35}
cls=np.full(B,2); cls[order[:n4]]=4; cls[order[B-2*ns:]]=1
pred_us=cost/np.array([cons[c] for c in cls])/100.0
act_us=dur/250
for c in (4,2,1):
    m=cls==c
    print("class J=%d: n=%d predicted us/step p50 %.0f max %.0f | actual p50 %.0f max %.0f | ratio actual/pred p10 %.2f p50 %.2f p90 %.2f"%(c,m.sum(),np.median(pred_us[m]),pred_us[m].max(),np.median(act_us[m]),act_us[m].max(),*np.percentile(act_us[m]/pred_us[m],[10,50,90])))
print("sum of per-chain busy wave-time / (2048 waves x makespan): %.3f"%(((dur*np.array([ {4:4,2:2,1:1}[c] for c in cls])).sum())/(2048*en.max())))

bb=(ctypes.c_ulonglong*8192)()
lib.mhd_debug_blk(bb,0)
raw5=np.array(list(bb),dtype=np.uint64).reshape(1024,8)[:,5]
hw=(raw5&np.uint64(0xFFFFFFFF)).astype(np.int64); xcc=((raw5>>np.uint64(32))&np.uint64(0xF)).astype(np.int64)
nwv=(raw5>>np.uint64(60)).astype(np.int64); sp=((raw5>>np.uint64(59))&np.uint64(1)).astype(np.int64)
cu=(hw>>8)&0xF; sh=(hw>>12)&1; se=(hw>>13)&7
key=xcc*1000+se*100+sh*16+cu
early=st<1000
waves=np.where(sp==1,1,nwv)
import collections
occ=collections.Counter(); quads=collections.Counter()
for i in range(1024):
    if early[i]:
        occ[key[i]]+=waves[i]
        if nwv[i]==4: quads[key[i]]+=1
print("distinct CUs seen:",len(occ),"waves resident at start: total",sum(occ.values()),"histogram of waves per CU:",sorted(collections.Counter(occ.values()).items()))
print("CUs with a quad: waves per CU histogram:",sorted(collections.Counter(occ[k] for k in quads).items()), "quads per CU:",sorted(collections.Counter(quads.values()).items()))
print("xcc distribution of early waves:",sorted(collections.Counter({x:sum(v for k,v in occ.items() if k//1000==x) for x in range(8)}).items()))

# what the best assignment of candidates could reach: smallest T with sum of waves <= 2048
def need(T, cons):
    tot = 0
    for c in cost:
        for J in sorted(cons):
            if c / cons[J] <= T:
                tot += J; break
        else:
            return 10**9
    return tot
for name, cons_ in (("1,2,4", {1:1.0,2:1.65,4:2.35}), ("1,2,3,4", {1:1.0,2:1.65,3:2.07,4:2.35}), ("1,2,3,4,8", {1:1.0,2:1.65,3:2.07,4:2.35,8:2.9}), ("1,2,4,8", {1:1.0,2:1.65,4:2.35,8:2.9})):
    for budget in (2048, 1960):
        T = 1000.0
        for cand in np.arange(150, 900, 2.0):
            if need(cand*100.0, cons_) <= budget:
                T = cand; break
        print("classes", name, "budget", budget, "-> best predicted makespan (ticks/100 = us/step):", T)
print("cost ticks percentiles (one candidate) p1 p10 p50 p90 p99 max:", np.percentile(cost,[1,10,50,90,99,100]).round())

ratio = act_us / pred_us
print("actual/predicted by xcc:", [(x, round(float(np.median(ratio[xcc==x])),3), round(float(ratio[xcc==x].max()),2)) for x in range(8)])
worst = np.argsort(ratio)[-12:]
print("worst ratios:", [(int(i), "J%d"%cls[i], "xcc%d se%d sh%d cu%d"%(xcc[i],se[i],sh[i],cu[i]), round(float(ratio[i]),2), int(act_us[i]), int(st[i])) for i in worst])
slowest = np.argsort(act_us)[-12:]
print("slowest:", [(int(i), "J%d"%cls[i], "xcc%d se%d sh%d cu%d"%(xcc[i],se[i],sh[i],cu[i]), round(float(ratio[i]),2), int(act_us[i]), int(pred_us[i])) for i in slowest])
# neighbours on the same CU: their mean duration against the chain's own
bykey = collections.defaultdict(list)
for i in range(1024): bykey[key[i]].append(i)
nb = np.array([np.mean([dur[j] for j in bykey[key[i]]]) for i in range(1024)])
print("correlation of a chain's actual/predicted with the mean duration of the chains on its CU: %.3f" % np.corrcoef(ratio, nb)[0,1])

pos = np.full(1024, -1)
for pblk in range(ns):
    for w_ in range(2): pos[order[B - 2*ns + 2*pblk + w_]] = pblk
for pblk in range(ns, B - n4 - ns): pos[order[n4 + pblk - ns]] = pblk
for q in range(n4): pos[order[q]] = 10000 + q
shown = 0
for kk_, members in sorted(bykey.items()):
    if shown < 10 and all(early[j] for j in members):
        print("CU", kk_, "block positions", sorted(int(pos[j]) for j in members)); shown += 1
np.save(os.path.join(ROOT, "gpurun_out", "cost.npy"), cost)
np.save(os.path.join(ROOT, "gpurun_out", "act_us.npy"), act_us)

rawb = np.array(list(bb), dtype=np.uint64).reshape(1024, 8)
rounds = (rawb[:, 7] >> np.uint64(40)).astype(np.float64); fw = ((rawb[:, 7] >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.float64)
cyc = rawb[:, :5].astype(np.float64); mism = rawb[:, 6].astype(np.float64)
cnt_ = ch.counters.cpu().numpy()
print("top chains by cost: (chain, cost us, class, forwards/step, kcycles/forward by phase coarse/refine/replay/verify/repair, mismatches/forward, accept rate, bounds rate)")
for i in np.argsort(-cost)[:14]:
    tot = cnt_[i, 0] + cnt_[i, 1]
    print("  %4d %5.0f J%d fw/step %.2f | %s | mism %.2f | acc %.3f bounds %.3f" % (i, cost[i] / 100, cls[i], fw[i] / 750, " ".join("%4.0f" % (c / max(fw[i], 1) / 1000) for c in cyc[i]), mism[i] / max(fw[i], 1), cnt_[i, 0] / tot, cnt_[i, 2] / tot))
print("median chains:")
for i in np.argsort(-cost)[510:514]:
    tot = cnt_[i, 0] + cnt_[i, 1]
    print("  %4d %5.0f J%d fw/step %.2f | %s | mism %.2f | acc %.3f bounds %.3f" % (i, cost[i] / 100, cls[i], fw[i] / 750, " ".join("%4.0f" % (c / max(fw[i], 1) / 1000) for c in cyc[i]), mism[i] / max(fw[i], 1), cnt_[i, 0] / tot, cnt_[i, 2] / tot))
