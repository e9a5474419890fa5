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
ph1 = st > 1000
print("second launch: chains %d, starts at us %.0f; their finish us p50 %.0f p90 %.0f p99 %.0f max %.0f | first-launch-only chains finish p50 %.0f max %.0f" % (ph1.sum(), st[ph1].min() if ph1.any() else 0, *np.percentile(en[ph1], [50, 90, 99, 100]), np.median(en[~ph1]), en[~ph1].max()))
print("makespan us %.0f" % en.max())
late = np.argsort(en)[-8:]
lib.mhd_debug_blk(bb0 := (ctypes.c_ulonglong*8192)(), 0)
r5 = np.array(list(bb0), dtype=np.uint64).reshape(1024, 8)[:, 5]
r4 = np.array(list(tt),dtype=np.float64).reshape(2048,2)[1024:, 0]
st0 = (r4 - r4.min()) / 100.0
print("first launch: chains starting later than 1 ms: %d; their starts us: %s" % ((st0 > 1000).sum(), sorted(int(v) for v in st0[st0 > 1000])))
p0 = np.array(list(tt),dtype=np.uint64).reshape(2048,2)[1024:, 1]
p0_steps = (p0 >> np.uint64(48)).astype(np.int64); p0_us = ((p0 >> np.uint64(12)) & np.uint64(0xFFFFFFFF)).astype(np.float64)/100; p0_nw = (p0 & np.uint64(7)).astype(np.int64); p0_NW = ((p0 >> np.uint64(4)) & np.uint64(15)).astype(np.int64)
print("first launch by class: ", [(j, int((p0_nw==j).sum()), "steps p10/p50/p90 %d %d %d" % tuple(np.percentile(p0_steps[p0_nw==j],[10,50,90])), "us %.0f" % np.median(p0_us[p0_nw==j])) for j in (1,2,4) if (p0_nw==j).any()])
slowp0 = np.argsort(p0_steps)[:12]
print("fewest steps in the first launch (chain, steps, us, candidates, cost us):", [(int(i), int(p0_steps[i]), int(p0_us[i]), int(p0_nw[i]), int(cost[i]/100)) for i in slowp0])
sb = (r5 >> np.uint64(32)).astype(np.int64); jw = (r5 & np.uint64(0xFF)).astype(np.int64)
print("latest finishers (chain, cost us, step at which its last launch began, candidates there, start us, finish us, us per step there):", [(int(i), int(cost[i]/100), int(sb[i]), int(jw[i]), int(st[i]), int(en[i]), int((en[i]-st[i])/max(250-sb[i],1))) for i in late])
m1 = ph1
print("second launch: us per step by candidates: J2 median %.0f, J4 median %.0f | steps left at its start: p10 %.0f p50 %.0f p90 %.0f max %.0f" % (np.median(((en-st)/np.maximum(250-sb,1))[m1 & (jw==2)]), np.median(((en-st)/np.maximum(250-sb,1))[m1 & (jw==4)]), *np.percentile((250-sb)[m1], [10,50,90,100])))
bb=(ctypes.c_ulonglong*8192)()
lib.mhd_debug_blk(bb,0)
raw5=np.array(list(bb),dtype=np.uint64).reshape(1024,8)[:,5]
rawb = np.array(list(bb), dtype=np.uint64).reshape(1024, 8)
rounds = (rawb[:, 7] >> np.uint64(40)).astype(np.float64); fw = ((rawb[:, 7] >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.float64)
cyc = rawb[:, :5].astype(np.float64); mism = rawb[:, 6].astype(np.float64)
cnt_ = ch.counters.cpu().numpy()
seqf = (rawb[:, 7] & np.uint64(0xFFFFF)).astype(np.float64)
print("top chains by cost: (chain, cost us, class, forwards/step, kcycles/forward by phase coarse/refine/replay/verify/repair, mismatches/forward, accept rate, bounds rate)")
for i in np.argsort(-cost)[:14]:
    tot = cnt_[i, 0] + cnt_[i, 1]
    print("  %4d %5.0f J%d fw/step %.2f | %s | mism %.2f | acc %.3f bounds %.3f" % (i, cost[i] / 100, 0, fw[i] / 750, " ".join("%4.0f" % (c / max(fw[i], 1) / 1000) for c in cyc[i]), mism[i] / max(fw[i], 1), cnt_[i, 0] / tot, cnt_[i, 2] / tot))
print("median chains:")
for i in np.argsort(-cost)[510:514]:
    tot = cnt_[i, 0] + cnt_[i, 1]
    print("  %4d %5.0f J%d fw/step %.2f | %s | mism %.2f | acc %.3f bounds %.3f" % (i, cost[i] / 100, 0, fw[i] / 750, " ".join("%4.0f" % (c / max(fw[i], 1) / 1000) for c in cyc[i]), mism[i] / max(fw[i], 1), cnt_[i, 0] / tot, cnt_[i, 2] / tot))

gc = (ctypes.c_uint * 32768)(); lib.mhd_debug_chain(gc)
ev = np.array(list(gc), dtype=np.float64).reshape(1024, 32)
names = ["curves", "hint tried", "hint gave up", "-", "repairs", "predicted again", "same root", "not plain", "same answer", "stepwise", "finished sequentially"] + ["-"]*15 + ["resynchronised"]
def show(i):
    n = max(ev[i, 0], 1)
    return "chain %4d cost %4.0f | curves %5d | per curve: " % (i, cost[i] / 100, ev[i, 0]) + ", ".join("%s %.2f" % (names[j], ev[i, j] / n) for j in (1, 2, 9, 4, 8, 6, 26, 5, 7, 10))
print("events of the most expensive chains:")
for i in np.argsort(-cost)[:8]: print("  " + show(i))
print("of median chains:")
for i in np.argsort(-cost)[510:514]: print("  " + show(i))
tot = ev.sum(0); print("not plain, per curve: no bracket %.3f, more than 62 scan steps %.3f, scan reaches the floor %.3f, scan ends without a change %.3f, root outside the final ends %.3f; in the first period %.3f" % tuple(tot[j]/tot[0] for j in (11,12,13,14,15,3)))
print("all chains per curve: " + ", ".join("%s %.3f" % (names[j], tot[j] / tot[0]) for j in (1, 2, 9, 4, 8, 6, 26, 5, 7, 10)))

rp = max(tot[4], 1)
print("repairs: sequential search ended above the predicted root %.3f (more than 8 dc above %.3f), below %.3f, near %.3f | first wrong sign in the scan %.3f, in the halvings %.3f | period >=30 %.3f, 20-29 %.3f, 10-19 %.3f, <10 %.3f" % tuple(tot[j] / rp for j in (16, 21, 17, 18, 19, 20, 22, 23, 24, 25)))
