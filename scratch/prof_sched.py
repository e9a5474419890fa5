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
print("latest finishers (chain, cost us, finish us):", [(int(i), int(cost[i]/100), int(en[i])) for i in late])
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
