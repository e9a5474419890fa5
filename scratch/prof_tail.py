"""Where the chains of the last 500-step call of a bench run end (profiling build): the call's
first launch (ends at its deadline) and its second; who is last and with how many candidates.
python scratch/prof_tail.py [bench args]"""
import os, sys, ctypes, json
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mcmc_amd._lib as L
L.LIB_PATH=os.environ.get("MCMC_AMD_LIB") or os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
import torch; torch.cuda.init(); torch.zeros(1,device="cuda"); lib=L.lib()
sys.argv=['bench.py','--steps','1000','--warmup','200','--no-cpu']+sys.argv[1:]
import runpy, io, contextlib
bb=(ctypes.c_ulonglong*8192)()
buf=io.StringIO()
lib.mhd_debug_blk(bb,1)
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT,'bench.py'), run_name='__main__')
d=json.loads(buf.getvalue())
print("value %.0f launch_ms %.3f"%(d['value'], d['roofline']['launch_ms']))
tt=(ctypes.c_ulonglong*4096)()
lib.mhd_debug_times(tt)
a=np.array(list(tt),dtype=np.uint64)
last=a[:2048].astype(np.float64).reshape(1024,2)       # start, end of each chain's last launch
first_start=a[2048::2].astype(np.float64)               # start of the call's first launch
packed=a[2049::2]
steps1=(packed>>np.uint64(48)).astype(int); ticks1=((packed>>np.uint64(12))&np.uint64(0xFFFFFFFF)).astype(float)
nw1=(packed&np.uint64(7)).astype(int)
t0=first_start.min()
end1=(first_start-t0+ticks1)/100.0
st2=(last[:,0]-t0)/100.0; en2=(last[:,1]-t0)/100.0
done1=steps1>=500
print("first launch: chains done in it %d; end us p10 %.0f med %.0f max %.0f; candidates 1/2/4: %d/%d/%d"%(done1.sum(),*np.percentile(end1,[10,50,100]),(nw1==1).sum(),(nw1==2).sum(),(nw1==4).sum()))
two=~done1
print("second launch: %d chains, from step p10 %d med %d p90 %d; start us med %.0f; end us p10 %.0f med %.0f p90 %.0f p99 %.0f max %.0f"%(two.sum(),*np.percentile(steps1[two],[10,50,90]),np.median(st2[two]),*np.percentile(en2[two],[10,50,90,99,100])))
lib.mhd_debug_blk(bb,0)
b=np.array(list(bb),dtype=np.uint64).reshape(1024,8)
nw2=(b[:,5]&np.uint64(0xffffffff)).astype(int)
cc=(ctypes.c_uint*32768)(); lib.mhd_debug_chain(cc)
ev=np.array(list(cc),dtype=np.int64).reshape(1024,32)
rep=ev[:,4]/np.maximum(ev[:,0],1)
nw2c = np.where(two & (nw2 < 8), nw2, 0)
print("second launch: candidates 1/2/4: %d/%d/%d; end us of the chains with 1: med %.0f max %.0f, with 2: med %.0f max %.0f, with 4: med %.0f max %.0f" % (
    (nw2c == 1).sum(), (nw2c == 2).sum(), (nw2c == 4).sum(),
    *[f(en2[nw2c == k]) if (nw2c == k).any() else 0 for k in (1, 2, 4) for f in (np.median, np.max)]))
order=np.argsort(np.where(two,en2,0))
print("the last 25 chains: end us, step the second launch began at, its candidates, repairs per forward (whole run), hint failures per try")
for i in order[-25:]:
    print("  chain %4d end %7.0f from step %3d nw2 %d nw1 %d repairs/fwd %.2f hintfail %.2f notplain/fwd %.2f"%(i,en2[i],steps1[i],nw2[i],nw1[i],rep[i],ev[i,2]/max(ev[i,1],1),ev[i,7]/max(ev[i,0],1)))
print("all: repairs/fwd p50 %.2f p90 %.2f p99 %.2f; corr(repairs/fwd, end) %.2f"%(*np.percentile(rep,[50,90,99]),np.corrcoef(rep[two],en2[two])[0,1]))
# where the cycles of the last chain's forwards go, beside the median chain's (whole run, all
# waves of the chain): predict, refine, replay, verify, repair
stage = b[:, :5].astype(np.float64)
fw = np.maximum(ev[:, 0], 1)
last_i = order[-1]
names = ["predict", "refine", "replay", "verify", "repair"]
med = np.median(stage / fw[:, None], axis=0)
print("cycles per forward by stage: last chain %d: %s = %.0f; median chain: %s = %.0f" % (
    last_i, " ".join("%s %.0f" % (n, x) for n, x in zip(names, stage[last_i] / fw[last_i])), stage[last_i].sum() / fw[last_i],
    " ".join("%s %.0f" % (n, x) for n, x in zip(names, med)), med.sum()))
print("last chain: forwards %d, hint tried %d gave up %d, repairs %d, same-answer %d, same-root-other-lattice %d, predicted anew %d, resynchronised %d, beyond MAX_REPAIRS %d, not plain %d" % (
    ev[last_i, 0], ev[last_i, 1], ev[last_i, 2], ev[last_i, 4], ev[last_i, 8], ev[last_i, 6], ev[last_i, 5], ev[last_i, 26], ev[last_i, 10], ev[last_i, 7]))
print("  repairs: search ended above the predicted root %d, below %d, at it %d; wrong sign in the scan %d, in the halvings %d; period >=30: %d, 20-29: %d, 10-19: %d, <10: %d" % tuple(ev[last_i, i] for i in (16, 17, 18, 19, 20, 22, 23, 24, 25)))
