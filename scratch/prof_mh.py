import os, sys, ctypes, json, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mcmc_amd._lib as L
L.LIB_PATH=os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
import torch; torch.cuda.init(); torch.zeros(1,device="cuda"); lib=L.lib()
out=(ctypes.c_ulonglong*16)()
sys.argv=['bench.py','--steps','400','--warmup','100','--no-cpu']+sys.argv[1:]
import runpy, io, contextlib
lib.mhd_debug_profile(out,1)
buf=io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT,'bench.py'), run_name='__main__')
d=json.loads(buf.getvalue())
lib.mhd_debug_profile(out,0)
o=list(out); n=1024*500*0.87
print("value %.0f ms/step %.3f exec/lit %.2f"%(d['value'], d['ms_per_step'], d['roofline']['executed_evals_per_launch']/d['roofline']['algorithmic_evals_per_launch']))
print("kernel cycles per wave-step %.0f, wallclock ticks(100MHz) per wave-step %.1f => clock %.2f GHz, => us per step %.1f"%(o[9]/(1024*500), o[10]/(1024*500), o[9]/max(o[10],1)/10.0, o[10]/(1024*500)/100.0))
print("cycles per forward: coarse %.0f refine %.0f replay %.0f verify %.0f repair %.0f giveup %.0f | mismatch events %d v1-tail periods %d clean %d | repairs: same-c %d same-root %d re-predict %d"%tuple([o[i]/n for i in range(6)]+[o[6],o[7],o[8],o[11],o[12],o[13]]))

import numpy as np
tt=(ctypes.c_ulonglong*4096)()
lib.mhd_debug_times(tt)
a=np.array(list(tt),dtype=np.float64).reshape(2048,2)[:1024]
t0=a[:,0].min()
st=(a[:,0]-t0)/100.0; en=(a[:,1]-t0)/100.0
print("last launch: block start us: min %.0f median %.0f p90 %.0f max %.0f | end us: min %.0f median %.0f max %.0f | duration median %.0f"%(st.min(),np.median(st),np.percentile(st,90),st.max(),en.min(),np.median(en),en.max(),np.median(en-st)))
print("start histogram (10 bins):", np.histogram(st,bins=10)[0].tolist())

bb=(ctypes.c_ulonglong*8192)()
lib.mhd_debug_blk(bb,0)
b=np.array(list(bb),dtype=np.float64).reshape(1024,8)
dur=en-st
order=np.argsort(dur)
names=["coarse","refine","replay","verify","repair","giveup","nmismatch"]
for label,idxs in (("fastest 5",order[:5]),("median 5",order[510:515]),("slowest 8",order[-8:])):
    print(label)
    for i in idxs:
        print("  blk %4d dur %.0f us | Mcycles(total over all launches): "%(i,dur[i])+" ".join("%s %.1f"%(n,b[i,j]/1e6) for j,n in enumerate(names[:6]))+" | mismatches %d"%b[i,6])
