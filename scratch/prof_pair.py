import os, sys, ctypes, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mcmc_amd._lib as L
L.LIB_PATH=os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
import torch; torch.cuda.init(); torch.zeros(1,device="cuda"); lib=L.lib()
sys.argv=['bench.py','--steps','500','--warmup','250','--no-cpu']+sys.argv[1:]
import runpy, io, contextlib
import numpy as np
bb=(ctypes.c_ulonglong*8192)()
lib.mhd_debug_blk(bb,1)
buf=io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT,'bench.py'), run_name='__main__')
d=json.loads(buf.getvalue())
print("value %.0f ms/step %.3f exec/lit %.2f"%(d['value'], d['ms_per_step'], d['roofline']['executed_evals_per_launch']/d['roofline']['algorithmic_evals_per_launch']))
tt=(ctypes.c_ulonglong*4096)()
lib.mhd_debug_times(tt)
a=np.array(list(tt),dtype=np.float64).reshape(2048,2)[:1024]
t0=a[:,0].min()
st=(a[:,0]-t0)/100.0; en=(a[:,1]-t0)/100.0
dur=en-st
print("last launch (250 steps): start us max %.0f | per-chain us/step: min %.0f p10 %.0f median %.0f p90 %.0f p99 %.0f max %.0f mean %.0f"%(st.max(), *(np.percentile(dur,q)/250 for q in (0,10,50,90,99,100)), dur.mean()/250))
print("duration histogram us/step:", np.histogram(dur/250,bins=12))
lib.mhd_debug_blk(bb,0)
raw=np.array(list(bb),dtype=np.uint64).reshape(1024,8)
b=np.zeros((1024,8)); b[:,0]=(raw[:,7]>>np.uint64(40)); b[:,1]=((raw[:,7]>>np.uint64(20))&np.uint64(0xFFFFF)); b[:,2]=(raw[:,7]&np.uint64(0xFFFFF)); cyc=raw[:,:7].astype(np.float64)
nst=750.0
out=(ctypes.c_ulonglong*16)(); hh=(ctypes.c_ulonglong*16)(); lib.mhd_debug_profile_hi(hh); lib.mhd_debug_profile(out,0); o=list(out); print("coarse: calls %d rounds/call %.2f hint kept %.2f | hint-secant calls %d failed %.4f"%(o[15], o[14]/max(o[15],1), o[13]/max(o[15],1), o[11], o[12]/max(o[11],1)), end=" "); h=hh; print("failed %d after %.2f rounds, done %d after %.2f rounds"%(h[0],h[3]/max(h[0],1),h[2],h[1]/max(h[2],1)))
h2=(ctypes.c_ulonglong*16)(); lib.mhd_debug_profile_hi2(h2); n_=max(o[15],1); print("coarse per call: first-period rounds %.2f, shaped rounds %.2f resolving %.2f periods each, lattice rounds %.2f resolving %.2f each, carry rounds %.2f, with hint %.2f, hint misses %.2f"%(h2[0]/n_, h2[1]/n_, h2[2]/max(h2[1],1), h2[3]/n_, h2[4]/max(h2[3],1), h2[5]/n_, h2[6]/n_, h2[7]/n_))
print("shaped-round failures: no sign change in the window %d, several %d; failing q mean %.2f; failures at q<2 by period <10/<20/<30/>=30: %s"%(h2[8],h2[9],h2[10]/max(h2[8]+h2[9],1),[int(v) for v in h2[12:16]]))
h3=(ctypes.c_ulonglong*16)(); lib.mhd_debug_profile_hi3(h3); nr_=max(h3[5],1); print("replay per period (cycles): set-up %.0f, scan %.0f, halvings without the test %.0f, halvings with it %.0f, finish %.0f (periods %d)"%(h3[0]/nr_,h3[1]/nr_,h3[2]/nr_,h3[3]/nr_,h3[4]/nr_,h3[5]))
print("max shift/dc histogram (<.25,<.5,<1,<2,<4,<8,<16,<32,<64,<128,more):", [int(v) for v in hh[5:16]])
print("round: scan+forward cycles %.3g, barrier wait cycles %.3g (%.3f of the sum)"%(o[10],o[9],o[9]/(o[9]+o[10])))
print("per chain over 750 steps: rounds/step mean %.3f, forwards/step %.3f, sequential-search share %.3f"%(b[:,0].mean()/nst, b[:,1].mean()/nst, b[:,2].sum()/b[:,1].sum()))
order=np.argsort(dur)
for label,idxs in (("fastest 3",order[:3]),("median 3",order[510:513]),("slowest 10",order[-10:])):
    print(label)
    for i in idxs:
        print("  blk %4d %.0f us/step | rounds/step %.3f forwards/step %.3f seq %.2f step %.2f | kcycles/fwd coarse %.0f refine %.0f replay %.0f verify %.0f repair %.0f giveup %.0f mism/fwd %.2f"%(i,dur[i]/250,b[i,0]/nst,b[i,1]/nst,b[i,2]/max(b[i,1],1),cyc[i,6]/max(b[i,1],1),*(cyc[i,j]/1e3/max(b[i,1],1) for j in range(6)),cyc[i,6]/max(b[i,1],1)))
