"""Per-chain durations and rounds of a short call (the driver's bench: 5 warm-up + 20 steps)."""
import os, sys, ctypes, json
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mcmc_amd._lib as L
L.LIB_PATH=os.environ.get('MCMC_AMD_LIB') or os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
import torch; torch.cuda.init(); torch.zeros(1,device="cuda"); lib=L.lib()
steps = sys.argv[1] if len(sys.argv) > 1 else '20'
sys.argv=['bench.py','--steps',steps,'--warmup','5','--no-cpu']+sys.argv[2:]
import runpy, io, contextlib
bb=(ctypes.c_ulonglong*8192)()
buf=io.StringIO()
lib.mhd_debug_blk(bb,1)
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT,'bench.py'), run_name='__main__')
d=json.loads(buf.getvalue())
print("value %.0f ms/step %.4f launch_ms %.3f"%(d['value'], d['ms_per_step'], d['roofline']['launch_ms']))
tt=(ctypes.c_ulonglong*4096)()
lib.mhd_debug_times(tt)
a=np.array(list(tt),dtype=np.float64).reshape(2048,2)[:1024]
t0=a[:,0].min()
st=(a[:,0]-t0)/100.0; en=(a[:,1]-t0)/100.0
print("last launch: start us min %.0f med %.0f max %.0f | end us p10 %.0f med %.0f p90 %.0f p99 %.0f max %.0f"%(st.min(),np.median(st),st.max(),*np.percentile(en,[10,50,90,99,100])))
lib.mhd_debug_blk(bb,0)
b=np.array(list(bb),dtype=np.uint64).reshape(1024,8)
rounds=(b[:,7]>>np.uint64(40)).astype(float); fw=((b[:,7]>>np.uint64(20))&np.uint64(0xFFFFF)).astype(float)
print("rounds per chain (warm-up + timed): mean %.1f p90 %.1f max %.0f | forwards mean %.1f max %.0f"%(rounds.mean(),np.percentile(rounds,90),rounds.max(),fw.mean(),fw.max()))
dur=en-st
c=np.corrcoef(rounds,dur)[0,1]
print("duration us: mean %.0f p50 %.0f p90 %.0f max %.0f; corr(rounds,duration) %.2f; us per round mean %.0f"%(dur.mean(),np.median(dur),np.percentile(dur,90),dur.max(),c,(dur/np.maximum(rounds,1)).mean()))
cc=(ctypes.c_uint*32768)()
lib.mhd_debug_chain(cc)
ev=np.array(list(cc),dtype=np.int64).reshape(1024,32)
order=np.argsort(en) if int(steps) >= 100 else np.argsort(dur)
last_start=b[:,5]
names={0:"fwd",1:"hint",2:"hintfail",4:"repair",5:"repredict",6:"sameroot",7:"notplain",8:"samec",9:"stepwise",10:"seqfinish",26:"resync"}
for label,idxs in (("median 3",order[510:513]),("slowest 40",order[-40:])):
    print(label)
    for i in idxs:
        print("  chain %4d end %6.0f us (last launch from step %3d with %d waves) dur %5.0f us rounds %2d fw %2d | Mcyc predict %.2f refine %.2f replay %.2f verify %.2f repair %.2f giveup %.2f | "%(i,en[i],int(last_start[i])>>32,int(last_start[i])&0xffffffff,dur[i],rounds[i],fw[i],*(b[i,j]/1e6 for j in range(6)))+" ".join("%s %d"%(n,ev[i,k]) for k,n in names.items()))
tot=ev.sum(axis=0)
print("all chains: forwards %d hint tried %d gave up %d | repairs %d (after a settled hint %d) | sequential answer above/below/at the predicted root %d/%d/%d | first wrong sign in scan %d in halvings %d | resynced %d repredicted %d seqfinish %d notplain %d"%(tot[0],tot[1],tot[2],tot[4],tot[27],tot[16],tot[17],tot[18],tot[19],tot[20],tot[26],tot[5],tot[10],tot[7]))
rep=ev[:,4]/np.maximum(ev[:,0],1)
print("repairs per forward over chains: p50 %.2f p90 %.2f p99 %.2f max %.2f; corr with duration %.2f"%(*np.percentile(rep,[50,90,99,100]),np.corrcoef(rep,dur)[0,1]))
