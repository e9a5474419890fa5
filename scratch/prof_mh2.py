"""Stage profile of the MH kernels (profiling build): cycles per forward by stage, rounds of the
predictors, for bench.py's headline run.  python scratch/prof_mh2.py [bench args]"""
import os, sys, ctypes, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mcmc_amd._lib as L
L.LIB_PATH=os.environ.get('MCMC_AMD_LIB') or os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
import torch; torch.cuda.init(); torch.zeros(1,device="cuda"); lib=L.lib()
sys.argv=['bench.py','--steps','400','--warmup','100','--no-cpu']+sys.argv[1:]
import runpy, io, contextlib
out=(ctypes.c_ulonglong*16)()
lib.mhd_debug_profile(out,1)
buf=io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT,'bench.py'), run_name='__main__')
d=json.loads(buf.getvalue())
g=[]
for f in (lib.mhd_debug_profile, ):
    f(out,0); g+=list(out)
for f in (lib.mhd_debug_profile_hi, lib.mhd_debug_profile_hi2, lib.mhd_debug_profile_hi3):
    f(out); g+=list(out)
r=d['roofline']
print("value %.0f exec/lit %.2f"%(d['value'], r['executed_evals_per_launch']/r['algorithmic_evals_per_launch']))
nf = g[11] if g[11] else 1   # forwards that tried predict_hint = all hinted forwards
ncoarse=max(g[15],1)
print("forwards (hint tried) %d, hint gave up %d (%.2f), coarse calls %d, coarse rounds/call %.2f"%(g[11], g[12], g[12]/nf, g[15], g[14]/ncoarse))
print("hint iterations: settled %.2f (n %d) gave up %.2f (n %d)"%(g[17]/max(g[18],1), g[18], g[19]/max(g[16],1), g[16]))
tot=sum(g[i] for i in range(6))
print("cycles per forward: predict %.0f refine %.0f replay %.0f verify %.0f repair %.0f giveup %.0f total %.0f"%tuple([g[i]/nf for i in range(6)]+[tot/nf]))
print("repairs %d (%.3f per forward), clean passes %d"%(g[6], g[6]/nf, g[8]))
print("raw", g[:48])
