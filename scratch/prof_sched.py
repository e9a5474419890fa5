import os, sys, ctypes, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch; torch.cuda.init(); torch.zeros(1,device="cuda")
import mcmc_amd._lib as L
L.LIB_PATH=os.path.join(ROOT,'mcmc_amd','lib','libmcmc_amd_prof.so')
lib=L.lib()
sys.argv=['bench.py','--steps','500','--warmup','250','--no-cpu']+sys.argv[1:]
import runpy, io, contextlib
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
