import os, sys, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch; torch.zeros(1,device='cuda')
import mcmc_amd._lib as L
lib=ctypes.CDLL(L.LIB_PATH)
a=ctypes.c_int(); b=ctypes.c_int()
print(lib.mhd_debug_occupancy(ctypes.byref(a),ctypes.byref(b)), "mh blocks/CU", a.value, "fwd blocks/CU", b.value)
p=torch.cuda.get_device_properties(0); print(p.multi_processor_count, p.max_threads_per_multi_processor, getattr(p,'shared_memory_per_multiprocessor',None), p.regs_per_multiprocessor if hasattr(p,'regs_per_multiprocessor') else None)
