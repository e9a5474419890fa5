"""Does the gloo backend gather CUDA tensors (two ranks on one GPU)?  Run under torchrun."""
import os, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
x = torch.full((4,), float(r), dtype=torch.float64, device="cuda")
out = torch.zeros(8, dtype=torch.float64, device="cuda")
print(r, "before all_gather_into_tensor", flush=True)
try:
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    print(r, "gathered", out.tolist(), flush=True)
except Exception as e:
    print(r, "failed:", type(e).__name__, str(e)[:200], flush=True)
lst = [torch.zeros(4, dtype=torch.float64, device="cuda") for _ in range(2)]
dist.all_gather(lst, x)
print(r, "all_gather list ok", [t.tolist() for t in lst], flush=True)
dist.barrier()
dist.destroy_process_group()
