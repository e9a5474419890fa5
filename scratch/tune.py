"""Build variants of the library (name=flags ...) in parallel, each with the package's own flags
plus its own (BASE="..." replaces the package's), then print their names for ab_traj.sh.
usage: tune.py name="-DX=1 -DY=2" ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd.build import FLAGS
if os.environ.get("BASE") is not None:
    FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + os.environ["BASE"].split()
src = os.path.join(ROOT, "mcmc_amd", "csrc", "mhd_api.hip")
lib = os.path.join(ROOT, "mcmc_amd", "lib")
for f in os.listdir(lib):
    if f.startswith("libv_"):
        os.remove(os.path.join(lib, f))
procs = []
names = ["libmcmc_amd.so"]
for arg in sys.argv[1:]:
    name, flags = arg.split("=", 1)
    out = os.path.join(lib, "libv_%s.so" % name)
    names.append("libv_%s.so" % name)
    procs.append(subprocess.Popen(["hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include")] + flags.split() + [src, "-o", out]))
procs.append(subprocess.Popen(["hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), src, "-o", os.path.join(lib, "libmcmc_amd.so")]))
assert all(p.wait() == 0 for p in procs)
print(" ".join(names))
