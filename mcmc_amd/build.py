"""Build libmcmc_amd.so in-tree with hipcc for gfx950:  python -m mcmc_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mhd_api.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", h)
                for h in ("mhd_device.hpp", "mhd_search2.hpp", "mhd_math.hpp")] + \
       [os.path.join(os.path.dirname(HERE), "include", "mcmc_amd.h")]
OUT = os.path.join(HERE, "lib", "libmcmc_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def build(force=False, verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(
            os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [HIPCC] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
