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
# The Metropolis-Hastings kernels keep a chain's state across out-of-line calls of the search
# (256 registers per lane at two waves per SIMD).  The compiler's machine-level hoisting of
# loop invariants out of the round loop made ~280 registers live across those calls, all of them
# spilled to scratch memory and reloaded around every forward model; with it off, and
# instructions sunk into loops where that avoids a spill, 14 are (320 instead of 1304 bytes of
# scratch per lane) and the chains run 3-4 % faster.  Results do not depend on either switch
# (instruction order only; every multiply-add is written out in the sources).
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]
CODEGEN_FLAGS = ["-mllvm", "-disable-machine-licm", "-mllvm", "-sink-insts-to-avoid-spills"]
FLAGS = BASE_FLAGS + CODEGEN_FLAGS


# A test build: every third interpolated root of the deferring period search is replaced by a
# wrong one, so that the later verification of the inferred halvings fails and the periods are
# searched again; and the candidates of the Metropolis-Hastings kernels are told to give their
# curves up without reason, so that their steps have to be run again.  Results must be those of
# the product build (tests/test_gpu_deferred_tails.py).
OUT_SABOTAGE = os.path.join(HERE, "lib", "libmcmc_amd_sabotage.so")


def _compile(out, extra, force, verbose):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not force and os.path.exists(out) and all(
            os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS):
        return out
    cmd = [HIPCC] + FLAGS + extra + ["-o", out, SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    sys.stderr.write(res.stderr)
    if res.returncode != 0 and "Unknown command line argument" in res.stderr:
        # a compiler that does not know the two back-end switches: the same library, slower
        print("mcmc_amd.build: this hipcc does not know %s; building without (slower kernels)"
              % " ".join(CODEGEN_FLAGS), file=sys.stderr)
        res = subprocess.run([HIPCC] + BASE_FLAGS + extra + ["-o", out, SRC])
    if res.returncode != 0:
        raise subprocess.CalledProcessError(res.returncode, cmd)
    return out


# The profiling twin (-DMHD_PROFILE): the same kernels with cycle counters per search stage, per
# chain timings and event counts, read through the mhd_debug_* exports that only this build has
# (scratch/prof_mh2.py, scratch/prof_short.py).  Not loaded by the package unless MCMC_AMD_LIB
# names it.
OUT_PROFILE = os.path.join(HERE, "lib", "libmcmc_amd_prof.so")
# Its light form (-DMHD_PROFILE -DMHD_PROFILE_LIGHT) records only when each chain's launches
# begin and end: the stage counters slow the kernels by a quarter and stretch the chains that
# need repairs, this one runs at full speed (scratch/prof_tail.py).
OUT_PROFILE_LIGHT = os.path.join(HERE, "lib", "libmcmc_amd_prof_light.so")


# The same library for more chains than the GPU holds at two waves per SIMD: three waves per SIMD
# (168 registers per lane), workgroups small enough in LDS for three to a compute unit, which
# limits it to problems of at most 64 periods (data values).  Chains beyond 1024 per GPU run
# 10 ... 19 % faster on it (1536: 6.0 against 5.1 M steps/s, 4096: 7.2 against 6.2 M); 1024 and
# fewer are faster on the library above.  mcmc_amd.engine picks it by the number of chains;
# callers of the C ABI load the one they want.
OUT_MANY = os.path.join(HERE, "lib", "libmcmc_amd_w3.so")
MANY_FLAGS = ["-DMHD_MAX_PERIODS=64", "-DMHD_WAVE_LDS_BYTES=5120", "-DMHD_WAVES_PER_SIMD=3"]


def build(force=False, verbose=False):
    return _compile(OUT, [], force, verbose)


def build_many_chains(force=False, verbose=False):
    return _compile(OUT_MANY, MANY_FLAGS, force, verbose)


def build_test_variants(force=False, verbose=False):
    return _compile(OUT_SABOTAGE, ["-DMHD_DEFER_SABOTAGE", "-DMHD_ABORT_SABOTAGE"], force, verbose)


def build_profile(force=False, verbose=False, light=False):
    if light:
        return _compile(OUT_PROFILE_LIGHT, ["-DMHD_PROFILE", "-DMHD_PROFILE_LIGHT",
                                            "-Wno-unused-value", "-Wno-unused-variable"],
                        force, verbose)
    return _compile(OUT_PROFILE, ["-DMHD_PROFILE", "-Wno-unused-value"], force, verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_test_variants(force="--force" in sys.argv, verbose=True)
    build_many_chains(force="--force" in sys.argv, verbose=True)
    if "--profile" in sys.argv:
        build_profile(force="--force" in sys.argv, verbose=True)
        build_profile(force="--force" in sys.argv, verbose=True, light=True)
    print(OUT)
