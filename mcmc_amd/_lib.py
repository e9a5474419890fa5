"""ctypes binding of the C ABI in include/mcmc_amd.h.

The shared library is built in-tree by `python -m mcmc_amd.build` (hipcc,
--offload-arch=gfx950).  There is no CPU fallback: if the library is missing
or cannot be loaded this module raises, and every product entry point fails.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MCMC_AMD_LIB: another build of the same library, for A/B measurements)
LIB_PATH = os.environ.get("MCMC_AMD_LIB") or os.path.join(_HERE, "lib", "libmcmc_amd.so")

c_int32, c_double, c_void_p = ctypes.c_int32, ctypes.c_double, ctypes.c_void_p

ST_OK, ST_BOUNDS, ST_PHYSICS, ST_NOROOT, ST_ZERODIV = 0, 1, 2, 3, 4
RAYLEIGH, LOVE = 0, 1
PHASE, GROUP, PHASE_AND_GROUP, PHASE_AND_ELL, ELL = 0, 1, 2, 3, 4
NORMAL, ASYM_LAPLACE = 0, 1
ROT_ACCUMULATE, ROT_FRAME, PROPOSE_UNIFORM = 1, 2, 4
MAX_LAYERS, MAX_PERIODS = 15, 128
ABI_VERSION = 8

EXPORTS = (
    "mhd_abi_version",
    "mhd_last_error",
    "mhd_init",
    "mhd_debug_occupancy",
    "mhd_wait_timeouts",
    "mhd_close_decisions",
    "mhd_sched_bytes",
    "mhd_sched_min_steps",
    "mhd_secular_batch",
    "mhd_debug_secular_pairs",
    "mhd_forward_batch",
    "mhd_loglik_batch",
    "mhd_mh_steps",
    "mhd_mh_steps_static",
    "mhd_static_reset",
    "mhd_stepsize_tuning",
    "mhd_cov_accumulate",
    "mhd_tempering_sweep",
)


class Problem(ctypes.Structure):
    """struct mhd_problem"""
    _fields_ = [
        ("n_layers", c_int32), ("n_periods", c_int32), ("wave", c_int32), ("mode", c_int32),
        ("itype", c_int32), ("noise_dist", c_int32), ("n_noise_a", c_int32),
        ("n_noise_b", c_int32), ("vpvs", c_double), ("dc", c_double), ("dt", c_double),
        ("periods", c_void_p), ("obs", c_void_p), ("noise_a", c_void_p), ("noise_b", c_void_p),
        ("bounds", c_void_p), ("n_layers_b", c_void_p), ("search", c_int32),
        ("candidates", c_int32),
    ]


class Chains(ctypes.Structure):
    """struct mhd_chains"""
    _fields_ = [
        ("m", c_void_p), ("logL", c_void_p), ("pred", c_void_p), ("params_prev", c_void_p),
        ("width", c_void_p), ("rot", c_void_p), ("cov_sum", c_void_p), ("n_cov", c_void_p),
        ("counters", c_void_p), ("counters_ind", c_void_p), ("beta", c_void_p),
        ("sched", c_void_p),
    ]


class Diag(ctypes.Structure):
    """struct mhd_diag"""
    _fields_ = [("hist_counters", c_void_p), ("hist_counters_ind", c_void_p),
                ("hist_cov", c_void_p)]


class MhdError(RuntimeError):
    pass


_lib = None
_lib_many = None
# the build of the same library for more chains than the GPU holds at once (three waves per
# SIMD, at most MANY_MAX_PERIODS data values per curve; mcmc_amd/build.py): used by
# mcmc_amd.engine for such chain sets when it has been built, MCMC_AMD_MANY=0 turns that off
MANY_LIB_PATH = os.path.join(_HERE, "lib", "libmcmc_amd_w3.so")
MANY_MAX_PERIODS = 64


def lib():
    """Load libmcmc_amd.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MhdError(
            f"{LIB_PATH} not found: build the HIP library first (python -m mcmc_amd.build). "
            "There is no CPU fallback.")
    _lib = _load(LIB_PATH)
    return _lib


def lib_many():
    """The many-chains build (once), or None when it has not been built, is switched off, or
    MCMC_AMD_LIB names another library to run."""
    global _lib_many
    if _lib_many is None:
        if (os.environ.get("MCMC_AMD_MANY", "1") == "0" or os.environ.get("MCMC_AMD_LIB")
                or not os.path.exists(MANY_LIB_PATH)):
            return None
        lib()
        _lib_many = _load(MANY_LIB_PATH)
    return _lib_many


def loaded():
    """The libraries loaded so far."""
    return [L for L in (_lib, _lib_many) if L is not None]


def _load(path):
    # torch ships its own HIP runtime; it has to be the one already mapped when this library,
    # which is linked against libamdhip64 too, is loaded (the other order leaves the process
    # with two runtimes, and the second one finds no device)
    import torch  # noqa: F401
    L = ctypes.CDLL(path)
    pp, cp = ctypes.POINTER(Problem), ctypes.POINTER(Chains)
    L.mhd_abi_version.restype = ctypes.c_int
    L.mhd_abi_version.argtypes = []
    L.mhd_last_error.restype = ctypes.c_char_p
    L.mhd_last_error.argtypes = []
    L.mhd_init.argtypes = [c_int32]
    L.mhd_debug_occupancy.argtypes = [c_void_p, c_void_p]
    L.mhd_wait_timeouts.argtypes = [c_void_p]
    L.mhd_close_decisions.argtypes = [c_void_p]
    L.mhd_sched_bytes.argtypes = [c_int32]
    L.mhd_sched_bytes.restype = ctypes.c_size_t
    L.mhd_secular_batch.argtypes = [pp, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                    c_void_p, c_void_p]
    L.mhd_debug_secular_pairs.argtypes = L.mhd_secular_batch.argtypes
    L.mhd_forward_batch.argtypes = [pp, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
    L.mhd_loglik_batch.argtypes = [pp, c_void_p, c_int32, c_void_p, c_void_p]
    L.mhd_sched_min_steps.argtypes = []
    L.mhd_sched_min_steps.restype = ctypes.c_int
    L.mhd_mh_steps.argtypes = [pp, cp, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_double,
                               c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                               c_void_p, ctypes.POINTER(Diag), c_void_p, c_void_p]
    L.mhd_mh_steps_static.argtypes = L.mhd_mh_steps.argtypes
    L.mhd_static_reset.argtypes = [c_void_p]
    L.mhd_stepsize_tuning.argtypes = [pp, cp, c_int32, c_void_p]
    L.mhd_cov_accumulate.argtypes = [pp, cp, c_int32, c_void_p]
    L.mhd_tempering_sweep.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                      ctypes.c_int64, ctypes.c_int64, c_void_p, c_void_p]
    for name in EXPORTS:
        getattr(L, name)
    if L.mhd_abi_version() != ABI_VERSION:
        raise MhdError(f"{os.path.basename(path)}: ABI version mismatch; rebuild")
    return L


def check(rc, L=None):
    if rc != 0:
        raise MhdError(f"mhd call failed ({rc}): {(L or lib()).mhd_last_error().decode()}")


def close_decisions():
    """Accept tests settled so far on the current device with u within 16 units in the last place
    of exp(dl) (mhd_close_decisions): the only decisions that numpy's exponential could take the
    other way.  Synchronises with the device."""
    lib()
    total = 0
    for L in loaded():
        n = ctypes.c_uint64(0)
        check(L.mhd_close_decisions(ctypes.byref(n)), L)
        total += int(n.value)
    return total


def check_waits():
    """Raises if a wait between the waves of a workgroup has ever timed out on the current device
    (mhd_wait_timeouts): the results since then are not valid.  Synchronises with the device."""
    lib()
    total = 0
    for L in loaded():
        n = ctypes.c_uint64(0)
        check(L.mhd_wait_timeouts(ctypes.byref(n)), L)
        total += int(n.value)
    if total:
        raise MhdError(f"{total} wait(s) between the waves of a chain timed out on the device: "
                       "the chains computed since are not valid")
