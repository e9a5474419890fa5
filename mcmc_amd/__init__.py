"""MI355X-native hot path for Metropolis-Hastings inversion of surface-wave
dispersion curves: HIP kernels (mcmc_amd/csrc) behind the C ABI of
include/mcmc_amd.h, and a Python mirror of the reference's src/inversion API.
"""
__all__ = ["build", "engine", "inversion"]
