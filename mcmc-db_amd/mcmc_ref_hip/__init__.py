"""mcmc_ref_hip -- MI355X-native statistics hot path behind mcmc_ref's plug-in surface.

Host-side mirror of the reference's `mcmc_ref.backends` / `diagnostics` / `compare`
interfaces; all arithmetic on draws runs in hand-written HIP kernels reached
through the C ABI declared in include/mcmcref_hip.h (ctypes, no torch).
"""
__version__ = "0.1.0"
