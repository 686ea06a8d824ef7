"""pnp_svrg_amd -- MI355X-native hot path of the PnP-SVRG/SAGA/SARAH engine.

csrc/   hand-written gfx950 HIP kernels + the C ABI (include/pnp_hip.h)
ops     torch-tensor front end of the C ABI
The reference-compatible call surface lives in the top-level packages
`algorithms`, `problems`, `denoisers` (host-side mirror of vmonardo/pnp-svrg).
"""
__version__ = '0.1.0'
