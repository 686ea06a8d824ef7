from pnp_svrg_amd.denoisers import Denoise  # noqa: F401
