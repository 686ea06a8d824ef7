from pnp_svrg_amd.denoisers import BM3DDenoiser  # noqa: F401
