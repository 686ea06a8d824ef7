from pnp_svrg_amd.denoisers import TVDenoiser  # noqa: F401
