from pnp_svrg_amd.denoisers import MMODenoiser  # noqa: F401
