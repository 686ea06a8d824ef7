from pnp_svrg_amd.denoisers import NLMDenoiser  # noqa: F401
