from pnp_svrg_amd.denoisers import RealSN_DnCNNDenoiser  # noqa: F401
