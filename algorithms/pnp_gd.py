from pnp_svrg_amd.algorithms import pnp_gd, tune_pnp_gd  # noqa: F401
