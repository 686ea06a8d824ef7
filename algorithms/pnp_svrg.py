from pnp_svrg_amd.algorithms import pnp_svrg, tune_pnp_svrg  # noqa: F401
