from pnp_svrg_amd.algorithms import pnp_saga, tune_pnp_saga  # noqa: F401
