from pnp_svrg_amd.algorithms import pnp_sgd, tune_pnp_sgd  # noqa: F401
