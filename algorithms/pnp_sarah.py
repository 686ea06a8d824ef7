from pnp_svrg_amd.algorithms import pnp_sarah, tune_pnp_sarah  # noqa: F401
