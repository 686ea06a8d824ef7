from pnp_svrg_amd.problems import Problem  # noqa: F401
