from pnp_svrg_amd.problems import PhaseRetrieval  # noqa: F401
