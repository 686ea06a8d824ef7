from pnp_svrg_amd.problems import CSMRI  # noqa: F401
