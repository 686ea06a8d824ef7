from pnp_svrg_amd.problems import Deblur  # noqa: F401
