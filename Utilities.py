from pnp_svrg_amd.utilities import display_results, metrics_line, metrics_row  # noqa: F401  (reference Utilities.py)
