"""Drop-in `algorithms` package: same import surface as vmonardo/pnp-svrg (algorithms/__init__.py:4-8),
implemented by pnp_svrg_amd.algorithms on the MI355X."""
import os, sys; sys.path.append(os.path.dirname(os.path.realpath(__file__)))  # flat-import style of the reference
from pnp_svrg_amd.algorithms import (pnp_gd, tune_pnp_gd, pnp_sgd, tune_pnp_sgd, pnp_svrg, tune_pnp_svrg,
                                     pnp_saga, tune_pnp_saga, pnp_sarah, tune_pnp_sarah)
from pnp_svrg_amd.algorithms import CountingClock  # noqa: F401  (extension: deterministic clock for `clock=`)
