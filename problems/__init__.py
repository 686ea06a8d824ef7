"""Drop-in `problems` package: same import surface as vmonardo/pnp-svrg (problems/__init__.py:4-7),
implemented by pnp_svrg_amd.problems on the MI355X."""
import os, sys; sys.path.append(os.path.dirname(os.path.realpath(__file__)))  # flat-import style of the reference
from pnp_svrg_amd.problems import Problem, CSMRI, Deblur, PhaseRetrieval
