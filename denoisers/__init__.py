"""Drop-in `denoisers` package: same import surface as vmonardo/pnp-svrg (denoisers/__init__.py:4-8),
implemented by pnp_svrg_amd.denoisers on the MI355X."""
import os, sys; sys.path.append(os.path.dirname(os.path.realpath(__file__)))  # flat-import style of the reference
from pnp_svrg_amd.denoisers import Denoise, BM3DDenoiser, RealSN_DnCNNDenoiser, NLMDenoiser, TVDenoiser
