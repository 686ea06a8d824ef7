"""CPU oracle for the PnP-SVRG/SAGA/SARAH hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy (float64) restatement of the reference algorithms
(vmonardo/pnp-svrg @ v1) and of the third-party routines the reference leans on
(scikit-image 0.18 `estimate_sigma` / `denoise_wavelet` / `denoise_nl_means`,
PyWavelets 1.1.1 db1/db2 filters, skimage PSNR).  Every function cites the
reference file:line it follows.

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` -- as the checker / the timed CPU baseline, never as a product
code path.  Nothing under `pnp_svrg_amd/`, `algorithms/`, `problems/` or
`denoisers/` imports this package; the product path raises when the HIP library
is missing instead of falling back to it.

Parity pin: the oracle is pinned against golden vectors produced by running the
real reference in the build container (`tests/golden/make_golden.py`, run with
/opt/conda/bin/python3.9 against /root/reference) -- see
`tests/test_oracle_golden.py`.  Pieces that are NOT pinned that way are flagged
"parity unpinned" where they are defined (pylops Bilinear).
"""
from . import denoise, problems, loops  # noqa: F401
