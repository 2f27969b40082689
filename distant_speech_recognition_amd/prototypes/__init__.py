"""Nyquist(M) analysis / synthesis prototypes for the oversampled DFT filter banks.

The reference ships one designed pair (unit_test/prototype.ny/{h,g}-M256-m4-r1.pickle) and a designer
(tools/filterbank/design_nyquist_filter.py, Kumatani et al. ICASSP 2018) for everything else.  nyquist_m4_r1.npz holds
the pairs that designer produces for M = 512, 1024, 2048 at m = 4, r = 1 (the BASELINE configs) next to the shipped M = 256
pair; it is generated in the
dev container by tests/golden/gen_prototypes.py, which imports the reference's designer and checks the whole pipeline
against the shipped M = 256 pair (<= 1e-10).  Data only: the designer itself is not part of this package.
"""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}


def available():
    """(M, m, r) geometries with a designed prototype pair in this package."""
    return [(256, 4, 1), (512, 4, 1), (1024, 4, 1), (2048, 4, 1)]


def load(M, m=4, r=1):
    """(h, g): float64 [m*M] analysis / synthesis prototypes as the reference's designer produces them."""
    if (M, m, r) not in available():
        raise KeyError("no designed Nyquist(M) prototype for M=%d m=%d r=%d; design one with the reference's "
                       "tools/filterbank/design_nyquist_filter.py (see tests/golden/gen_prototypes.py)" % (M, m, r))
    if "z" not in _CACHE:
        with np.load(os.path.join(_HERE, "nyquist_m4_r1.npz")) as z:      # read everything, keep no open file (fork-safe)
            _CACHE["z"] = {k: np.array(z[k], np.float64) for k in z.files}
    z = _CACHE["z"]
    return z["h_%d" % M].copy(), z["g_%d" % M].copy()
