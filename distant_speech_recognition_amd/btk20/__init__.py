"""btk20 -- the reference's SWIG module surface for the beamforming hot path (btk20_src/*/*.i): same class names, constructor
kwargs, iterator protocol (`__iter__` = reset + self, `.next()`, `StopIteration` at end) and error behaviour as the reference's
btk20.{stream,feature,modulated,beamformer,postfilter,dereverberation}.

There is ONE host layer: the C++ node layer of host/ (libbtk20hip.so, linked against the C-ABI of include/btkhip.h), bound to Python
with pybind11 (distant_speech_recognition_amd.btk20cpp).  This package and its sub-modules only give it the reference's import names;
a node pulls its finite upstream once, runs the whole block through the HIP kernels and serves frames from a host mirror, so `next()`
keeps the reference's per-frame semantics (node-owned buffer, same-frame caching, end-of-stream)."""
from ..btk20cpp import *   # noqa: F401,F403
from ..btk20cpp import __all__  # noqa: F401

BACKEND = "cpp"
