"""btk20 -- host-side mirror of the reference's SWIG module surface for the beamforming hot path.

Same class names, constructor kwargs, iterator protocol (`__iter__` = reset + self, `.next()`,
`StopIteration` at end of stream) and error behaviour as the reference's
btk20.{stream,feature,modulated,beamformer,postfilter} (btk20_src/*/*.i).  Every node computes on the
MI355X through the C-ABI (include/btkhip.h): a node pulls its finite upstream once, runs the
whole block through the HIP kernels and then serves frames from a host mirror, so `next()`
keeps the reference's per-frame semantics (node-owned buffer, same-frame caching, end-of-stream).
"""
import os as _os
BACKEND = _os.environ.get("BTK20_BACKEND", "cpp")
if BACKEND == "cpp":
    # the C++ node layer (host/, pybind11): the default.  BTK20_BACKEND=python selects the ctypes mirror below, which is kept as the
    # executable specification the C++ nodes are tested against (same tests, both backends)
    from ..btk20cpp import *   # noqa: F401,F403
elif BACKEND != "python":
    raise ImportError("BTK20_BACKEND must be 'cpp' or 'python', got %r" % BACKEND)
else:
    from .common import *      # noqa: F401,F403
    from .stream import *      # noqa: F401,F403
    from .feature import *     # noqa: F401,F403
    from .modulated import *   # noqa: F401,F403
    from .beamformer import *  # noqa: F401,F403
    from .postfilter import *  # noqa: F401,F403
    from .dereverberation import *  # noqa: F401,F403
